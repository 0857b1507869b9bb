// C-ABI of the multi-device 3D loop-closure context: ndt2d_multi_api.hpp for ndt3d_batch contexts (one ndt3d_batch +
// one host thread per device; the device-resident form ends in ONE grouped ncclAllGather of the 408-byte result rows).
// Included at the end of ndt2d_api.hip after ndt2d_multi_api.hpp (RCCL_TRY, ndt2d_multi_plan) and ndt3d_batch_api.hpp.
#pragma once

#include <thread>
#include <vector>

struct ndt3d_multi {
  std::vector<ndt3d_batch*> ctx;
  int32_t iterations_hint = 30;      // expected evaluations per pair over all levels (shard balancing)
  // device-resident form (ndt3d_multi_align_dev): one RCCL communicator per context, created on first use
  std::vector<ncclComm_t> comms;
  std::vector<ndt3d_result*> d_send;  // [ctx]: this device's rows, padded to the longest shard
  std::vector<ndt3d_result*> d_recv;  // [ctx]: every device's rows after the all-gather
  size_t gather_cap = 0;              // rows per shard the buffers hold
};

int32_t ndt3d_multi_destroy(ndt3d_multi* m) {
  if (!m) return NDT_OK;
  for (size_t d = 0; d < m->ctx.size(); ++d) {
    (void)hipSetDevice(m->ctx[d]->device);
    if (d < m->d_send.size() && m->d_send[d]) (void)hipFree(m->d_send[d]);
    if (d < m->d_recv.size() && m->d_recv[d]) (void)hipFree(m->d_recv[d]);
  }
  for (ncclComm_t c : m->comms) if (c && ndt::rccl().ok) (void)ndt::rccl().CommDestroy(c);
  for (ndt3d_batch* b : m->ctx) ndt3d_batch_destroy(b);
  delete m;
  return NDT_OK;
}

int32_t ndt3d_multi_create_pyramid(const ndt3d_params* levels, int32_t n_levels, const int32_t* device_ids,
                                   int32_t n_devices, ndt3d_multi** out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = nullptr;
  if (!levels || n_levels < 1 || n_devices < 0 || (n_devices > 0 && !device_ids)) return NDT_ERR_INVALID_ARG;
  const int visible = ndt_device_count();
  if (visible <= 0) { ndt::set_error("no HIP device visible: this library has no CPU fallback"); return NDT_ERR_NO_DEVICE; }
  ndt3d_multi* m = new (std::nothrow) ndt3d_multi();
  if (!m) return NDT_ERR_ALLOC;
  m->iterations_hint = 0;
  for (int32_t i = 0; i < n_levels; ++i)
    m->iterations_hint += levels[i].fixed_iterations > 0 ? levels[i].fixed_iterations : 30;
  const int n = n_devices > 0 ? n_devices : visible;
  for (int i = 0; i < n; ++i) {
    ndt3d_batch* b = nullptr;
    const int32_t st = ndt3d_batch_create_pyramid(levels, n_levels, n_devices > 0 ? device_ids[i] : i, &b);
    if (st != NDT_OK) { ndt3d_multi_destroy(m); return st; }
    m->ctx.push_back(b);
  }
  *out = m;
  return NDT_OK;
}

int32_t ndt3d_multi_create(const ndt3d_params* p, const int32_t* device_ids, int32_t n_devices, ndt3d_multi** out) {
  if (!p) { if (out) *out = nullptr; return NDT_ERR_INVALID_ARG; }
  return ndt3d_multi_create_pyramid(p, 1, device_ids, n_devices, out);
}

int32_t ndt3d_multi_device_count(const ndt3d_multi* m) { return m ? static_cast<int32_t>(m->ctx.size()) : 0; }

int32_t ndt3d_multi_align(ndt3d_multi* m, const float* tx, const float* ty, const float* tz, const uint64_t* toff,
                          const float* sx, const float* sy, const float* sz, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt3d_result* results) {
  if (!m || m->ctx.empty() || !tx || !ty || !tz || !toff || !sx || !sy || !sz || !soff || !init || !results || n_pairs == 0)
    return NDT_ERR_INVALID_ARG;
  const int nd = static_cast<int>(m->ctx.size());
  std::vector<uint64_t> begin(nd + 1);
  int32_t st = ndt2d_multi_plan(nd, toff, soff, n_pairs, m->iterations_hint, begin.data());
  if (st != NDT_OK) return st;
  std::vector<int32_t> status(nd, NDT_OK);
  std::vector<std::string> message(nd);
  auto run = [&](int d) {
    const size_t k0 = begin[d], k1 = begin[d + 1];
    if (k1 == k0) return;
    // the shard's offsets rebased to its first point, so only its own points are uploaded
    std::vector<uint64_t> to(k1 - k0 + 1), so(k1 - k0 + 1);
    for (size_t k = k0; k <= k1; ++k) { to[k - k0] = toff[k] - toff[k0]; so[k - k0] = soff[k] - soff[k0]; }
    status[d] = ndt3d_batch_align(m->ctx[d], tx + toff[k0], ty + toff[k0], tz + toff[k0], to.data(), sx + soff[k0],
                                  sy + soff[k0], sz + soff[k0], so.data(), init + 6 * k0, k1 - k0, results + k0);
    if (status[d] != NDT_OK) message[d] = ndt_last_error();   // last_error is per thread
  };
  std::vector<std::thread> workers;
  for (int d = 1; d < nd; ++d) workers.emplace_back(run, d);
  run(0);
  for (std::thread& w : workers) w.join();
  for (int d = 0; d < nd; ++d)
    if (status[d] != NDT_OK) { ndt::set_error(message[d].c_str()); return status[d]; }
  return NDT_OK;
}


// Device-resident form with the RCCL gather (BASELINE.json north_star: "shards scan pairs across the 8
// GPUs of one node with a final RCCL gather over xGMI").  One host thread enqueues everything: the
// batch kernels on every context's stream, then one grouped ncclAllGather of the padded result rows
// on the same streams - no host copy of a result, no host synchronisation between alignment and gather.
int32_t ndt3d_multi_align_dev(ndt3d_multi* m, const float* const* d_tx, const float* const* d_ty, const float* const* d_tz,
                              const uint64_t* const* d_toff, const float* const* d_sx, const float* const* d_sy,
                              const float* const* d_sz, const uint64_t* const* d_soff, const double* const* d_init,
                              const size_t* n_pairs, ndt3d_result** d_results_all, size_t* shard_stride,
                              ndt3d_result* results) {
  if (!m || m->ctx.empty() || !d_tx || !d_ty || !d_tz || !d_toff || !d_sx || !d_sy || !d_sz || !d_soff || !d_init || !n_pairs)
    return NDT_ERR_INVALID_ARG;
  const int nd = static_cast<int>(m->ctx.size());
  size_t longest = 0, total = 0;
  for (int d = 0; d < nd; ++d) {
    if (n_pairs[d] > 0x7fffffffull) return NDT_ERR_INVALID_ARG;
    if (n_pairs[d] > 0 && (!d_tx[d] || !d_ty[d] || !d_tz[d] || !d_toff[d] || !d_sx[d] || !d_sy[d] || !d_sz[d] || !d_soff[d] || !d_init[d]))
      return NDT_ERR_INVALID_ARG;
    longest = n_pairs[d] > longest ? n_pairs[d] : longest;
    total += n_pairs[d];
  }
  if (total == 0) return NDT_ERR_INVALID_ARG;
  { const int32_t rs = require_rccl(); if (rs != NDT_OK) return rs; }
  if (m->comms.empty()) {
    // one communicator per context, all in this process (ncclCommInitAll); a device listed twice
    // cannot take part in a collective with itself
    std::vector<int> devs(nd);
    for (int d = 0; d < nd; ++d) {
      devs[d] = m->ctx[d]->device;
      for (int e = 0; e < d; ++e)
        if (devs[e] == devs[d]) { ndt::set_error("the RCCL gather needs distinct devices"); return NDT_ERR_INVALID_ARG; }
    }
    m->comms.assign(nd, nullptr);
    const ncclResult_t r = ndt::rccl().CommInitAll(m->comms.data(), nd, devs.data());
    if (r != ncclSuccess) {
      m->comms.clear();
      ndt::last_error() = std::string("ncclCommInitAll: ") + ndt::rccl().GetErrorString(r);
      return NDT_ERR_RCCL;
    }
  }
  if (longest > m->gather_cap) {
    const int32_t gs = grow_gather_buffers(m->ctx, m->d_send, m->d_recv, &m->gather_cap, longest + longest / 4 + 16);
    if (gs != NDT_OK) return gs;
  }
  // the gather moves `stride` rows per shard: the longest shard (padding rows are zero)
  const size_t stride = longest;
  for (int d = 0; d < nd; ++d) {
    HIP_TRY(hipSetDevice(m->ctx[d]->device));
    hipStream_t st = m->ctx[d]->stream;
    if (n_pairs[d] < stride)
      HIP_TRY(hipMemsetAsync(m->d_send[d] + n_pairs[d], 0, (stride - n_pairs[d]) * sizeof(ndt3d_result), st));
    if (n_pairs[d] > 0) {
      const float* const t3[3] = {d_tx[d], d_ty[d], d_tz[d]};
      const float* const s3[3] = {d_sx[d], d_sy[d], d_sz[d]};
      const int32_t bs = batch3_launch(m->ctx[d], t3, reinterpret_cast<const unsigned long long*>(d_toff[d]), s3,
                                       reinterpret_cast<const unsigned long long*>(d_soff[d]), d_init[d], n_pairs[d],
                                       m->d_send[d], st);
      if (bs != NDT_OK) return bs;
    }
  }
  ndt::TraceRange range("ndt3d_multi: RCCL all-gather of the result rows");
  static_assert(sizeof(ndt3d_result) % sizeof(double) == 0, "rows travel as doubles");
  const size_t count = stride * (sizeof(ndt3d_result) / sizeof(double));
  RCCL_TRY(ndt::rccl().GroupStart());
  for (int d = 0; d < nd; ++d) {
    const ncclResult_t r = ndt::rccl().AllGather(m->d_send[d], m->d_recv[d], count, ncclDouble, m->comms[d], m->ctx[d]->stream);
    if (r != ncclSuccess) {
      (void)ndt::rccl().GroupEnd();
      ndt::last_error() = std::string("ncclAllGather: ") + ndt::rccl().GetErrorString(r);
      return NDT_ERR_RCCL;
    }
  }
  RCCL_TRY(ndt::rccl().GroupEnd());
  if (results) {     // global pair order, padding dropped, from the first device's copy of the gather
    HIP_TRY(hipSetDevice(m->ctx[0]->device));
    size_t k = 0;
    for (int d = 0; d < nd; ++d) {
      if (n_pairs[d] > 0)
        HIP_TRY(hipMemcpyAsync(results + k, m->d_recv[0] + (size_t)d * stride, n_pairs[d] * sizeof(ndt3d_result),
                               hipMemcpyDeviceToHost, m->ctx[0]->stream));
      k += n_pairs[d];
    }
  }
  for (int d = 0; d < nd; ++d) {
    HIP_TRY(hipSetDevice(m->ctx[d]->device));
    HIP_TRY(hipStreamSynchronize(m->ctx[d]->stream));
    if (d_results_all) d_results_all[d] = m->d_recv[d];
  }
  if (shard_stride) *shard_stride = stride;
  return NDT_OK;
}
