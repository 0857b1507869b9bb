// C-ABI of the multi-device loop-closure context (included at the end of ndt2d_api.hip).
// For a C++ host process that owns all GPUs of a node itself, in two forms:
//   ndt2d_multi_align      host pointers: one ndt2d_batch + one host thread per device, pairs split into
//                          contiguous, work-balanced shards (ndt2d_multi_plan), results written straight into
//                          the caller's array - no collective, the host array is the meeting point;
//   ndt2d_multi_align_dev  device-resident shards: every context aligns its shard on its own stream and the
//                          result rows are exchanged with ONE grouped ncclAllGather (RCCL over xGMI) on those
//                          streams - north_star's "final RCCL gather".  RCCL is loaded on first use (ndt_dyn.hpp).
// Pairs are independent, so there is no exchange step during the alignments in either form.  The
// one-process-per-GPU deployment (torch.distributed, backend "nccl" = RCCL) lives in gtsam_ndt_amd/dist.py / bench.py.
#pragma once

#include <thread>
#include <vector>

struct ndt2d_multi {
  std::vector<ndt2d_batch*> ctx;
  ndt2d_params prm{};
  int32_t iterations_hint = 30;      // expected evaluations per pair over all levels (shard balancing)
  // device-resident form (ndt2d_multi_align_dev): one RCCL communicator per context, created on first use
  std::vector<ncclComm_t> comms;
  std::vector<ndt2d_result*> d_send;  // [ctx]: this device's rows, padded to the longest shard
  std::vector<ndt2d_result*> d_recv;  // [ctx]: every device's rows after the all-gather
  size_t gather_cap = 0;              // rows per shard the buffers hold
};

namespace {

// Result-row buffers of the gather: d_send[d] holds `rows` rows on device d, d_recv[d] holds rows * n_devices.
// Grows to `want` rows by allocating EVERY new buffer first and swapping afterwards: a failed allocation leaves the
// old buffers, and the capacity that describes them, exactly as they were (a later call that fits them still works).
template <typename Ctx, typename Row>
int32_t grow_gather_buffers(const std::vector<Ctx*>& ctx, std::vector<Row*>& d_send, std::vector<Row*>& d_recv,
                            size_t* cap, size_t want) {
  const int nd = static_cast<int>(ctx.size());
  std::vector<Row*> ns(nd, nullptr), nr(nd, nullptr);
  hipError_t err = hipSuccess;
  for (int d = 0; d < nd && err == hipSuccess; ++d) {
    err = hipSetDevice(ctx[d]->device);
    if (err == hipSuccess) err = hipMalloc((void**)&ns[d], want * sizeof(Row));
    if (err == hipSuccess) err = hipMalloc((void**)&nr[d], want * nd * sizeof(Row));
  }
  if (err == hipSuccess)
    for (int d = 0; d < nd && err == hipSuccess; ++d) {      // the old buffers may still be in use by the last call
      err = hipSetDevice(ctx[d]->device);
      if (err == hipSuccess) err = hipStreamSynchronize(ctx[d]->stream);
    }
  if (err != hipSuccess) {
    for (int d = 0; d < nd; ++d) {
      (void)hipSetDevice(ctx[d]->device);
      if (ns[d]) (void)hipFree(ns[d]);
      if (nr[d]) (void)hipFree(nr[d]);
    }
    ::ndt::last_error() = std::string("gather buffers: ") + hipGetErrorString(err);
    (void)hipGetLastError();
    return err == hipErrorOutOfMemory ? NDT_ERR_ALLOC : NDT_ERR_HIP;
  }
  d_send.resize(nd, nullptr);
  d_recv.resize(nd, nullptr);
  for (int d = 0; d < nd; ++d) {
    (void)hipSetDevice(ctx[d]->device);
    if (d_send[d]) (void)hipFree(d_send[d]);
    if (d_recv[d]) (void)hipFree(d_recv[d]);
    d_send[d] = ns[d];
    d_recv[d] = nr[d];
  }
  *cap = want;
  return NDT_OK;
}

inline int32_t require_rccl() {
  if (ndt::rccl().ok) return NDT_OK;
  ndt::set_error("librccl.so.1 could not be loaded: the device-resident multi-GPU gather needs RCCL (everything else does not)");
  return NDT_ERR_RCCL;
}

}  // namespace

#define RCCL_TRY(expr)                                                                    \
  do {                                                                                    \
    const ncclResult_t _r = (expr);                                                       \
    if (_r != ncclSuccess) {                                                              \
      ::ndt::last_error() = std::string(#expr) + ": " + ndt::rccl().GetErrorString(_r);           \
      return NDT_ERR_RCCL;                                                                \
    }                                                                                     \
  } while (0)

int32_t ndt2d_multi_destroy(ndt2d_multi* m) {
  if (!m) return NDT_OK;
  for (size_t d = 0; d < m->ctx.size(); ++d) {
    (void)hipSetDevice(m->ctx[d]->device);
    if (d < m->d_send.size() && m->d_send[d]) (void)hipFree(m->d_send[d]);
    if (d < m->d_recv.size() && m->d_recv[d]) (void)hipFree(m->d_recv[d]);
  }
  for (ncclComm_t c : m->comms) if (c && ndt::rccl().ok) (void)ndt::rccl().CommDestroy(c);
  for (ndt2d_batch* b : m->ctx) ndt2d_batch_destroy(b);
  delete m;
  return NDT_OK;
}

int32_t ndt2d_multi_create_pyramid(const ndt2d_params* levels, int32_t n_levels, const int32_t* device_ids,
                                   int32_t n_devices, ndt2d_multi** out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = nullptr;
  if (!levels || n_levels < 1 || n_devices < 0 || (n_devices > 0 && !device_ids)) return NDT_ERR_INVALID_ARG;
  const int visible = ndt_device_count();
  if (visible <= 0) { ndt::set_error("no HIP device visible: this library has no CPU fallback"); return NDT_ERR_NO_DEVICE; }
  ndt2d_multi* m = new (std::nothrow) ndt2d_multi();
  if (!m) return NDT_ERR_ALLOC;
  m->prm = levels[n_levels - 1];
  m->iterations_hint = 0;
  for (int32_t i = 0; i < n_levels; ++i)
    m->iterations_hint += levels[i].fixed_iterations > 0 ? levels[i].fixed_iterations : 30;
  const int n = n_devices > 0 ? n_devices : visible;
  for (int i = 0; i < n; ++i) {
    ndt2d_batch* b = nullptr;
    const int32_t st = ndt2d_batch_create_pyramid(levels, n_levels, n_devices > 0 ? device_ids[i] : i, &b);
    if (st != NDT_OK) { ndt2d_multi_destroy(m); return st; }
    m->ctx.push_back(b);
  }
  *out = m;
  return NDT_OK;
}

int32_t ndt2d_multi_create(const ndt2d_params* p, const int32_t* device_ids, int32_t n_devices, ndt2d_multi** out) {
  if (!p) { if (out) *out = nullptr; return NDT_ERR_INVALID_ARG; }
  return ndt2d_multi_create_pyramid(p, 1, device_ids, n_devices, out);
}

int32_t ndt2d_multi_device_count(const ndt2d_multi* m) { return m ? static_cast<int32_t>(m->ctx.size()) : 0; }

// shard_begin[d] .. shard_begin[d+1] = the pairs device slot d would receive for these offsets
// (exposed so a caller can pre-place data, and so the split is testable without devices)
int32_t ndt2d_multi_plan_hinted(int32_t n_shards, const uint64_t* toff, const uint64_t* soff, size_t n_pairs,
                                int32_t iterations_hint, const int32_t* pair_iterations, uint64_t* shard_begin) {
  if (n_shards <= 0 || !toff || !soff || !shard_begin) return NDT_ERR_INVALID_ARG;
  const double kk = iterations_hint > 0 ? iterations_hint : 30;
  // work of a pair = its target points once (grid build: three passes) + its source points per iteration; the
  // iterations from the caller's per-pair hint where it gives one (converged-mode batches: what the candidate took at
  // the coarser level, or last time), else the common hint
  auto work = [&](size_t k) {
    const double it = pair_iterations && pair_iterations[k] > 0 ? (double)pair_iterations[k] : kk;
    return 3.0 * double(toff[k + 1] - toff[k]) + it * double(soff[k + 1] - soff[k]) + 1.0;
  };
  double total = 0;
  for (size_t k = 0; k < n_pairs; ++k) {
    if (toff[k + 1] < toff[k] || soff[k + 1] < soff[k]) return NDT_ERR_INVALID_ARG;
    total += work(k);
  }
  size_t k = 0;
  double acc = 0;
  shard_begin[0] = 0;
  for (int d = 1; d <= n_shards; ++d) {
    const double goal = total * d / n_shards;
    // a pair goes to the shard in which its midpoint falls: contiguous, deterministic, balanced
    while (k < n_pairs && acc + 0.5 * work(k) <= goal) acc += work(k++);
    shard_begin[d] = d == n_shards ? n_pairs : k;
  }
  return NDT_OK;
}

int32_t ndt2d_multi_plan(int32_t n_shards, const uint64_t* toff, const uint64_t* soff, size_t n_pairs,
                         int32_t iterations_hint, uint64_t* shard_begin) {
  return ndt2d_multi_plan_hinted(n_shards, toff, soff, n_pairs, iterations_hint, nullptr, shard_begin);
}

int32_t ndt2d_multi_align(ndt2d_multi* m, const float* tx, const float* ty, const uint64_t* toff,
                          const float* sx, const float* sy, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt2d_result* results) {
  if (!m || m->ctx.empty() || !tx || !ty || !toff || !sx || !sy || !soff || !init || !results || n_pairs == 0)
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
    status[d] = ndt2d_batch_align(m->ctx[d], tx + toff[k0], ty + toff[k0], to.data(), sx + soff[k0], sy + soff[k0],
                                  so.data(), init + 3 * k0, k1 - k0, results + k0);
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
int32_t ndt2d_multi_align_dev(ndt2d_multi* m, const float* const* d_tx, const float* const* d_ty,
                              const uint64_t* const* d_toff, const float* const* d_sx, const float* const* d_sy,
                              const uint64_t* const* d_soff, const double* const* d_init, const size_t* n_pairs,
                              ndt2d_result** d_results_all, size_t* shard_stride, ndt2d_result* results) {
  if (!m || m->ctx.empty() || !d_tx || !d_ty || !d_toff || !d_sx || !d_sy || !d_soff || !d_init || !n_pairs)
    return NDT_ERR_INVALID_ARG;
  const int nd = static_cast<int>(m->ctx.size());
  size_t longest = 0, total = 0;
  for (int d = 0; d < nd; ++d) {
    if (n_pairs[d] > 0x7fffffffull) return NDT_ERR_INVALID_ARG;
    if (n_pairs[d] > 0 && (!d_tx[d] || !d_ty[d] || !d_toff[d] || !d_sx[d] || !d_sy[d] || !d_soff[d] || !d_init[d]))
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
      HIP_TRY(hipMemsetAsync(m->d_send[d] + n_pairs[d], 0, (stride - n_pairs[d]) * sizeof(ndt2d_result), st));
    if (n_pairs[d] > 0) {
      const int32_t bs = batch_launch(m->ctx[d], d_tx[d], d_ty[d], reinterpret_cast<const unsigned long long*>(d_toff[d]),
                                      d_sx[d], d_sy[d], reinterpret_cast<const unsigned long long*>(d_soff[d]), d_init[d],
                                      n_pairs[d], m->d_send[d], st);
      if (bs != NDT_OK) return bs;
    }
  }
  ndt::TraceRange range("ndt2d_multi: RCCL all-gather of the result rows");
  static_assert(sizeof(ndt2d_result) % sizeof(double) == 0, "rows travel as doubles");
  const size_t count = stride * (sizeof(ndt2d_result) / sizeof(double));
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
        HIP_TRY(hipMemcpyAsync(results + k, m->d_recv[0] + (size_t)d * stride, n_pairs[d] * sizeof(ndt2d_result),
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
