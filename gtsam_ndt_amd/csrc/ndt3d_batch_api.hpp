// C-ABI of the 3D loop-closure batch (included at the end of ndt2d_api.hip: one translation unit).
#pragma once
#include <vector>

#include "ndt3d_batch.hpp"

struct ndt3d_batch {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<ndt3d_params> levels;   // coarse to fine; one entry unless created as a pyramid
  int n_cu = 0;
  unsigned int* d_queue = nullptr;
  unsigned char* d_slab = nullptr;    // [n_cu][kB3SlabBytes]
  unsigned char* d_gslab = nullptr;   // [global_blocks][kG3SlabBytes]: tables of the global-memory variant
  int global_blocks = ndt::kG3BlocksStart;   // its workgroups: a few to begin with, one per CU once a call has used them
  bool global_pinned = false;         // ... unless NDT_TUNE_BATCH_GLOBAL_WORKGROUPS fixed the number (grow_global_slabs, ndt2d_batch_api.hpp)
  unsigned int* h_fb_seen = nullptr;  // pinned host word the global-memory variant counts its pairs in
  int* d_fb = nullptr;                // [n_pairs]: marks of the pairs k_batch3 left to the global-memory variant
  size_t fb_cap = 0;
  // staging for the host-pointer entry point (one capacity per buffer)
  float *d_t[3] = {nullptr, nullptr, nullptr}, *d_s[3] = {nullptr, nullptr, nullptr};
  size_t cap_t[3] = {0, 0, 0}, cap_s[3] = {0, 0, 0};
  unsigned long long *d_toff = nullptr, *d_soff = nullptr;
  double* d_init = nullptr;
  ndt3d_result* d_out = nullptr;
  size_t cap_toff = 0, cap_soff = 0, cap_init = 0, cap_out = 0;
  std::vector<ndt3d_handle*> fallback;   // single-pair path (one handle per level) for pairs over the LDS capacity
};

static_assert(sizeof(ndt::Result3Dev) == sizeof(ndt3d_result), "Result3Dev mirrors ndt3d_result");
static_assert(offsetof(ndt::Result3Dev, H) == offsetof(ndt3d_result, H), "Result3Dev layout");
static_assert(offsetof(ndt::Result3Dev, score) == offsetof(ndt3d_result, score), "Result3Dev layout");
static_assert(offsetof(ndt::Result3Dev, status) == offsetof(ndt3d_result, status), "Result3Dev layout");

namespace {

int32_t batch3_launch(ndt3d_batch* b, const float* const d_t[3], const unsigned long long* d_toff, const float* const d_s[3],
                      const unsigned long long* d_soff, const double* d_init, size_t n_pairs, ndt3d_result* d_out,
                      hipStream_t st) {
  ndt::TraceRange range("ndt3d_batch: voxel grid build + Gauss-Newton loops on chip");
  ndt::Batch3Args a{};
  a.tx = d_t[0]; a.ty = d_t[1]; a.tz = d_t[2]; a.toff = d_toff;
  a.sx = d_s[0]; a.sy = d_s[1]; a.sz = d_s[2]; a.soff = d_soff;
  a.init = d_init;
  a.out = reinterpret_cast<ndt::Result3Dev*>(d_out);
  a.queue = b->d_queue;
  grow_global_slabs(b, &b->d_gslab, ndt::kG3SlabBytes, b->n_cu < ndt::kG3Blocks ? b->n_cu : ndt::kG3Blocks, st);
  a.slab = b->d_slab;
  a.gslab = b->d_gslab;
  a.fb_seen = b->h_fb_seen;
  a.n_pairs = (int)n_pairs;
  const int blocks = (int)(n_pairs < (size_t)b->n_cu ? n_pairs : (size_t)b->n_cu);
  const int blocks_fb = (int)(n_pairs < (size_t)b->global_blocks ? n_pairs : (size_t)b->global_blocks);
  if (n_pairs > b->fb_cap) {
    if (b->d_fb) (void)hipFree(b->d_fb);
    b->d_fb = nullptr; b->fb_cap = 0;
    const size_t want = n_pairs + n_pairs / 4 + 64;
    HIP_TRY(hipMalloc((void**)&b->d_fb, want * sizeof(int)));
    b->fb_cap = want;
  }
  a.fb_marks = b->d_fb;
  for (size_t lv = 0; lv < b->levels.size(); ++lv) {
    const ndt3d_params& p = b->levels[lv];
    a.chain = lv > 0 ? 1 : 0;
    a.min_points = p.min_points;
    a.fixed_iterations = p.fixed_iterations;
    a.cell = p.cell_size;
    a.eig_ratio = p.eig_ratio;
    a.prm.d1 = (float)p.d1; a.prm.d2 = (float)p.d2;
    a.prm.hessian_mode = p.hessian_mode;
    a.prm.max_iterations = p.max_iterations;
    a.prm.min_hits = p.min_hits;
    a.prm.line_search = p.line_search;
    a.prm.eps_trans = p.eps_trans; a.prm.eps_rot = p.eps_rot;
    a.prm.step_max_trans = p.step_max_trans; a.prm.step_max_rot = p.step_max_rot;
    a.prm.step_scale = p.step_scale > 0.0 ? p.step_scale : 1.0;
    HIP_TRY(hipMemsetAsync(b->d_queue, 0, 16, st));
    HIP_TRY(hipMemsetAsync(b->d_fb, 0, n_pairs * sizeof(int), st));
    const bool newton = p.hessian_mode == NDT_HESSIAN_NEWTON;
    if (newton)
      hipLaunchKernelGGL((ndt::k_batch3<1>), dim3(blocks), dim3(ndt::kB3Threads), ndt::kB3LdsBytes, st, a);
    else
      hipLaunchKernelGGL((ndt::k_batch3<0>), dim3(blocks), dim3(ndt::kB3Threads), ndt::kB3LdsBytes, st, a);
    HIP_TRY(hipGetLastError());
    // pairs whose voxel grid does not fit the LDS carve (handed over through fb_marks): tables in global memory
    if (newton)
      hipLaunchKernelGGL((ndt::k_batch3_fallback<1>), dim3(blocks_fb), dim3(ndt::kB3Threads), ndt::kB3LdsBytes, st, a);
    else
      hipLaunchKernelGGL((ndt::k_batch3_fallback<0>), dim3(blocks_fb), dim3(ndt::kB3Threads), ndt::kB3LdsBytes, st, a);
    HIP_TRY(hipGetLastError());
  }
  return NDT_OK;
}

}  // namespace

extern "C" {

int32_t ndt3d_batch_destroy(ndt3d_batch* b) {
  if (!b) return NDT_OK;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  void* dev[] = {b->d_slab, b->d_gslab, b->d_fb, b->d_queue, b->d_t[0], b->d_t[1], b->d_t[2], b->d_s[0], b->d_s[1], b->d_s[2],
                 b->d_toff, b->d_soff, b->d_init, b->d_out};
  for (void* p : dev) if (p) (void)hipFree(p);
  if (b->h_fb_seen) (void)hipHostFree(b->h_fb_seen);
  for (ndt3d_handle* f : b->fallback) ndt3d_destroy(f);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return NDT_OK;
}

int32_t ndt3d_batch_create_pyramid(const ndt3d_params* levels, int32_t n_levels, int32_t device_id, ndt3d_batch** out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = nullptr;
  if (!levels || n_levels < 1 || n_levels > 8) return NDT_ERR_INVALID_ARG;
  for (int32_t i = 0; i < n_levels; ++i) {
    const int32_t st = check_params(&levels[i]);
    if (st != NDT_OK) return st;
    if (levels[i].overlap_grids == 4) { set_error("overlapping grids are a 2D option"); return NDT_ERR_INVALID_ARG; }
  }
  const int ndev = ndt_device_count();
  if (ndev <= 0) { set_error("no HIP device visible: this library has no CPU fallback"); return NDT_ERR_NO_DEVICE; }
  if (device_id < 0 || device_id >= ndev) return NDT_ERR_INVALID_ARG;
  ndt3d_batch* b = new (std::nothrow) ndt3d_batch();
  if (!b) return NDT_ERR_ALLOC;
  b->device = device_id;
  b->levels.assign(levels, levels + n_levels);
  auto fail = [&](int32_t code) { ndt3d_batch_destroy(b); return code; };
  if (hipSetDevice(device_id) != hipSuccess) return fail(NDT_ERR_HIP);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(NDT_ERR_HIP);
  b->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipMalloc((void**)&b->d_queue, 16) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&b->d_slab, (size_t)b->n_cu * ndt::kB3SlabBytes) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&b->d_gslab, (size_t)b->global_blocks * ndt::kG3SlabBytes) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipHostMalloc((void**)&b->h_fb_seen, 64, hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  *b->h_fb_seen = 0;
  // more than 64 KiB of dynamic LDS needs an explicit opt-in per kernel
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch3<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::kB3LdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch3<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::kB3LdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch3_fallback<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::kB3LdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch3_fallback<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::kB3LdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  *out = b;
  return NDT_OK;
}

int32_t ndt3d_batch_create(const ndt3d_params* p, int32_t device_id, ndt3d_batch** out) {
  if (!p) { if (out) *out = nullptr; return NDT_ERR_INVALID_ARG; }
  return ndt3d_batch_create_pyramid(p, 1, device_id, out);
}

void* ndt3d_batch_stream(ndt3d_batch* b) { return b ? (void*)b->stream : nullptr; }

int32_t ndt3d_batch_set_tuning(ndt3d_batch* b, int32_t knob, int64_t value) {
  if (!b || knob != NDT_TUNE_BATCH_GLOBAL_WORKGROUPS || value < 1 || value > ndt::kG3BlocksMax) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if ((int)value != b->global_blocks) {                // one table slab per workgroup: re-allocate
    unsigned char* slab = nullptr;
    if (hipMalloc((void**)&slab, (size_t)value * ndt::kG3SlabBytes) != hipSuccess) { (void)hipGetLastError(); return NDT_ERR_ALLOC; }
    (void)hipFree(b->d_gslab);
    b->d_gslab = slab;
    b->global_blocks = (int)value;
  }
  b->global_pinned = true;                             // the caller's number stands: no growth on demand
  return NDT_OK;
}

int32_t ndt3d_batch_wait_stream(ndt3d_batch* b, void* producer_stream) {
  if (!b) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(ndt::order_after(b->stream, (hipStream_t)producer_stream));
  return NDT_OK;
}

int32_t ndt3d_batch_align_dev(ndt3d_batch* b, const float* d_tx, const float* d_ty, const float* d_tz, const uint64_t* d_toff,
                              const float* d_sx, const float* d_sy, const float* d_sz, const uint64_t* d_soff,
                              const double* d_init, size_t n_pairs, ndt3d_result* d_results, void* stream) {
  if (!b || !d_tx || !d_ty || !d_tz || !d_toff || !d_sx || !d_sy || !d_sz || !d_soff || !d_init || !d_results) return NDT_ERR_INVALID_ARG;
  if (n_pairs == 0 || n_pairs > 0x7fffffffull) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  const float* const t[3] = {d_tx, d_ty, d_tz};
  const float* const s[3] = {d_sx, d_sy, d_sz};
  return batch3_launch(b, t, reinterpret_cast<const unsigned long long*>(d_toff), s,
                       reinterpret_cast<const unsigned long long*>(d_soff), d_init, n_pairs, d_results,
                       stream ? (hipStream_t)stream : b->stream);
}

int32_t ndt3d_batch_align(ndt3d_batch* b, const float* tx, const float* ty, const float* tz, const uint64_t* toff,
                          const float* sx, const float* sy, const float* sz, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt3d_result* results) {
  if (!b || !tx || !ty || !tz || !toff || !sx || !sy || !sz || !soff || !init || !results || n_pairs == 0) return NDT_ERR_INVALID_ARG;
  if (n_pairs > 0x7fffffffull) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  const size_t nt = toff[n_pairs], ns = soff[n_pairs];
  for (size_t k = 0; k < n_pairs; ++k) {
    if (toff[k + 1] < toff[k] || soff[k + 1] < soff[k] || toff[k + 1] - toff[k] > (size_t)ndt::kBatchMaxCloud ||
        soff[k + 1] - soff[k] > (size_t)ndt::kBatchMaxCloud) return NDT_ERR_INVALID_ARG;
  }
  int32_t st;
  const float* th[3] = {tx, ty, tz};
  const float* sh[3] = {sx, sy, sz};
  hipStream_t s = b->stream;
  for (int c = 0; c < 3; ++c) {
    if ((st = ensure_dev(&b->d_t[c], &b->cap_t[c], nt)) != NDT_OK) return st;
    if ((st = ensure_dev(&b->d_s[c], &b->cap_s[c], ns)) != NDT_OK) return st;
  }
  if ((st = ensure_dev(&b->d_toff, &b->cap_toff, n_pairs + 1)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_soff, &b->cap_soff, n_pairs + 1)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_init, &b->cap_init, 6 * (n_pairs + 1))) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_out, &b->cap_out, n_pairs + 1)) != NDT_OK) return st;
  for (int c = 0; c < 3; ++c) {
    HIP_TRY(hipMemcpyAsync(b->d_t[c], th[c], nt * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->d_s[c], sh[c], ns * sizeof(float), hipMemcpyHostToDevice, s));
  }
  HIP_TRY(hipMemcpyAsync(b->d_toff, toff, (n_pairs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_soff, soff, (n_pairs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_init, init, 6 * n_pairs * sizeof(double), hipMemcpyHostToDevice, s));
  st = batch3_launch(b, b->d_t, b->d_toff, b->d_s, b->d_soff, b->d_init, n_pairs, b->d_out, s);
  if (st != NDT_OK) return st;
  HIP_TRY(hipMemcpyAsync(results, b->d_out, n_pairs * sizeof(ndt3d_result), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  // pairs beyond the global-memory variant's limits as well go through the single-pair path
  for (size_t k = 0; k < n_pairs; ++k) {
    if (results[k].status != NDT_ERR_CAPACITY) continue;
    if (b->fallback.empty()) {
      for (const ndt3d_params& lp : b->levels) {
        ndt3d_handle* f = nullptr;
        st = ndt3d_create(&lp, b->device, &f);
        if (st != NDT_OK) return st;
        b->fallback.push_back(f);
      }
    }
    double pose[6];
    for (int j = 0; j < 6; ++j) pose[j] = init[6 * k + j];
    int total = 0;
    for (ndt3d_handle* f : b->fallback) {
      st = ndt3d_set_target(f, tx + toff[k], ty + toff[k], tz + toff[k], toff[k + 1] - toff[k]);
      if (st == NDT_OK) st = ndt3d_align(f, sx + soff[k], sy + soff[k], sz + soff[k], soff[k + 1] - soff[k], pose, &results[k]);
      if (st < 0) break;
      total += results[k].iterations;
      results[k].iterations = total;
      if (results[k].status != NDT_OK && results[k].status != NDT_NOT_CONVERGED) break;
      for (int j = 0; j < 6; ++j) pose[j] = results[k].pose[j];
    }
    if (st < 0) return st;
  }
  return NDT_OK;
}

}  // extern "C"
