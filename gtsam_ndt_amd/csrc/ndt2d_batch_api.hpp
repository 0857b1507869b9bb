// C-ABI of the loop-closure batch (included at the end of ndt2d_api.hip: one translation unit,
// so the kernels of ndt2d_kernels.hpp are defined once).
#pragma once
#include <vector>

#include "ndt2d_batch.hpp"

struct ndt2d_batch {
  int device = 0;
  hipStream_t stream = nullptr;
  ndt2d_params prm{};                 // the finest (last) level
  std::vector<ndt2d_params> levels;   // coarse to fine; one entry unless created as a pyramid
  int n_cu = 0;
  unsigned int* d_queue = nullptr;
  // staging for the host-pointer entry point
  // (one capacity per buffer: a failed allocation of one must not leave its sibling's capacity standing)
  float *d_tx = nullptr, *d_ty = nullptr, *d_sx = nullptr, *d_sy = nullptr;
  size_t cap_tx = 0, cap_ty = 0, cap_sx = 0, cap_sy = 0;
  unsigned long long *d_toff = nullptr, *d_soff = nullptr;
  double* d_init = nullptr;
  ndt2d_result* d_out = nullptr;
  size_t cap_toff = 0, cap_soff = 0, cap_init = 0, cap_out = 0;
  std::vector<ndt2d_handle*> fallback;   // global-memory path (one handle per level) for pairs over the LDS capacity
  int* d_marks = nullptr;                // [n_pairs]: pairs the small variant left to the large one
  int* d_fb_list = nullptr;              // [n_pairs]: marks of the pairs the large variant left to the global-table one
  size_t marks_cap = 0;
  unsigned char* d_slab = nullptr;       // [global_blocks][BatchGlobal::kTabBytes]
  int global_blocks = ndt::kBatchGlobalBlocksStart;   // workgroups (and table slabs) of the global-table variant: a few to begin with,
  bool global_pinned = false;            // one per CU once a call has used them - unless NDT_TUNE_BATCH_GLOBAL_WORKGROUPS fixed the number
  unsigned int* h_fb_seen = nullptr;     // pinned host word the global-table variant counts its pairs in
  bool use_small = true;                 // lidar-sized pairs run on the 256-thread variant first (ndt2d_batch_set_tuning)
  int64_t last_large = -1;               // pairs the last host-pointer call's final level ran on the large variant
};

static_assert(sizeof(ndt::ResultDev) == sizeof(ndt2d_result), "ResultDev mirrors ndt2d_result");
static_assert(offsetof(ndt::ResultDev, score) == offsetof(ndt2d_result, score), "ResultDev layout");
static_assert(offsetof(ndt::ResultDev, status) == offsetof(ndt2d_result, status), "ResultDev layout");

namespace {

// The global-table variant's slabs (3.7 MB / 7.9 MB per workgroup in 2D / 3D) are most of a context's memory and most
// batches never touch them, so a context is created with a few and gets one per CU only when a previous call has
// handed pairs to that variant (counted by the kernel in a pinned host word - no synchronisation to learn it).  The
// call that first meets such pairs runs them on the starting set: slower for that call, the same results.
template <typename Ctx>
void grow_global_slabs(Ctx* b, unsigned char** slab, size_t slab_bytes, int full, hipStream_t st) {
  if (b->global_pinned || b->global_blocks >= full || !b->h_fb_seen) return;
  if (__atomic_load_n(b->h_fb_seen, __ATOMIC_RELAXED) == 0) return;
  // earlier launches on either stream may still be using the present slabs
  if (hipStreamSynchronize(st) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) { (void)hipGetLastError(); return; }
  unsigned char* bigger = nullptr;
  if (hipMalloc((void**)&bigger, (size_t)full * slab_bytes) != hipSuccess) {
    (void)hipGetLastError();
    b->global_pinned = true;            // not enough memory for the full set: stay with what there is
    return;
  }
  (void)hipFree(*slab);
  *slab = bigger;
  b->global_blocks = full;
}

int32_t batch_launch(ndt2d_batch* b, const float* d_tx, const float* d_ty, const unsigned long long* d_toff,
                     const float* d_sx, const float* d_sy, const unsigned long long* d_soff,
                     const double* d_init, size_t n_pairs, ndt2d_result* d_out, hipStream_t st) {
  ndt::TraceRange range("ndt2d_batch: grid build + Gauss-Newton loops on chip");
  ndt::BatchArgs a{};
  a.tx = d_tx; a.ty = d_ty; a.toff = d_toff;
  a.sx = d_sx; a.sy = d_sy; a.soff = d_soff;
  a.init = d_init;
  a.out = reinterpret_cast<ndt::ResultDev*>(d_out);
  a.queue = b->d_queue;
  a.n_pairs = (int)n_pairs;
  const int blocks = (int)(n_pairs < (size_t)b->n_cu ? n_pairs : (size_t)b->n_cu);
  // the small variant keeps two workgroups resident per CU
  const size_t small_max = 2 * (size_t)b->n_cu;
  const int blocks_small = (int)(n_pairs < small_max ? n_pairs : small_max);
  if (n_pairs > b->marks_cap) {
    if (b->d_marks) (void)hipFree(b->d_marks);
    if (b->d_fb_list) (void)hipFree(b->d_fb_list);
    b->d_marks = b->d_fb_list = nullptr; b->marks_cap = 0;
    const size_t want = n_pairs + n_pairs / 4 + 64;
    HIP_TRY(hipMalloc((void**)&b->d_marks, want * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&b->d_fb_list, want * sizeof(int)));
    b->marks_cap = want;
  }
  grow_global_slabs(b, &b->d_slab, ndt::BatchGlobal::kTabBytes, b->n_cu < ndt::kBatchGlobalBlocks ? b->n_cu : ndt::kBatchGlobalBlocks, st);
  a.slab = b->d_slab;
  a.fb_marks = b->d_fb_list;
  a.fb_seen = b->h_fb_seen;
  // Per resolution level: the small variant takes every pair it can hold (lidar-sized scans) and
  // marks the rest, the large variant then takes exactly the marked ones.  A later level starts
  // every pair from the pose the previous one left in d_out; stream order is the only
  // synchronisation between launches.
  for (size_t lv = 0; lv < b->levels.size(); ++lv) {
    const ndt2d_params& p = b->levels[lv];
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
    const bool newton = p.hessian_mode == NDT_HESSIAN_NEWTON;
    const int blocks_fb = (int)(n_pairs < (size_t)b->global_blocks ? n_pairs : (size_t)b->global_blocks);
    if (p.overlap_grids == 4) {
      // Biber's four overlapping grids: every pair of this level goes to the global-table variant (process_pair's NG),
      // so every pair is marked for it (any non-zero word is a mark)
      HIP_TRY(hipMemsetAsync(b->d_fb_list, 1, n_pairs * sizeof(int), st));
      a.marks = nullptr;
      a.queue = b->d_queue + 1;
      if (newton)
        hipLaunchKernelGGL((ndt::k_batch_fallback<1, 4>), dim3(blocks_fb), dim3(ndt::BatchGlobal::kThreads), ndt::BatchGlobal::kLdsBytes, st, a);
      else
        hipLaunchKernelGGL((ndt::k_batch_fallback<0, 4>), dim3(blocks_fb), dim3(ndt::BatchGlobal::kThreads), ndt::BatchGlobal::kLdsBytes, st, a);
      HIP_TRY(hipGetLastError());
      continue;
    }
    HIP_TRY(hipMemsetAsync(b->d_queue, 0, 16, st));
    HIP_TRY(hipMemsetAsync(b->d_fb_list, 0, n_pairs * sizeof(int), st));
    a.marks = nullptr;
    if (b->use_small) {
      a.marks = b->d_marks;
      a.queue = b->d_queue;
      if (newton)
        hipLaunchKernelGGL((ndt::k_batch<1, ndt::BatchSmall>), dim3(blocks_small), dim3(ndt::BatchSmall::kThreads),
                           ndt::BatchSmall::kLdsBytes, st, a);
      else
        hipLaunchKernelGGL((ndt::k_batch<0, ndt::BatchSmall>), dim3(blocks_small), dim3(ndt::BatchSmall::kThreads),
                           ndt::BatchSmall::kLdsBytes, st, a);
      HIP_TRY(hipGetLastError());
    }
    a.queue = b->d_queue + 1;                        // its own dequeue counter
    if (newton)
      hipLaunchKernelGGL((ndt::k_batch<1, ndt::BatchLarge>), dim3(blocks), dim3(ndt::kBatchThreads), ndt::kBatchLdsBytes, st, a);
    else
      hipLaunchKernelGGL((ndt::k_batch<0, ndt::BatchLarge>), dim3(blocks), dim3(ndt::kBatchThreads), ndt::kBatchLdsBytes, st, a);
    HIP_TRY(hipGetLastError());
    // pairs whose grid does not fit on chip (handed over through fb_marks): tables in global memory
    if (newton)
      hipLaunchKernelGGL((ndt::k_batch_fallback<1>), dim3(blocks_fb), dim3(ndt::BatchGlobal::kThreads), ndt::BatchGlobal::kLdsBytes, st, a);
    else
      hipLaunchKernelGGL((ndt::k_batch_fallback<0>), dim3(blocks_fb), dim3(ndt::BatchGlobal::kThreads), ndt::BatchGlobal::kLdsBytes, st, a);
    HIP_TRY(hipGetLastError());
  }
  return NDT_OK;
}

template <typename T>
int32_t ensure_dev(T** p, size_t* cap, size_t n) {
  if (n <= *cap) return NDT_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  HIP_TRY(hipMalloc((void**)p, (n + n / 4 + 64) * sizeof(T)));
  *cap = n + n / 4 + 64;
  return NDT_OK;
}

}  // namespace

extern "C" {

// The library's standard coarse-to-fine schedule: 4c and 2c cells with a
// stronger eigenvalue clamp, loose stops and a proportionally larger step limit, then `fine`.
int32_t ndt2d_default_pyramid(const ndt2d_params* fine, ndt2d_params levels[3]) {
  if (!fine || !levels) return NDT_ERR_INVALID_ARG;
  const double mult[2] = {4.0, 2.0}, ratio[2] = {0.1, 0.03};
  for (int i = 0; i < 2; ++i) {
    ndt2d_params p = *fine;
    p.cell_size = fine->cell_size * mult[i];
    p.eig_ratio = ratio[i];
    p.eps_trans = 1e-3; p.eps_rot = 1e-4;
    p.max_iterations = 30; p.fixed_iterations = 0;
    p.step_max_trans = fine->step_max_trans * mult[i];
    levels[i] = p;
  }
  levels[2] = *fine;
  return NDT_OK;
}

int32_t ndt2d_batch_create_pyramid(const ndt2d_params* levels, int32_t n_levels, int32_t device_id, ndt2d_batch** out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = nullptr;
  if (!levels || n_levels < 1 || n_levels > 8) return NDT_ERR_INVALID_ARG;
  for (int32_t i = 0; i < n_levels; ++i) {
    const int32_t st = check_params(&levels[i]);
    if (st != NDT_OK) return st;
  }
  bool any_overlap = false;
  for (int32_t i = 0; i < n_levels; ++i) any_overlap = any_overlap || levels[i].overlap_grids == 4;
  const ndt2d_params* p = &levels[n_levels - 1];
  const int ndev = ndt_device_count();
  if (ndev <= 0) { set_error("no HIP device visible: this library has no CPU fallback"); return NDT_ERR_NO_DEVICE; }
  if (device_id < 0 || device_id >= ndev) return NDT_ERR_INVALID_ARG;
  ndt2d_batch* b = new (std::nothrow) ndt2d_batch();
  if (!b) return NDT_ERR_ALLOC;
  b->device = device_id;
  b->prm = *p;
  b->levels.assign(levels, levels + n_levels);
  auto fail = [&](int32_t code) { ndt2d_batch_destroy(b); return code; };
  if (hipSetDevice(device_id) != hipSuccess) return fail(NDT_ERR_HIP);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(NDT_ERR_HIP);
  b->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return fail(NDT_ERR_HIP);
  // a level with overlapping grids runs EVERY pair on the global-table variant: one table slab per CU from the start
  if (any_overlap) b->global_blocks = b->n_cu < ndt::kBatchGlobalBlocks ? b->n_cu : ndt::kBatchGlobalBlocks;
  if (hipMalloc((void**)&b->d_queue, 16) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&b->d_slab, (size_t)b->global_blocks * ndt::BatchGlobal::kTabBytes) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipHostMalloc((void**)&b->h_fb_seen, 64, hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  *b->h_fb_seen = 0;
  // more than 64 KiB of dynamic LDS needs an explicit opt-in per kernel
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch<0, ndt::BatchSmall>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::BatchSmall::kLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch<1, ndt::BatchSmall>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::BatchSmall::kLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::kBatchLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::kBatchLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch_fallback<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::BatchGlobal::kLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch_fallback<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::BatchGlobal::kLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch_fallback<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::BatchGlobal::kLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ndt::k_batch_fallback<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          ndt::BatchGlobal::kLdsBytes) != hipSuccess) return fail(NDT_ERR_HIP);
  *out = b;
  return NDT_OK;
}

int32_t ndt2d_batch_create(const ndt2d_params* p, int32_t device_id, ndt2d_batch** out) {
  if (!p) { if (out) *out = nullptr; return NDT_ERR_INVALID_ARG; }
  return ndt2d_batch_create_pyramid(p, 1, device_id, out);
}

int32_t ndt2d_batch_destroy(ndt2d_batch* b) {
  if (!b) return NDT_OK;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  void* dev[] = {b->d_slab, b->d_fb_list, b->d_queue, b->d_tx, b->d_ty, b->d_sx, b->d_sy, b->d_toff, b->d_soff, b->d_init, b->d_out, b->d_marks};
  for (void* p : dev) if (p) (void)hipFree(p);
  if (b->h_fb_seen) (void)hipHostFree(b->h_fb_seen);
  for (ndt2d_handle* f : b->fallback) ndt2d_destroy(f);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return NDT_OK;
}

int64_t ndt2d_batch_last_large_count(const ndt2d_batch* b) { return b ? b->last_large : -1; }

void* ndt2d_batch_stream(ndt2d_batch* b) { return b ? (void*)b->stream : nullptr; }

int32_t ndt2d_batch_set_tuning(ndt2d_batch* b, int32_t knob, int64_t value) {
  if (!b) return NDT_ERR_INVALID_ARG;
  if (knob != NDT_TUNE_BATCH_SMALL_VARIANT && knob != NDT_TUNE_BATCH_GLOBAL_WORKGROUPS) return NDT_ERR_INVALID_ARG;
  if (knob == NDT_TUNE_BATCH_GLOBAL_WORKGROUPS && (value < 1 || value > ndt::kBatchGlobalBlocksMax)) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (knob == NDT_TUNE_BATCH_SMALL_VARIANT) {
    b->use_small = value != 0;
  } else {
    if ((int)value != b->global_blocks) {            // one table slab per workgroup: re-allocate
      unsigned char* slab = nullptr;
      if (hipMalloc((void**)&slab, (size_t)value * ndt::BatchGlobal::kTabBytes) != hipSuccess) { (void)hipGetLastError(); return NDT_ERR_ALLOC; }
      (void)hipFree(b->d_slab);
      b->d_slab = slab;
      b->global_blocks = (int)value;
    }
    b->global_pinned = true;                         // the caller's number stands: no growth on demand
  }
  return NDT_OK;
}

int32_t ndt2d_batch_wait_stream(ndt2d_batch* b, void* producer_stream) {
  if (!b) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(ndt::order_after(b->stream, (hipStream_t)producer_stream));
  return NDT_OK;
}

int32_t ndt2d_batch_align_dev(ndt2d_batch* b, const float* d_tx, const float* d_ty, const uint64_t* d_toff,
                              const float* d_sx, const float* d_sy, const uint64_t* d_soff,
                              const double* d_init, size_t n_pairs, ndt2d_result* d_results, void* stream) {
  if (!b || !d_tx || !d_ty || !d_toff || !d_sx || !d_sy || !d_soff || !d_init || !d_results) return NDT_ERR_INVALID_ARG;
  if (n_pairs == 0 || n_pairs > 0x7fffffffull) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  return batch_launch(b, d_tx, d_ty, reinterpret_cast<const unsigned long long*>(d_toff), d_sx, d_sy,
                      reinterpret_cast<const unsigned long long*>(d_soff), d_init, n_pairs, d_results,
                      stream ? (hipStream_t)stream : b->stream);
}

int32_t ndt2d_batch_align(ndt2d_batch* b, const float* tx, const float* ty, const uint64_t* toff,
                          const float* sx, const float* sy, const uint64_t* soff, const double* init,
                          size_t n_pairs, ndt2d_result* results) {
  if (!b || !tx || !ty || !toff || !sx || !sy || !soff || !init || !results || n_pairs == 0) return NDT_ERR_INVALID_ARG;
  if (n_pairs > 0x7fffffffull) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(b->device));
  const size_t nt = toff[n_pairs], ns = soff[n_pairs];
  for (size_t k = 0; k < n_pairs; ++k) {      // a cloud of a pair is indexed with 32-bit byte offsets on the device
    if (toff[k + 1] < toff[k] || soff[k + 1] < soff[k] || toff[k + 1] - toff[k] > (size_t)ndt::kBatchMaxCloud ||
        soff[k + 1] - soff[k] > (size_t)ndt::kBatchMaxCloud) return NDT_ERR_INVALID_ARG;
  }
  int32_t st;
  if ((st = ensure_dev(&b->d_tx, &b->cap_tx, nt)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_ty, &b->cap_ty, nt)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_sx, &b->cap_sx, ns)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_sy, &b->cap_sy, ns)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_toff, &b->cap_toff, n_pairs + 1)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_soff, &b->cap_soff, n_pairs + 1)) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_init, &b->cap_init, 3 * (n_pairs + 1))) != NDT_OK) return st;
  if ((st = ensure_dev(&b->d_out, &b->cap_out, n_pairs + 1)) != NDT_OK) return st;
  hipStream_t s = b->stream;
  HIP_TRY(hipMemcpyAsync(b->d_tx, tx, nt * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_ty, ty, nt * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_sx, sx, ns * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_sy, sy, ns * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_toff, toff, (n_pairs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_soff, soff, (n_pairs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(b->d_init, init, 3 * n_pairs * sizeof(double), hipMemcpyHostToDevice, s));
  st = batch_launch(b, b->d_tx, b->d_ty, b->d_toff, b->d_sx, b->d_sy, b->d_soff, b->d_init, n_pairs, b->d_out, s);
  if (st != NDT_OK) return st;
  HIP_TRY(hipMemcpyAsync(results, b->d_out, n_pairs * sizeof(ndt2d_result), hipMemcpyDeviceToHost, s));
  std::vector<int> marks;
  const bool small_ran = b->use_small && b->levels.back().overlap_grids != 4;     // (an overlapping-grids level runs neither on-chip variant)
  if (small_ran) {
    marks.resize(n_pairs);
    HIP_TRY(hipMemcpyAsync(marks.data(), b->d_marks, n_pairs * sizeof(int), hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  b->last_large = small_ran ? 0 : (int64_t)n_pairs;
  for (int m : marks) b->last_large += m != 0;
  // pairs whose grid does not fit the on-chip capacity go through the global-memory path
  for (size_t k = 0; k < n_pairs; ++k) {
    if (results[k].status != NDT_ERR_CAPACITY) continue;
    if (b->fallback.empty()) {
      for (const ndt2d_params& lp : b->levels) {
        ndt2d_handle* f = nullptr;
        st = ndt2d_create(&lp, b->device, &f);
        if (st != NDT_OK) return st;
        b->fallback.push_back(f);
      }
    }
    double pose[3] = {init[3 * k], init[3 * k + 1], init[3 * k + 2]};
    int total = 0;
    for (ndt2d_handle* f : b->fallback) {
      st = ndt2d_set_target(f, tx + toff[k], ty + toff[k], toff[k + 1] - toff[k]);
      if (st == NDT_OK) st = ndt2d_align(f, sx + soff[k], sy + soff[k], soff[k + 1] - soff[k], pose, &results[k]);
      if (st < 0) break;
      total += results[k].iterations;
      results[k].iterations = total;
      if (results[k].status != NDT_OK && results[k].status != NDT_NOT_CONVERGED) break;
      for (int j = 0; j < 3; ++j) pose[j] = results[k].pose[j];
    }
    if (st < 0) return st;
  }
  return NDT_OK;
}

}  // extern "C"
