// C-ABI host side of the 2D NDT matcher (include/ndt_hip.h).  Owns device memory and one
// HIP stream per handle; every compute entry point launches the gfx950 kernels of
// ndt2d_kernels.hpp.  There is deliberately no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/ndt_hip.h"
#include "ndt2d_kernels.hpp"
#include "ndt2d_small.hpp"
#include "ndt2d_build.hpp"
#include "ndt2d_build_sorted.hpp"
#include "ndt2d_multi_start.hpp"
#include "ndt_host.hpp"

#include <atomic>

using namespace ndt;

struct ndt2d_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  ndt2d_params prm{};
  // target grid
  GridDev grid{};
  size_t cell_capacity = 0;
  bool has_target = false;
  int n_valid = 0;
  size_t n_points = 0;
  unsigned int* d_bounds = nullptr;        // [4]
  int* d_counters = nullptr;               // [2] valid cells, overflowed cells
  unsigned long long* d_outside = nullptr; // [1]
  void* h_small = nullptr;                 // pinned scratch (256 B: counter shards at 0, the outside count at 128)
  // staging for host-pointer entry points
  float* d_tx = nullptr; float* d_ty = nullptr; size_t tcap = 0;
  float* d_sx = nullptr; float* d_sy = nullptr; size_t scap = 0;
  // alignment context (see ndt2d_kernels.hpp: who writes what)
  AlignStatic* d_static = nullptr;
  AlignCall* d_call = nullptr;
  AlignDyn* d_dyn = nullptr;
  AlignStatic* h_static = nullptr;         // pinned
  hipEvent_t upload_ev = nullptr;          // recorded after the last upload from h_static
  hipEvent_t wait_ev = nullptr;            // ndt2d_wait_stream's event
  IterState* h_state = nullptr;            // pinned
  int* h_flag = nullptr;                   // pinned: [0] raised by the launch that ends a converged-mode loop, [1] progress
  int last_parity = 0;
  bool pending = false;
  // binned grid build scratch (ndt2d_build.hpp)
  float* d_bx = nullptr; float* d_by = nullptr; size_t bcap = 0;
  unsigned int* d_tiles = nullptr; size_t tile_cap = 0;   // total[nt] | start[nt+1] | cursor[nt]
  ndt::GeomDev* d_geom = nullptr;          // geometry decided on the device (one-round-trip ndt2d_set_target)
  ndt::GeomDev* h_geom = nullptr;          // pinned: its read-back
  int last_ntile = 0;                      // tiles of the cached grid: the next build's launch bound follows it
  bool static_on_device = false;           // d_static holds this handle's parameters (some upload_static has run)
  bool one_round_trip = true;              // NDT_TUNE_SINGLE_SYNC_BUILD
  bool use_binned_build = true;
  int build_variant = 1;                   // NDT_TUNE_BINNED_BUILD: 1 chunk-sorted build (ndt2d_build_sorted.hpp), 2 the round-1 binned build
  float2* d_bxy = nullptr; size_t bxy_cap = 0;             // chunk-sorted copy of the cloud ([chunks][chunk points])
  unsigned int* d_table = nullptr; size_t table_cap = 0;   // [tiles][chunks]: start | len << 16 of every run
  float4* d_parts = nullptr;               // [kBoundsParts]: per-workgroup partial bounding boxes
  unsigned char* d_split = nullptr; size_t split_cap = 0;  // tickets | cursor | touched | parts | pool of the shared tiles (SplitBufs)
  unsigned int build_seq = 0;              // sorted builds so far on this handle (SplitBufs::seq; never 0 in a launch)
  // accumulators of the sorted build, ping-pong: half p = 256 bytes: int counters[kCountInts] | u64 outside at 128.  A build
  // adds to half acc_parity and clears the other one for the next build (k_tile_gather), so no fill launch precedes it.
  unsigned char* d_acc2 = nullptr;
  int acc_parity = 0;
  bool acc_clean = false;                  // both halves known to be zero where the next build needs it
  int publish_seq = 0;                     // k_build_publish's flag value of the build in flight (h_small + 192)
  // hipGraph of the launch chain (launch-bound inner loop: one replay instead of K+1 launches)
  ChainGraphCache graphs;
  hipGraphExec_t graph_exec = nullptr;     // the one ensure_graph selected last (owned by `graphs`)
  AlignDynMulti* d_dyn_multi = nullptr;    // multi-start chains (ndt2d_multi_start.hpp), allocated on first use
  IterState* h_state_multi = nullptr;      // pinned [kMaxStarts]
  int split_from = 12;                     // multi-start / multi-scan calls of this many starts use the split chain (NDT_TUNE_SPLIT_FROM)
  ChunkRun chunk_run;                      // converged-mode loop begun by ndt2d_align_dev_async
  int call_seq = 0;                        // alignments enqueued so far (never 0 once one has run)
  bool wide = false;                       // this alignment's k_iterate launches use 1024-thread workgroups
  bool small_run = false;                  // a k_align_small launch whose flag has not been waited for
  // execution strategy knobs (ndt2d_set_tuning; results do not depend on them beyond float32 summation order)
  bool use_small = true;                   // short scans run the whole loop in one workgroup (k_align_small)
  bool use_wide = true;                    // 1024-thread workgroups for large scans ...
  size_t wide_threshold = 300000;          // ... from this many source points
  bool use_graph = true;
  int check_every = 8;                     // converged mode: launches per chunk
};

namespace {

constexpr size_t kMaxCells = (size_t)1 << 27;
// k_iterate indexes points with int and looks three strides (of up to 256 x 1024 threads) ahead
constexpr size_t kMaxSourcePoints = 0x7fffffffull - 4ull * kMaxBlocks * 1024ull;
#ifndef NDT_ITER_THREADS
#define NDT_ITER_THREADS 256
#endif
constexpr int kIterThreads = NDT_ITER_THREADS;   // workgroup size of k_iterate (256 workgroups always)
// Scans of a few hundred thousand points and more are no longer latency-bound per launch but short
// of loads in flight (15+ dependent gather rounds per thread at 256 threads): 1024-thread
// workgroups (four waves per SIMD) hide that latency.  Measured at 1M source points: see DESIGN 5.1.
constexpr int kIterThreadsWide = 1024;

int32_t check_params(const ndt2d_params* p) {
  if (!p) return NDT_ERR_INVALID_ARG;
  if (!(p->cell_size > 0.0) || !std::isfinite(p->cell_size)) return NDT_ERR_INVALID_ARG;
  if (p->min_points < 2 || p->max_iterations < 1 || p->fixed_iterations < 0) return NDT_ERR_INVALID_ARG;
  if (!(p->eig_ratio > 0.0) || !(p->eig_ratio <= 1.0)) return NDT_ERR_INVALID_ARG;
  if (p->hessian_mode != NDT_HESSIAN_GAUSS_NEWTON && p->hessian_mode != NDT_HESSIAN_NEWTON)
    return NDT_ERR_INVALID_ARG;
  if (!(p->step_max_trans > 0.0) || !(p->step_max_rot > 0.0)) return NDT_ERR_INVALID_ARG;
  if (!(p->d1 > 0.0) || !(p->d2 > 0.0) || !std::isfinite(p->d1) || !std::isfinite(p->d2)) return NDT_ERR_INVALID_ARG;
  if (p->overlap_grids != 0 && p->overlap_grids != 1 && p->overlap_grids != 4) return NDT_ERR_INVALID_ARG;
  if (p->line_search < 0 || p->line_search > 16) return NDT_ERR_INVALID_ARG;
  if (!(p->step_scale >= 0.0 && p->step_scale <= 8.0)) return NDT_ERR_INVALID_ARG;   // 0 means 1
  return NDT_OK;
}

int32_t ensure_points(float** dx, float** dy, size_t* cap, size_t n) {
  if (n <= *cap) return NDT_OK;
  if (*dx) (void)hipFree(*dx);
  if (*dy) (void)hipFree(*dy);
  *dx = *dy = nullptr; *cap = 0;
  const size_t want = n + n / 4 + 1024;
  HIP_TRY(hipMalloc((void**)dx, want * sizeof(float)));
  HIP_TRY(hipMalloc((void**)dy, want * sizeof(float)));
  *cap = want;
  return NDT_OK;
}

int blocks_for(size_t) { return kMaxBlocks; }   // one workgroup per CU, always (see k_iterate)

// k_bounds ends in four same-address atomics per block (about 10 ns each, serialised): few blocks
#ifndef NDT_BOUNDS_BLOCKS
#define NDT_BOUNDS_BLOCKS 128
#endif
constexpr int kBoundsBlocks = NDT_BOUNDS_BLOCKS;
int stream_blocks(size_t n) {   // streaming kernels: up to 8 blocks per CU
  size_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

int32_t upload_static(ndt2d_handle* h);

int32_t finalise_grid(ndt2d_handle* h) {
  const size_t ncell = (size_t)h->grid.W * h->grid.H * h->grid.ngrid;
  HIP_TRY(hipMemsetAsync(h->d_counters, 0, ndt::kCountInts * sizeof(int), h->stream));
  hipLaunchKernelGGL(k_finalise, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     h->stream, h->grid, h->prm.min_points, h->prm.eig_ratio, h->d_counters);
  HIP_TRY(hipGetLastError());
  int* hc = (int*)h->h_small;
  HIP_TRY(hipMemcpyAsync(hc, h->d_counters, ndt::kCountInts * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  int n_valid_sum = 0, n_over_sum = 0;
  sum_count_shards(hc, &n_valid_sum, &n_over_sum);
  h->n_valid = n_valid_sum;
  if (n_over_sum > 0) { set_error("a target cell holds more than 2^20 points"); return NDT_ERR_CAPACITY; }
  return NDT_OK;
}

// ---- chunk-sorted build (ndt2d_build_sorted.hpp): host side -------------------------------------------------------
struct SortPlan { int P = 0, chunk = 0, nchunks = 0; };

// Points per thread of k_chunk_sort (chunk = 256 P points): small clouds take small chunks so that more CUs share
// the work (a 100k-point scan: 98 chunks of 1024), large ones the longest runs.  false: the table would exceed its
// bounds (tiles x chunks <= 2^20, chunks <= 4096) - the caller takes the round-1 path.
bool plan_sorted(size_t n, long long ntile, SortPlan* sp) {
  int P = n <= 131072 ? 4 : (n <= 524288 ? 8 : 16);
  for (;; P *= 2) {
    const size_t C = (size_t)kSortThreads * P, nch = (n + C - 1) / C;
    if (nch <= (size_t)kSortMaxChunks && nch * (size_t)ntile <= kSortMaxTable) { sp->P = P; sp->chunk = (int)C; sp->nchunks = (int)nch; return true; }
    if (P == 16) return false;
  }
}

int32_t ensure_sorted_buffers(ndt2d_handle* h, const SortPlan& sp, size_t ntile) {
  const size_t need_pts = (size_t)sp.nchunks * sp.chunk, need_tab = ntile * (size_t)sp.nchunks;
  if (need_pts > h->bxy_cap) {
    if (h->d_bxy) (void)hipFree(h->d_bxy);
    h->d_bxy = nullptr; h->bxy_cap = 0;
    const size_t want = need_pts + need_pts / 4 + 4096;
    HIP_TRY(hipMalloc((void**)&h->d_bxy, want * sizeof(float2)));
    h->bxy_cap = want;
  }
  if (need_tab > h->table_cap) {
    if (h->d_table) (void)hipFree(h->d_table);
    h->d_table = nullptr; h->table_cap = 0;
    const size_t want = need_tab + need_tab / 4 + 1024;
    HIP_TRY(hipMalloc((void**)&h->d_table, want * sizeof(unsigned int)));
    h->table_cap = want;
  }
  return NDT_OK;
}

// the buffers the workgroups that share a tile use (ndt2d_build_sorted.hpp: SplitBufs), sized for `tiles` tiles and n points
int32_t ensure_split_buffers(ndt2d_handle* h, size_t n, int tiles, SplitBufs* sb) {
  const size_t pool_entries = n < (size_t)tiles * kGatherSplit * kTileCells ? n : (size_t)tiles * kGatherSplit * kTileCells;
  const size_t off_touched = ((size_t)tiles + 1) * sizeof(unsigned int);
  const size_t off_part = (off_touched + (size_t)tiles * sizeof(unsigned int) + 15) / 16 * 16;
  const size_t off_pool = off_part + (size_t)tiles * kGatherSplit * sizeof(uint2);
  const size_t need = off_pool + (pool_entries + 1) * sizeof(CellAcc);
  if (need > h->split_cap) {
    if (h->d_split) (void)hipFree(h->d_split);
    h->d_split = nullptr; h->split_cap = 0;
    const size_t want = need + need / 4;
    HIP_TRY(hipMalloc((void**)&h->d_split, want));
    HIP_TRY(hipMemsetAsync(h->d_split, 0, want, h->stream));       // (the touched numbers must not start as garbage)
    h->split_cap = want;
    h->build_seq = 0;
  }
  if (++h->build_seq == 0u) {                                      // wrapped: start the numbers over
    HIP_TRY(hipMemsetAsync(h->d_split, 0, h->split_cap, h->stream));
    h->build_seq = 1u;
  }
  sb->ticket = reinterpret_cast<unsigned int*>(h->d_split);
  sb->touched = reinterpret_cast<unsigned int*>(h->d_split + off_touched);
  sb->part = reinterpret_cast<uint2*>(h->d_split + off_part);
  sb->pool = reinterpret_cast<CellAcc*>(h->d_split + off_pool);
  sb->tiles = tiles;
  sb->seq = h->build_seq;
  // a tile is shared so that no workgroup sums much more than a CU's share of the cloud; clouds too small to fill the
  // chip are cut finer, down to 1024 points per workgroup
  size_t sp = n / 200;
  sb->split_points = (int)(sp < 1024 ? 1024 : (sp > (size_t)1 << 24 ? (size_t)1 << 24 : sp));
  return NDT_OK;
}

// Workgroups per tile (grid.y of k_tile_gather).  A cloud that fills most of the grid's tiles keeps most CUs busy with one
// workgroup per tile (169 tiles of a 1M-point submap: sharing them measured slower - the surplus workgroups cost more
// than the 87 idle CUs could give back); a cloud that lands on a few tiles (a 100k-point scan touches about 16) is
// where sharing pays: its tiles get up to kGatherSplit workgroups each.
int gather_split(size_t n, int tiles) {
  size_t touched = n / 6000 + 1;
  if (touched > (size_t)tiles) touched = (size_t)tiles;
  size_t s = 256 / touched;
  return (int)(s < 1 ? 1 : (s > (size_t)kGatherSplit ? (size_t)kGatherSplit : s));
}

void launch_chunk_sort(ndt2d_handle* h, const SortPlan& sp, const float* d_x, const float* d_y, size_t n, const BinGeom& bg,
                       int hist_tiles, const MoveArgs& mv, unsigned long long* d_outside, const GeomArgs& ga, const SplitBufs& sb) {
  const size_t lds = (size_t)sp.chunk * sizeof(float2) + (size_t)hist_tiles * sizeof(unsigned int);
#define NDT_SORT(PP) hipLaunchKernelGGL((k_chunk_sort<PP>), dim3((unsigned)sp.nchunks), dim3(kSortThreads), lds, h->stream, d_x, d_y, n, \
                                        bg, sp.nchunks, mv, h->d_bxy, h->d_table, d_outside, ga, sb.ticket, sb.tiles + 1, sb.touched, sb.seq)
  if (sp.P == 4) NDT_SORT(4); else if (sp.P == 8) NDT_SORT(8); else NDT_SORT(16);
#undef NDT_SORT
}

// the bounding box of a cloud as per-workgroup partials in h->d_parts; returns the number of partials
int launch_bounds_parts(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n, GeomDev* zero = nullptr) {
  int nb = stream_blocks(n);
  if (nb > kBoundsParts) nb = kBoundsParts;
  const bool vec = (((uintptr_t)d_x | (uintptr_t)d_y) & 15u) == 0;
  if (vec) hipLaunchKernelGGL((k_bounds_parts<true>), dim3(nb), dim3(kBlock), 0, h->stream, d_x, d_y, n, h->d_parts, zero);
  else hipLaunchKernelGGL((k_bounds_parts<false>), dim3(nb), dim3(kBlock), 0, h->stream, d_x, d_y, n, h->d_parts, zero);
  return nb;
}

// The read-back of a sorted build: k_build_publish writes `nwords` words from device memory into pinned host memory and
// raises the flag at h_small + 192; the host spins on it.  If the flag does not come (a stream error, or a second of
// silence), a plain copy after a stream synchronisation is the safety net.
int32_t publish_and_wait(ndt2d_handle* h, const void* d_src, void* h_dst, int nwords) {
  int* flag = reinterpret_cast<int*>(static_cast<char*>(h->h_small) + 192);
  h->publish_seq = h->publish_seq == 0x7fffffff ? 1 : h->publish_seq + 1;
  hipLaunchKernelGGL(k_build_publish, dim3(1), dim3(64), 0, h->stream, (const unsigned int*)d_src, (unsigned int*)h_dst, nwords, flag,
                     h->publish_seq);
  HIP_TRY(hipGetLastError());
  bool seen = false;
  const int want = h->publish_seq;
  HIP_TRY(spin_until(h->stream, [&]() { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want; }, &seen));
  if (!seen) {
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)nwords * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return NDT_OK;
}

// a2 + a3 for n points: binned LDS build when the tile histogram fits in LDS (always, below
// ~2.9 km x 2.9 km at 0.5 m cells), else scattered global atomics + k_finalise.  merge = add to
// the cached sums (incremental submap update) instead of starting from zero.
int32_t accumulate_and_finalise(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n, bool merge,
                                unsigned long long* h_outside, const MoveArgs* move = nullptr) {
  TraceRange range(merge ? "ndt2d: submap update (moments + finalise)" : "ndt2d: moments + finalise");
  GridDev& g = h->grid;
  const size_t ncell1 = (size_t)g.W * g.H, ncell = ncell1 * g.ngrid;
  const int ntx = (g.W + kTile - 1) >> kTileShift, nty = (g.H + kTile - 1) >> kTileShift;
  const long long ntile_ll = (long long)ntx * nty;
  SortPlan sp;
  if (h->use_binned_build && h->build_variant == 1 && ntile_ll <= kBinMaxTiles && plan_sorted(n, ntile_ll, &sp)) {
    // chunk-sorted build: sort every chunk of the cloud by tile, then one workgroup per tile gathers its runs
    const int ntile = (int)ntile_ll;
    { const int32_t es = ensure_sorted_buffers(h, sp, (size_t)ntile); if (es != NDT_OK) return es; }
    SplitBufs sb{};
    { const int32_t es = ensure_split_buffers(h, n, ntile, &sb); if (es != NDT_OK) return es; }
    const int split = gather_split(n, ntile);
    if (!h->acc_clean) HIP_TRY(hipMemsetAsync(h->d_acc2, 0, 512, h->stream));      // (first build, or one that failed half-way)
    h->acc_clean = false;
    unsigned char* cur = h->d_acc2 + 256 * h->acc_parity;
    unsigned char* nxt = h->d_acc2 + 256 * (1 - h->acc_parity);
    int* d_cnt = reinterpret_cast<int*>(cur);
    unsigned long long* d_out = reinterpret_cast<unsigned long long*>(cur + 128);
    const MoveArgs none{1.f, 0.f, 0.f, 0.f, 0};
    for (int q = 0; q < g.ngrid; ++q) {
      BinGeom bg{g.gx[q], g.gy[q], g.inv_c, g.W, g.H, ntx, ntile};
      launch_chunk_sort(h, sp, d_x, d_y, n, bg, ntile, move ? *move : none, q == 0 ? d_out : (unsigned long long*)nullptr,
                        GeomArgs{}, sb);
      hipLaunchKernelGGL(k_tile_gather, dim3(ntile, split), dim3(kGatherThreads), 0, h->stream, (const float2*)h->d_bxy, (const unsigned int*)h->d_table,
                         sp.nchunks, sp.chunk, g, q, ntx, merge ? 1 : 0, h->prm.min_points, h->prm.eig_ratio, d_cnt,
                         (const GeomDev*)nullptr, (const GridDev*)nullptr, sb,
                         q == g.ngrid - 1 ? reinterpret_cast<unsigned int*>(nxt) : (unsigned int*)nullptr);
      HIP_TRY(hipGetLastError());
    }
    h->last_ntile = ntile;
    int* hc = (int*)h->h_small;
    unsigned long long* ho = (unsigned long long*)((char*)h->h_small + 128);
    static_assert(ndt::kCountInts * sizeof(int) == 128, "the accumulator halves mirror h_small: counters at 0, outside at 128");
    { const int32_t ps = publish_and_wait(h, cur, hc, 34); if (ps != NDT_OK) return ps; }   // counter shards + outside count
    h->acc_parity ^= 1;
    h->acc_clean = true;
    int n_valid_sum = 0, n_over_sum = 0;
    sum_count_shards(hc, &n_valid_sum, &n_over_sum);
    h->n_valid = merge ? h->n_valid + n_valid_sum : n_valid_sum;       // merge: the gather kernel counts the change, tile by touched tile
    if (h_outside) *h_outside = *ho;
    if (n_over_sum > 0) { set_error("a target cell holds more than 2^20 points"); return NDT_ERR_CAPACITY; }
    return NDT_OK;
  }
  // the other paths take the points as they are: move them into the map frame first
  if (move && move->use) {
    const int32_t st = ensure_points(&h->d_tx, &h->d_ty, &h->tcap, n);
    if (st != NDT_OK) return st;
    hipLaunchKernelGGL(k_transform_points, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, d_x, d_y,
                       n, move->cs, move->sn, move->tx, move->ty, h->d_tx, h->d_ty);
    HIP_TRY(hipGetLastError());
    d_x = h->d_tx; d_y = h->d_ty;
  }
  HIP_TRY(hipMemsetAsync(h->d_outside, 0, sizeof(unsigned long long), h->stream));
  if (h->use_binned_build && ntile_ll <= kBinMaxTiles && n <= 0xFFFFFFFFull) {
    const int ntile = (int)ntile_ll;
    if (n > h->bcap) {
      if (h->d_bx) (void)hipFree(h->d_bx);
      if (h->d_by) (void)hipFree(h->d_by);
      h->d_bx = h->d_by = nullptr; h->bcap = 0;
      const size_t want = n + n / 4 + 1024;
      HIP_TRY(hipMalloc((void**)&h->d_bx, want * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&h->d_by, want * sizeof(float)));
      h->bcap = want;
    }
    const size_t tneed = 3 * (size_t)ntile + 4;
    if (tneed > h->tile_cap) {
      if (h->d_tiles) (void)hipFree(h->d_tiles);
      h->d_tiles = nullptr; h->tile_cap = 0;
      HIP_TRY(hipMalloc((void**)&h->d_tiles, tneed * sizeof(unsigned int)));
      h->tile_cap = tneed;
    }
    unsigned int* d_total = h->d_tiles;
    unsigned int* d_start = h->d_tiles + ntile;
    unsigned int* d_cursor = h->d_tiles + 2 * ntile + 1;
    HIP_TRY(hipMemsetAsync(h->d_counters, 0, ndt::kCountInts * sizeof(int), h->stream));
    const size_t chunk = (size_t)kBinThreads * kBinPerThread;
    size_t nb = (n + chunk - 1) / chunk;
    if (nb > 1024) nb = 1024;
    for (int q = 0; q < g.ngrid; ++q) {
      BinGeom bg{g.gx[q], g.gy[q], g.inv_c, g.W, g.H, ntx, ntile};
      HIP_TRY(hipMemsetAsync(d_total, 0, ntile * sizeof(unsigned int), h->stream));
      hipLaunchKernelGGL(k_tile_count, dim3((unsigned)nb), dim3(kBinThreads), ntile * sizeof(unsigned int), h->stream,
                         d_x, d_y, n, bg, d_total, q == 0 ? h->d_outside : (unsigned long long*)nullptr, (const GeomDev*)nullptr);
      hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, h->stream, d_total, d_start, d_cursor, ntile, (const GeomDev*)nullptr);
      hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)nb), dim3(kBinThreads), 2 * ntile * sizeof(unsigned int),
                         h->stream, d_x, d_y, n, bg, d_cursor, h->d_bx, h->d_by, (const GeomDev*)nullptr);
      hipLaunchKernelGGL(k_tile_accumulate, dim3(ntile), dim3(kBinThreads), 0, h->stream, h->d_bx, h->d_by, d_start, g, q,
                         ntx, merge ? 1 : 0, h->prm.min_points, h->prm.eig_ratio, h->d_counters, (const GeomDev*)nullptr,
                         (const GridDev*)nullptr);
      HIP_TRY(hipGetLastError());
    }
    h->last_ntile = ntile;
    int* hc = (int*)h->h_small;
    unsigned long long* ho = (unsigned long long*)((char*)h->h_small + 128);
    HIP_TRY(hipMemcpyAsync(hc, h->d_counters, ndt::kCountInts * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ho, h->d_outside, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int n_valid_sum = 0, n_over_sum = 0;
    sum_count_shards(hc, &n_valid_sum, &n_over_sum);
    h->n_valid = n_valid_sum;
    if (h_outside) *h_outside = *ho;
    if (n_over_sum > 0) { set_error("a target cell holds more than 2^20 points"); return NDT_ERR_CAPACITY; }
    return NDT_OK;
  }
  // fallback: scattered global atomics
  if (!merge) HIP_TRY(hipMemsetAsync(g.acc, 0, ncell * sizeof(CellAcc), h->stream));
  hipLaunchKernelGGL(k_accumulate, dim3(stream_blocks(n)), dim3(kBlock), 0, h->stream, d_x, d_y, n, g, h->d_outside);
  HIP_TRY(hipGetLastError());
  unsigned long long* ho = (unsigned long long*)((char*)h->h_small + 128);
  HIP_TRY(hipMemcpyAsync(ho, h->d_outside, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  const int32_t st = finalise_grid(h);
  if (h_outside) *h_outside = *ho;
  return st;
}

// Grid geometry for a bounding box (oracle/ndt2d.py grid_geometry) and storage for its cells.
int32_t setup_geometry(ndt2d_handle* h, float xmin, float xmax, float ymin, float ymax) {
  const double c = h->prm.cell_size;
  GridDev& g = h->grid;
  g.cell = c;
  g.cell32 = (float)c;
  g.inv_c = (float)(1.0 / c);
  g.ox = (float)((std::floor((double)xmin / c) - 1.0) * c);
  g.oy = (float)((std::floor((double)ymin / c) - 1.0) * c);
  const volatile float fx = (xmax - g.ox) * g.inv_c;   // float32 arithmetic, as the kernels
  const volatile float fy = (ymax - g.oy) * g.inv_c;
  const double kx = std::floor((double)fx), ky = std::floor((double)fy);
  if (!(kx >= 0.0) || !(ky >= 0.0) || (kx + 2.0) * (ky + 2.0) > (double)kMaxCells) {
    set_error("target extent / cell_size needs more than 2^27 cells");
    return NDT_ERR_CAPACITY;
  }
  // overlapping grids: origins move down by half a cell, one extra column/row covers the maximum
  g.ngrid = h->prm.overlap_grids == 4 ? 4 : 1;
  g.pad = 0;
  const int extra = g.ngrid > 1 ? 1 : 0;
  g.W = (int)kx + 2 + extra;
  g.H = (int)ky + 2 + extra;
  static const double kShift[4][2] = {{0.0, 0.0}, {0.5, 0.0}, {0.0, 0.5}, {0.5, 0.5}};
  for (int q = 0; q < kMaxGrids; ++q) {
    g.gx[q] = (float)((std::floor((double)xmin / c) - 1.0 - kShift[q][0]) * c);
    g.gy[q] = (float)((std::floor((double)ymin / c) - 1.0 - kShift[q][1]) * c);
  }
  g.fix_scale = std::ldexp(1.0, kFixShift) / c;
  const size_t ncell = (size_t)g.W * g.H * g.ngrid;     // all grids, back to back
  if (ncell > kMaxCells) { set_error("target extent / cell_size needs more than 2^27 cells"); return NDT_ERR_CAPACITY; }
  if (ncell > h->cell_capacity) {
    if (g.rec) (void)hipFree(g.rec);
    if (g.acc) (void)hipFree(g.acc);
    g.rec = nullptr; g.acc = nullptr; h->cell_capacity = 0;
    const size_t want = ncell + ncell / 8;
    HIP_TRY(hipMalloc((void**)&g.rec, 2 * want * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&g.acc, want * sizeof(CellAcc)));
    h->cell_capacity = want;
  }
  return NDT_OK;
}

// ndt2d_set_target with ONE host round trip: bounds -> geometry (k_geometry, on the device, into d_static) ->
// binned build, enqueued back to back; the host learns the geometry together with the counters.  Possible
// when the handle already holds storage and parameters on the device (any earlier target) and the new grid
// fits them and the launch bound chosen here; otherwise *done = false (with the bounding box in hb_out when
// the device got that far) and the caller builds the usual way.
int32_t set_target_single_sync(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n, bool* done, unsigned int* hb_out,
                               bool* have_bounds) {
  *done = false; *have_bounds = false;
  if (!h->one_round_trip || !h->use_binned_build || h->prm.overlap_grids == 4 || h->cell_capacity == 0 || !h->static_on_device ||
      h->last_ntile <= 0 || n > 0xFFFFFFFFull || !h->grid.rec || !h->grid.acc)
    return NDT_OK;
  long long tb = 2ll * h->last_ntile + 16;
  if (tb > kBinMaxTiles) tb = kBinMaxTiles;
  const int tile_bound = (int)tb;
  if (!h->d_geom) {
    HIP_TRY(hipMalloc((void**)&h->d_geom, sizeof(GeomDev)));
    HIP_TRY(hipHostMalloc((void**)&h->h_geom, sizeof(GeomDev), hipHostMallocDefault));
  }
  GeomDev* dg = h->d_geom;
  SortPlan sp;
  if (h->build_variant == 1 && plan_sorted(n, tile_bound, &sp)) {
    // chunk-sorted build: bounds partials -> chunk sort (reduce + geometry in its prologue) -> one workgroup per tile
    { const int32_t es = ensure_sorted_buffers(h, sp, (size_t)tile_bound); if (es != NDT_OK) return es; }
    SplitBufs sb{};
    { const int32_t es = ensure_split_buffers(h, n, tile_bound, &sb); if (es != NDT_OK) return es; }
    const int split = gather_split(n, tile_bound);
    const int nparts = launch_bounds_parts(h, d_x, d_y, n, dg);
    const BinGeom none{};
    const MoveArgs stay{1.f, 0.f, 0.f, 0.f, 0};
    GeomArgs ga{};
    ga.parts = h->d_parts; ga.nparts = nparts; ga.tile_bound = tile_bound; ga.cell = h->prm.cell_size;
    ga.cell_capacity = (unsigned long long)h->cell_capacity; ga.grid = &h->d_static->grid; ga.out = dg;
    launch_chunk_sort(h, sp, d_x, d_y, n, none, tile_bound, stay, &dg->n_outside, ga, sb);
    hipLaunchKernelGGL(k_tile_gather, dim3(tile_bound, split), dim3(kGatherThreads), 0, h->stream, (const float2*)h->d_bxy,
                       (const unsigned int*)h->d_table, sp.nchunks, sp.chunk, h->grid, 0, 0, 0, h->prm.min_points, h->prm.eig_ratio,
                       &dg->counters[0], (const GeomDev*)dg, (const GridDev*)&h->d_static->grid, sb, (unsigned int*)nullptr);
  } else {
  if (n > h->bcap) {
    if (h->d_bx) (void)hipFree(h->d_bx);
    if (h->d_by) (void)hipFree(h->d_by);
    h->d_bx = h->d_by = nullptr; h->bcap = 0;
    const size_t want = n + n / 4 + 1024;
    HIP_TRY(hipMalloc((void**)&h->d_bx, want * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&h->d_by, want * sizeof(float)));
    h->bcap = want;
  }
  const size_t tneed = 3 * (size_t)tile_bound + 4;
  if (tneed > h->tile_cap) {
    if (h->d_tiles) (void)hipFree(h->d_tiles);
    h->d_tiles = nullptr; h->tile_cap = 0;
    HIP_TRY(hipMalloc((void**)&h->d_tiles, tneed * sizeof(unsigned int)));
    h->tile_cap = tneed;
  }
  // tile tables laid out for the bound: total[tb] | start[tb+1] | cursor[tb]
  unsigned int* d_total = h->d_tiles;
  unsigned int* d_start = h->d_tiles + tile_bound;
  unsigned int* d_cursor = h->d_tiles + 2 * tile_bound + 1;
  hipLaunchKernelGGL(k_build_init, dim3(1), dim3(1024), 0, h->stream, dg, d_total, tile_bound);
  hipLaunchKernelGGL(k_bounds, dim3(stream_blocks(n) > kBoundsBlocks ? kBoundsBlocks : stream_blocks(n)), dim3(kBlock), 0, h->stream,
                     d_x, d_y, n, &dg->bounds[0]);
  hipLaunchKernelGGL(k_geometry, dim3(1), dim3(64), 0, h->stream, h->prm.cell_size, (unsigned long long)h->cell_capacity, tile_bound,
                     &h->d_static->grid, dg);
  const size_t chunk = (size_t)kBinThreads * kBinPerThread;
  size_t nb = (n + chunk - 1) / chunk;
  if (nb > 1024) nb = 1024;
  const BinGeom none{};
  hipLaunchKernelGGL(k_tile_count, dim3((unsigned)nb), dim3(kBinThreads), tile_bound * sizeof(unsigned int), h->stream, d_x, d_y, n,
                     none, d_total, &dg->n_outside, (const GeomDev*)dg);
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, h->stream, d_total, d_start, d_cursor, 0, (const GeomDev*)dg);
  hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)nb), dim3(kBinThreads), 2 * tile_bound * sizeof(unsigned int), h->stream, d_x,
                     d_y, n, none, d_cursor, h->d_bx, h->d_by, (const GeomDev*)dg);
  hipLaunchKernelGGL(k_tile_accumulate, dim3(tile_bound), dim3(kBinThreads), 0, h->stream, h->d_bx, h->d_by, d_start, h->grid, 0, 0,
                     0, h->prm.min_points, h->prm.eig_ratio, &dg->counters[0], (const GeomDev*)dg,
                     (const GridDev*)&h->d_static->grid);
  }
  HIP_TRY(hipGetLastError());
  GeomDev* hg = h->h_geom;
  static_assert(sizeof(GeomDev) % 4 == 0, "GeomDev travels to the host word by word");
  { const int32_t ps = publish_and_wait(h, dg, hg, (int)(sizeof(GeomDev) / 4)); if (ps != NDT_OK) return ps; }
  const int* hc = hg->counters;
  for (int j = 0; j < 4; ++j) hb_out[j] = hg->bounds[j];
  *have_bounds = true;
  if (!hg->ok) return NDT_OK;                            // does not fit storage or bound (or no finite point): the usual way
  // the host's view of the same geometry, from the same bounds; storage is large enough, nothing is reallocated
  const int32_t gs = setup_geometry(h, ordered_to_float(hg->bounds[0]), ordered_to_float(hg->bounds[1]),
                                    ordered_to_float(hg->bounds[2]), ordered_to_float(hg->bounds[3]));
  if (gs != NDT_OK) return NDT_OK;
  if (h->grid.W != hg->bin.W || h->grid.H != hg->bin.H || h->grid.ox != hg->bin.ox || h->grid.oy != hg->bin.oy) return NDT_OK;
  h->h_static->grid = h->grid;                           // what d_static holds already
  h->last_ntile = hg->bin.ntile;
  int n_valid_sum = 0, n_over_sum = 0;
  sum_count_shards(hc, &n_valid_sum, &n_over_sum);
  h->n_valid = n_valid_sum;
  if (n_over_sum > 0) { set_error("a target cell holds more than 2^20 points"); return NDT_ERR_CAPACITY; }
  *done = true;
  return NDT_OK;
}

int32_t set_target_impl(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n) {
  TraceRange range("ndt2d_set_target: grid build");
  h->has_target = false;
  if (n == 0) return NDT_ERR_INVALID_ARG;
  unsigned int fast_bounds[4];
  bool done = false, have_bounds = false;
  { const int32_t fs = set_target_single_sync(h, d_x, d_y, n, &done, fast_bounds, &have_bounds); if (fs != NDT_OK) return fs; }
  if (done) {
    h->n_points = n;
    h->has_target = true;
    return NDT_OK;
  }
  // a1: bounding box on the device, geometry on the host (oracle/ndt2d.py grid_geometry)
  unsigned int init[4] = {0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u};
  unsigned int* hb = (unsigned int*)h->h_small;
  if (have_bounds) {                                      // the single-sync attempt measured the box already
    std::memcpy(hb, fast_bounds, sizeof(init));
  } else {
    std::memcpy(hb, init, sizeof(init));
    {
      const int nparts = launch_bounds_parts(h, d_x, d_y, n);
      hipLaunchKernelGGL(k_bounds_reduce, dim3(1), dim3(64), 0, h->stream, (const float4*)h->d_parts, nparts, h->d_bounds);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hb, h->d_bounds, sizeof(init), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if (hb[0] == 0xFFFFFFFFu || hb[1] == 0u) { set_error("target has no finite point"); return NDT_ERR_INVALID_ARG; }
  const int32_t gs = setup_geometry(h, ordered_to_float(hb[0]), ordered_to_float(hb[1]), ordered_to_float(hb[2]),
                                    ordered_to_float(hb[3]));
  if (gs != NDT_OK) return gs;
  // a2 + a3
  const int32_t st = accumulate_and_finalise(h, d_x, d_y, n, /*merge=*/false, nullptr);
  if (st != NDT_OK) return st;
  h->n_points = n;
  h->has_target = true;
  return upload_static(h);
}

// Static part of the device context: grid + solver parameters.  Uploaded (synchronously)
// whenever the target changes; the per-call part is written by k_begin from kernel arguments,
// so no host buffer has to outlive an asynchronous call.
int32_t upload_static(ndt2d_handle* h) {
  // The copy is left in flight (everything that reads d_static is ordered behind it on the same
  // stream); the pinned source is only rewritten once the previous copy has left it.
  HIP_TRY(hipEventSynchronize(h->upload_ev));
  AlignStatic* c = h->h_static;
  c->grid = h->grid;
  SolveParams& p = c->prm;
  p.d1 = (float)h->prm.d1;
  p.d2 = (float)h->prm.d2;
  p.hessian_mode = h->prm.hessian_mode;
  p.max_iterations = h->prm.max_iterations;
  p.min_hits = h->prm.min_hits;
  p.line_search = h->prm.line_search;
  p.eps_trans = h->prm.eps_trans;
  p.eps_rot = h->prm.eps_rot;
  p.step_max_trans = h->prm.step_max_trans;
  p.step_max_rot = h->prm.step_max_rot;
  p.step_scale = h->prm.step_scale > 0.0 ? h->prm.step_scale : 1.0;
  HIP_TRY(hipMemcpyAsync(h->d_static, c, sizeof(AlignStatic), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipEventRecord(h->upload_ev, h->stream));
  h->static_on_device = true;
  return NDT_OK;
}

void launch_iter(ndt2d_handle* h, int blocks, int k) {
  const bool newton = h->prm.hessian_mode == NDT_HESSIAN_NEWTON, four = h->prm.overlap_grids == 4;
#define NDT_LAUNCH_ITER(MODE, NG)                                                                                       \
  do {                                                                                                                  \
    if (h->wide)                                                                                                        \
      hipLaunchKernelGGL((k_iterate<MODE, 0, kIterThreadsWide, NG>), dim3(blocks), dim3(kIterThreadsWide), 0, h->stream, \
                         h->d_static, h->d_call, h->d_dyn, k & 1);                                                      \
    else                                                                                                                \
      hipLaunchKernelGGL((k_iterate<MODE, 0, kIterThreads, NG>), dim3(blocks), dim3(kIterThreads), 0, h->stream,        \
                         h->d_static, h->d_call, h->d_dyn, k & 1);                                                      \
  } while (0)
  if (newton && four) NDT_LAUNCH_ITER(1, 4);
  else if (newton) NDT_LAUNCH_ITER(1, 1);
  else if (four) NDT_LAUNCH_ITER(0, 4);
  else NDT_LAUNCH_ITER(0, 1);
#undef NDT_LAUNCH_ITER
}

void drop_graph(ndt2d_handle* h) {
  h->graphs.clear();
  h->graph_exec = nullptr;
}

// Graph of `launches` consecutive k_iterate launches starting at parity 0.  The kernels read
// everything (grid, source pointers, n, parameters, state) from device memory, so one graph
// serves every target and every source.
int32_t ensure_graph(ndt2d_handle* h, int launches, int blocks) {
  const bool newton = h->prm.hessian_mode == NDT_HESSIAN_NEWTON, four = h->prm.overlap_grids == 4;
  const void* func;
  if (h->wide)
    func = newton ? (four ? (const void*)&k_iterate<1, 0, kIterThreadsWide, 4> : (const void*)&k_iterate<1, 0, kIterThreadsWide, 1>)
                  : (four ? (const void*)&k_iterate<0, 0, kIterThreadsWide, 4> : (const void*)&k_iterate<0, 0, kIterThreadsWide, 1>);
  else
    func = newton ? (four ? (const void*)&k_iterate<1, 0, kIterThreads, 4> : (const void*)&k_iterate<1, 0, kIterThreads, 1>)
                  : (four ? (const void*)&k_iterate<0, 0, kIterThreads, 4> : (const void*)&k_iterate<0, 0, kIterThreads, 1>);
  HIP_TRY(h->graphs.get(func, dim3(blocks), dim3(h->wide ? kIterThreadsWide : kIterThreads), (void*)h->d_static, (void*)h->d_call,
                        (void*)h->d_dyn, launches, h->prm.hessian_mode | (h->wide ? 16 : 0), h->stream, &h->graph_exec));
  return NDT_OK;
}

// Enqueue the Gauss-Newton loop.  check_every > 0: poll the done flag every that many
// launches (synchronous early exit); 0: enqueue all launches, finished ones are no-ops.
// Wait for a single-workgroup alignment: its last thread raises the flag in pinned host memory
// after writing the state there, so the result is on the host the moment the spin ends.
int32_t ensure_multi_buffers(ndt2d_handle* h) {
  if (!h->h_state_multi) HIP_TRY(hipHostMalloc((void**)&h->h_state_multi, kMaxStarts * sizeof(IterState), hipHostMallocDefault));
  return NDT_OK;
}

int32_t finish_small_run(ndt2d_handle* h) {
  if (!h->small_run) return NDT_OK;
  h->small_run = false;
  // No stream sync once the flag is up: the kernel read the scan into registers at its start and
  // raises the flag as its last action, so the caller's source buffers are already free.
  bool seen = false;
  HIP_TRY(spin_until(h->stream, [&]() { return __atomic_load_n(&h->h_flag[0], __ATOMIC_ACQUIRE) != 0; }, &seen));
  HIP_TRY(hipGetLastError());
  h->pending = !seen;            // not seen after real syncs: the kernel has ended anyway, fetch_state copies dyn->state[0]
  h->last_parity = 0;
  return NDT_OK;
}

int32_t finish_chunk_run(ndt2d_handle* h) {
  { const int32_t fs = finish_small_run(h); if (fs != NDT_OK) return fs; }
  if (!h->chunk_run.active) return NDT_OK;
  bool seen = false;
  HIP_TRY(chunk_run_finish(h->chunk_run, h->stream, h->h_flag, &seen));
  HIP_TRY(hipGetLastError());
  if (!seen) { set_error("the Gauss-Newton loop did not report its end"); return NDT_ERR_HIP; }
  h->pending = false;                                  // the finishing launch wrote the result into h_state
  return NDT_OK;
}

int32_t run_align(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n, const double pose[3],
                  int fixed_override, int check_every, bool wait = true, bool own_source = false) {
  TraceRange range("ndt2d_align: Gauss-Newton loop");
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }     // an unfinished asynchronous call
  if (n == 0 || n > kMaxSourcePoints || !pose) return NDT_ERR_INVALID_ARG;
  if (h->n_valid < 1) {
    h->pending = false;
    std::memset(h->h_state, 0, sizeof(IterState));
    h->h_state->pose[0] = pose[0]; h->h_state->pose[1] = pose[1]; h->h_state->pose[2] = pose[2];
    h->h_state->status = NDT_TOO_FEW_CELLS;
    h->h_state->done = 2;               // marks "result already on the host"
    return NDT_OK;
  }
  const int fixed = fixed_override >= 0 ? fixed_override : h->prm.fixed_iterations;
  const int K = fixed > 0 ? fixed : h->prm.max_iterations;
  const int blocks = blocks_for(n);
  h->wide = h->use_wide && n >= h->wide_threshold;
  const bool chunked = h->use_graph && check_every > 0 && fixed == 0;
  __atomic_store_n(&h->h_flag[0], 0, __ATOMIC_RELAXED);
  __atomic_store_n(&h->h_flag[1], 0, __ATOMIC_RELAXED);
  h->call_seq = h->call_seq == 0x7fffffff ? 1 : h->call_seq + 1;
  if (h->use_small && h->use_graph && n <= (size_t)kSmallMaxPoints) {
    // short scan: the whole loop in one launch of one workgroup (ndt2d_small.hpp)
    const bool newton = h->prm.hessian_mode == NDT_HESSIAN_NEWTON, four = h->prm.overlap_grids == 4;
#define NDT_LAUNCH_SMALL(MODE, NG)                                                                                       \
  do {                                                                                                                   \
    if (n <= (size_t)kSmallLoPoints)                                                                                     \
      hipLaunchKernelGGL((k_align_small<MODE, NG, kSmallThreadsLo>), dim3(1), dim3(kSmallThreadsLo), 0, h->stream,       \
                         h->d_static, d_sx, d_sy, (int)n, pose[0], pose[1], pose[2], fixed, &h->d_dyn->state[0],         \
                         h->h_state, h->h_flag);                                                                         \
    else                                                                                                                 \
      hipLaunchKernelGGL((k_align_small<MODE, NG, kSmallThreadsHi>), dim3(1), dim3(kSmallThreadsHi), 0, h->stream,       \
                         h->d_static, d_sx, d_sy, (int)n, pose[0], pose[1], pose[2], fixed, &h->d_dyn->state[0],         \
                         h->h_state, h->h_flag);                                                                         \
  } while (0)
    if (newton) { if (four) NDT_LAUNCH_SMALL(1, 4); else NDT_LAUNCH_SMALL(1, 1); }
    else        { if (four) NDT_LAUNCH_SMALL(0, 4); else NDT_LAUNCH_SMALL(0, 1); }
#undef NDT_LAUNCH_SMALL
    HIP_TRY(hipGetLastError());
    h->small_run = true;
    h->pending = false;
    return wait ? finish_small_run(h) : NDT_OK;
  }
  hipLaunchKernelGGL(k_begin, dim3(1), dim3(64), 0, h->stream, h->d_call, h->d_dyn, d_sx, d_sy, (int)n, pose[0], pose[1],
                     pose[2], fixed, chunked ? h->h_state : (IterState*)nullptr, chunked ? h->h_flag : (int*)nullptr,
                     h->call_seq);
  int k = 0;
  if (h->use_graph) {
    if (chunked) {
      // converged mode: chunks of launches until the finishing launch raises the host flag
      const int chunk = check_every + (check_every & 1);
      const int32_t gs = ensure_graph(h, chunk, blocks);
      if (gs != NDT_OK) return gs;
      h->chunk_run.drain = !own_source;                // the handle's own staging arrays outlive the call
      h->chunk_run.seq = h->call_seq;
      HIP_TRY(chunk_run_begin(h->chunk_run, h->graph_exec, h->stream, chunk, K + 1));
      h->pending = false;
      return wait ? finish_chunk_run(h) : NDT_OK;
    } else {
      const int32_t gs = ensure_graph(h, K + 1, blocks);
      if (gs != NDT_OK) return gs;
      HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
      k = K + 1;
    }
  } else {
    for (; k <= K; ++k) {
      launch_iter(h, blocks, k);
      if (check_every > 0 && fixed == 0 && k < K && (k % check_every) == check_every - 1) {
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->h_state, &h->d_dyn->state[k & 1], sizeof(IterState), hipMemcpyDeviceToHost,
                               h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->h_state->done) { ++k; break; }
      }
    }
  }
  HIP_TRY(hipGetLastError());
  h->last_parity = (k - 1) & 1;
  h->h_state->done = 0;
  h->pending = true;
  return NDT_OK;
}

int32_t fetch_state(ndt2d_handle* h) {
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  if (h->pending) {
    HIP_TRY(hipMemcpyAsync(h->h_state, &h->d_dyn->state[h->last_parity], sizeof(IterState),
                           hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->pending = false;
  }
  return NDT_OK;
}

void sym6_to_9(const double* s, double* H) {
  H[0] = s[0]; H[1] = s[1]; H[2] = s[3];
  H[3] = s[1]; H[4] = s[2]; H[5] = s[4];
  H[6] = s[3]; H[7] = s[4]; H[8] = s[5];
}

void state_to_result(const IterState& s, ndt2d_result* out) {
  std::memset(out, 0, sizeof(*out));
  for (int j = 0; j < 3; ++j) { out->pose[j] = s.pose[j]; out->g[j] = s.g[j]; }
  sym6_to_9(s.H, out->H);
  out->score = s.score;
  out->iterations = s.iter;
  out->n_hit = s.n_hit;
  out->status = s.status;
}

}  // namespace

// ------------------------------------------------------------------------------ C ABI
extern "C" {

int32_t ndt_abi_version(void) { return NDT_ABI_VERSION; }

const char* ndt_status_string(int32_t s) {
  switch (s) {
    case NDT_OK: return "ok";
    case NDT_NOT_CONVERGED: return "not converged (max_iterations reached)";
    case NDT_DEGENERATE_HESSIAN: return "degenerate Hessian";
    case NDT_TOO_FEW_HITS: return "too few source points in valid cells";
    case NDT_TOO_FEW_CELLS: return "target grid has no valid cell";
    case NDT_ERR_INVALID_ARG: return "invalid argument";
    case NDT_ERR_NO_TARGET: return "no target set";
    case NDT_ERR_HIP: return "HIP error";
    case NDT_ERR_NO_DEVICE: return "no HIP device";
    case NDT_ERR_CAPACITY: return "capacity limit exceeded";
    case NDT_ERR_ALLOC: return "allocation failed";
    case NDT_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
  }
}

const char* ndt_last_error(void) { return last_error().c_str(); }

int32_t ndt_set_host_wait(int32_t mode) {
  if (mode != 0 && mode != 1) return NDT_ERR_INVALID_ARG;
  ndt::host_wait_mode().store(mode, std::memory_order_relaxed);
  return NDT_OK;
}

int32_t ndt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int32_t ndt_magnusson_constants(double outlier_ratio, double cell_size, int32_t dim, double* d1, double* d2) {
  if (!d1 || !d2 || !(outlier_ratio > 0.0) || !(outlier_ratio < 1.0) || !(cell_size > 0.0) || (dim != 2 && dim != 3))
    return NDT_ERR_INVALID_ARG;
  const double c1 = 10.0 * (1.0 - outlier_ratio);
  const double c2 = outlier_ratio / std::pow(cell_size, (double)dim);
  const double d3 = -std::log(c2);
  const double md1 = -std::log(c1 + c2) - d3;                      // Magnusson's d1 (negative)
  const double md2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / md1);
  if (!std::isfinite(md1) || !std::isfinite(md2) || !(md1 < 0.0) || !(md2 > 0.0)) return NDT_ERR_INVALID_ARG;
  *d1 = -md1;
  *d2 = md2;
  return NDT_OK;
}

int32_t ndt2d_calibrated_covariance(const double H[9], int32_t hessian_mode, double cov[9]) {
  if (!H || !cov || (hessian_mode != NDT_HESSIAN_GAUSS_NEWTON && hessian_mode != NDT_HESSIAN_NEWTON)) return NDT_ERR_INVALID_ARG;
  for (int i = 0; i < 9; ++i) cov[i] = 0.0;
  // symmetric 3x3 inverse by cofactors, positive definiteness by the leading minors
  const double a = H[0], b = 0.5 * (H[1] + H[3]), c = 0.5 * (H[2] + H[6]), d = H[4], e = 0.5 * (H[5] + H[7]), f = H[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(a > 0.0) || !(a * d - b * b > 0.0) || !(det > 0.0) || !std::isfinite(det)) return NDT_DEGENERATE_HESSIAN;
  const bool newton = hessian_mode == NDT_HESSIAN_NEWTON;
  const double st = std::sqrt(newton ? NDT_COV_SCALE_NEWTON_TRANS : NDT_COV_SCALE_GN_TRANS);
  const double sr = std::sqrt(newton ? NDT_COV_SCALE_NEWTON_ROT : NDT_COV_SCALE_GN_ROT);
  const double S[3] = {st, st, sr};
  const double inv[9] = {c00 / det, c01 / det, c02 / det, c01 / det, (a * f - c * c) / det, (b * c - a * e) / det,
                         c02 / det, (b * c - a * e) / det, (a * d - b * b) / det};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) cov[3 * i + j] = S[i] * inv[3 * i + j] * S[j];
  return NDT_OK;
}

int32_t ndt2d_polar_to_points_dev(const float* d_ranges, size_t n, double angle_min, double angle_inc,
                                  double range_min, double range_max, float* d_x, float* d_y, void* stream) {
  if (!d_ranges || !d_x || !d_y || n == 0 || !std::isfinite(angle_min) || !std::isfinite(angle_inc))
    return NDT_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_polar_to_points, dim3(stream_blocks(n)), dim3(kBlock), 0, (hipStream_t)stream, d_ranges, n,
                     angle_min, angle_inc, (float)range_min, (float)range_max, d_x, d_y);
  HIP_TRY(hipGetLastError());
  return NDT_OK;
}

void ndt2d_default_params(ndt2d_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->cell_size = 0.5;
  p->min_points = 3;
  p->hessian_mode = NDT_HESSIAN_GAUSS_NEWTON;
  p->eig_ratio = 1e-3;
  p->d1 = 1.0;
  p->d2 = 1.0;
  p->max_iterations = 100;
  p->fixed_iterations = 0;
  p->eps_trans = 1e-5;
  p->eps_rot = 1e-5;
  p->step_max_trans = 0.5;
  p->step_max_rot = 0.2;
  p->min_hits = 3;
  p->step_scale = 1.0;
}

int32_t ndt2d_create(const ndt2d_params* p, int32_t device_id, ndt2d_handle** out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = nullptr;
  const int32_t st = check_params(p);
  if (st != NDT_OK) return st;
  const int ndev = ndt_device_count();
  if (ndev <= 0) { set_error("no HIP device visible: this library has no CPU fallback"); return NDT_ERR_NO_DEVICE; }
  if (device_id < 0 || device_id >= ndev) return NDT_ERR_INVALID_ARG;
  ndt2d_handle* h = new (std::nothrow) ndt2d_handle();
  if (!h) return NDT_ERR_ALLOC;
  h->device = device_id;
  h->prm = *p;
  auto fail = [&](int32_t code) { ndt2d_destroy(h); return code; };
  if (hipSetDevice(device_id) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipMalloc((void**)&h->d_bounds, 4 * sizeof(unsigned int)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_counters, ndt::kCountInts * sizeof(int)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_parts, kBoundsParts * sizeof(float4)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_acc2, 512) != hipSuccess) return fail(NDT_ERR_ALLOC);
  {   // k_chunk_sort: 32 KB of points + up to 32 KB of tile histogram, just over the 64 KB a kernel gets without asking
    const int lds = 4096 * (int)sizeof(float2) + kBinMaxTiles * (int)sizeof(unsigned int);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chunk_sort<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chunk_sort<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chunk_sort<16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return fail(NDT_ERR_HIP);
  }
  if (hipMalloc((void**)&h->d_outside, sizeof(unsigned long long)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_static, sizeof(AlignStatic)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_call, sizeof(AlignCall)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_dyn, sizeof(AlignDyn)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipHostMalloc((void**)&h->h_static, sizeof(AlignStatic), hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipHostMalloc((void**)&h->h_state, sizeof(IterState), hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipHostMalloc((void**)&h->h_flag, 64, hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipEventCreateWithFlags(&h->upload_ev, hipEventDisableTiming) != hipSuccess) return fail(NDT_ERR_HIP);
  *h->h_flag = 0;
  if (hipHostMalloc(&h->h_small, 256, hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMemset(h->d_dyn, 0, sizeof(AlignDyn)) != hipSuccess) return fail(NDT_ERR_HIP);
  *out = h;
  return NDT_OK;
}

int32_t ndt2d_destroy(ndt2d_handle* h) {
  if (!h) return NDT_OK;
  (void)hipSetDevice(h->device);
  (void)finish_chunk_run(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  drop_graph(h);
  if (h->h_state_multi) (void)hipHostFree(h->h_state_multi);
  void* dev[] = {h->d_acc2, h->d_split, h->d_bxy, h->d_table, h->d_parts, h->d_geom, h->d_dyn_multi, h->d_bounds, h->d_counters, h->d_outside, h->d_static, h->d_call, h->d_dyn, h->d_bx, h->d_by, h->d_tiles, h->d_tx, h->d_ty, h->d_sx, h->d_sy,
                 h->grid.rec, h->grid.acc};
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = {h->h_geom, h->h_static, h->h_state, h->h_small, h->h_flag};
  for (void* p : host) if (p) (void)hipHostFree(p);
  if (h->upload_ev) (void)hipEventDestroy(h->upload_ev);
  if (h->wait_ev) (void)hipEventDestroy(h->wait_ev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return NDT_OK;
}

void* ndt2d_stream(ndt2d_handle* h) { return h ? (void*)h->stream : nullptr; }

int32_t ndt2d_set_tuning(ndt2d_handle* h, int32_t knob, int64_t value) {
  if (!h) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  switch (knob) {
    case NDT_TUNE_LAUNCH_GRAPHS: h->use_graph = value != 0; return NDT_OK;
    case NDT_TUNE_WIDE_THRESHOLD: h->use_wide = value > 0; if (value > 0) h->wide_threshold = (size_t)value; return NDT_OK;
    case NDT_TUNE_SHORT_SCAN_KERNEL: h->use_small = value != 0; return NDT_OK;
    case NDT_TUNE_CHUNK_LAUNCHES: if (value < 2 || value > 128) return NDT_ERR_INVALID_ARG; h->check_every = (int)value; return NDT_OK;
    case NDT_TUNE_BINNED_BUILD:
      if (value < 0 || value > 2) return NDT_ERR_INVALID_ARG;
      h->use_binned_build = value != 0; if (value) h->build_variant = (int)value; return NDT_OK;
    case NDT_TUNE_SPLIT_FROM: if (value < 1 || value > 1000) return NDT_ERR_INVALID_ARG; h->split_from = (int)value; return NDT_OK;
    case NDT_TUNE_SINGLE_SYNC_BUILD: h->one_round_trip = value != 0; return NDT_OK;
    default: return NDT_ERR_INVALID_ARG;
  }
}

int32_t ndt2d_wait_stream(ndt2d_handle* h, void* producer_stream) {
  if (!h) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(order_after(h->stream, (hipStream_t)producer_stream, &h->wait_ev));
  return NDT_OK;
}

int32_t ndt2d_set_target(ndt2d_handle* h, const float* x, const float* y, size_t n) {
  if (!h || !x || !y || n == 0) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  const int32_t st = ensure_points(&h->d_tx, &h->d_ty, &h->tcap, n);
  if (st != NDT_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->d_tx, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_ty, y, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  return set_target_impl(h, h->d_tx, h->d_ty, n);
}

int32_t ndt2d_set_target_dev(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n, void* stream) {
  if (!h || !d_x || !d_y || n == 0) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  if (stream) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   // producer of d_x/d_y
  return set_target_impl(h, d_x, d_y, n);
}

int32_t ndt2d_reserve_target(ndt2d_handle* h, double xmin, double ymin, double xmax, double ymax) {
  if (!h || !(xmin <= xmax) || !(ymin <= ymax) || !std::isfinite(xmin) || !std::isfinite(xmax) || !std::isfinite(ymin) ||
      !std::isfinite(ymax))
    return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  h->has_target = false;
  const int32_t gs = setup_geometry(h, (float)xmin, (float)xmax, (float)ymin, (float)ymax);
  if (gs != NDT_OK) return gs;
  const size_t ncell = (size_t)h->grid.W * h->grid.H * h->grid.ngrid;
  HIP_TRY(hipMemsetAsync(h->grid.acc, 0, ncell * sizeof(CellAcc), h->stream));
  HIP_TRY(hipMemsetAsync(h->grid.rec, 0, 2 * ncell * sizeof(float4), h->stream));
  h->n_valid = 0;
  h->n_points = 0;
  h->has_target = true;
  return upload_static(h);
}

int32_t ndt2d_add_target_points_dev(ndt2d_handle* h, const float* d_x, const float* d_y, size_t n,
                                    const double pose[3], size_t* n_outside, void* stream) {
  if (!h || !d_x || !d_y || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  // order after the caller's producer stream, as ndt2d_set_target_dev does
  if (stream) HIP_TRY(order_after(h->stream, (hipStream_t)stream));
  // the scan is moved into the map frame inside the build's first kernel (k_chunk_sort; k_transform_points on the other paths)
  MoveArgs mv{1.f, 0.f, 0.f, 0.f, 0};
  if (pose) { mv.cs = (float)std::cos(pose[2]); mv.sn = (float)std::sin(pose[2]); mv.tx = (float)pose[0]; mv.ty = (float)pose[1]; mv.use = 1; }
  unsigned long long outside = 0;
  const int32_t fs = accumulate_and_finalise(h, d_x, d_y, n, /*merge=*/true, &outside, &mv);
  if (n_outside) *n_outside = (size_t)outside;
  if (fs != NDT_OK) { h->has_target = false; return fs; }
  h->n_points += n - (size_t)outside;
  return NDT_OK;          // geometry, storage and parameters are unchanged: the device context stays as it is
}

int32_t ndt2d_add_target_points(ndt2d_handle* h, const float* x, const float* y, size_t n, size_t* n_outside) {
  if (!h || !x || !y || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  const int32_t st = ensure_points(&h->d_tx, &h->d_ty, &h->tcap, n);
  if (st != NDT_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->d_tx, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_ty, y, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  unsigned long long outside = 0;
  const int32_t fs = accumulate_and_finalise(h, h->d_tx, h->d_ty, n, /*merge=*/true, &outside);
  if (n_outside) *n_outside = (size_t)outside;
  if (fs != NDT_OK) { h->has_target = false; return fs; }
  h->n_points += n - (size_t)outside;
  return NDT_OK;          // as above: nothing in the device context changes
}

int32_t ndt2d_get_grid_info(ndt2d_handle* h, ndt2d_grid_info* info) {
  if (!h || !info) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  info->ox = h->grid.ox; info->oy = h->grid.oy;
  info->inv_cell = h->grid.inv_c; info->cell = h->grid.cell32;
  info->width = h->grid.W; info->height = h->grid.H;
  info->n_valid = h->n_valid; info->n_points = (int32_t)h->n_points;
  return NDT_OK;
}

int32_t ndt2d_get_grid(ndt2d_handle* h, int32_t* count, float* mean_xy, float* icov_abc) {
  if (!h) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  const size_t ncell = (size_t)h->grid.W * h->grid.H;          // grid 0 (the unshifted one)
  float4* rec = new (std::nothrow) float4[2 * ncell];
  CellAcc* acc = count ? new (std::nothrow) CellAcc[ncell] : nullptr;
  int32_t rc = NDT_OK;
  if (!rec || (count && !acc)) rc = NDT_ERR_ALLOC;
  if (rc == NDT_OK && hipMemcpyAsync(rec, h->grid.rec, 2 * ncell * sizeof(float4), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = NDT_ERR_HIP;
  if (rc == NDT_OK && acc && hipMemcpyAsync(acc, h->grid.acc, ncell * sizeof(CellAcc), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = NDT_ERR_HIP;
  if (rc == NDT_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = NDT_ERR_HIP;
  if (rc == NDT_OK) {
    for (size_t k = 0; k < ncell; ++k) {
      const float4 a = rec[2 * k], b = rec[2 * k + 1];
      if (count) count[k] = (int32_t)acc[k].n;
      if (mean_xy) { mean_xy[2 * k] = a.x; mean_xy[2 * k + 1] = a.y; }
      if (icov_abc) {
        const bool valid = b.z > 0.f;
        icov_abc[3 * k] = valid ? a.z : 0.f;
        icov_abc[3 * k + 1] = valid ? a.w : 0.f;
        icov_abc[3 * k + 2] = valid ? b.y : 0.f;
      }
    }
  } else if (rc == NDT_ERR_HIP) {
    set_error(hipGetErrorString(hipGetLastError()));
  }
  delete[] rec; delete[] acc;
  return rc;
}

int32_t ndt2d_evaluate(ndt2d_handle* h, const float* sx, const float* sy, size_t n, const double pose[3],
                       ndt2d_eval* out) {
  if (!h || !sx || !sy || !pose || !out || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  int32_t st = ensure_points(&h->d_sx, &h->d_sy, &h->scap, n);
  if (st != NDT_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->d_sx, sx, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_sy, sy, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  st = run_align(h, h->d_sx, h->d_sy, n, pose, /*fixed_override=*/1, /*check_every=*/0);
  if (st != NDT_OK) return st;
  st = fetch_state(h);
  if (st != NDT_OK) return st;
  std::memset(out, 0, sizeof(*out));
  sym6_to_9(h->h_state->H, out->H);
  for (int j = 0; j < 3; ++j) out->g[j] = h->h_state->g[j];
  out->score = h->h_state->score;
  out->n_hit = h->h_state->n_hit;
  return NDT_OK;
}

// the same with the scan already on the device (order the handle behind its producer with ndt2d_wait_stream first)
int32_t ndt2d_evaluate_dev(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n, const double pose[3],
                           ndt2d_eval* out) {
  if (!h || !d_sx || !d_sy || !pose || !out || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  int32_t st = run_align(h, d_sx, d_sy, n, pose, /*fixed_override=*/1, /*check_every=*/0);
  if (st != NDT_OK) return st;
  st = fetch_state(h);
  if (st != NDT_OK) return st;
  std::memset(out, 0, sizeof(*out));
  sym6_to_9(h->h_state->H, out->H);
  for (int j = 0; j < 3; ++j) out->g[j] = h->h_state->g[j];
  out->score = h->h_state->score;
  out->n_hit = h->h_state->n_hit;
  return NDT_OK;
}

int32_t ndt2d_align_trace(ndt2d_handle* h, const float* sx, const float* sy, size_t n, const double init_pose[3],
                          ndt2d_result* rows, int32_t capacity, int32_t* n_rows, ndt2d_result* out) {
  if (!h || !sx || !sy || !init_pose || !rows || capacity < 1 || !n_rows || n == 0 || n > kMaxSourcePoints)
    return NDT_ERR_INVALID_ARG;
  *n_rows = 0;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  TraceRange range("ndt2d_align_trace");
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  if (h->n_valid < 1) {
    std::memset(&rows[0], 0, sizeof(ndt2d_result));
    for (int j = 0; j < 3; ++j) rows[0].pose[j] = init_pose[j];
    rows[0].status = NDT_TOO_FEW_CELLS;
    if (out) *out = rows[0];
    return NDT_OK;
  }
  const int32_t st = ensure_points(&h->d_sx, &h->d_sy, &h->scap, n);
  if (st != NDT_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->d_sx, sx, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_sy, sy, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  // the launch-per-iteration kernels, one plain launch and one state fetch per iteration
  const int fixed = h->prm.fixed_iterations;
  const int K = fixed > 0 ? fixed : h->prm.max_iterations;
  h->wide = h->use_wide && n >= h->wide_threshold;
  h->call_seq = h->call_seq == 0x7fffffff ? 1 : h->call_seq + 1;
  hipLaunchKernelGGL(k_begin, dim3(1), dim3(64), 0, h->stream, h->d_call, h->d_dyn, h->d_sx, h->d_sy, (int)n, init_pose[0],
                     init_pose[1], init_pose[2], fixed, (IterState*)nullptr, (int*)nullptr, h->call_seq);
  for (int k = 0; k <= K; ++k) {
    launch_iter(h, blocks_for(n), k);
    HIP_TRY(hipGetLastError());
    if (k == 0) continue;                                  // launch 0 only evaluates
    HIP_TRY(hipMemcpyAsync(h->h_state, &h->d_dyn->state[k & 1], sizeof(IterState), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (*n_rows < capacity) state_to_result(*h->h_state, &rows[(*n_rows)++]);
    if (h->h_state->done) break;
  }
  h->pending = false;
  h->h_state->done = 2;                                     // the final state is in h_state
  if (out) state_to_result(*h->h_state, out);
  return NDT_OK;
}

int32_t ndt2d_align_finish(ndt2d_handle* h, ndt2d_result* out) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const int32_t st = fetch_state(h);
  if (st != NDT_OK) return st;
  const IterState& s = *h->h_state;
  std::memset(out, 0, sizeof(*out));
  for (int j = 0; j < 3; ++j) { out->pose[j] = s.pose[j]; out->g[j] = s.g[j]; }
  sym6_to_9(s.H, out->H);
  out->score = s.score;
  out->iterations = s.iter;
  out->n_hit = s.n_hit;
  out->status = s.status;
  return NDT_OK;
}

int32_t ndt2d_align_dev_async(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n,
                              const double init_pose[3]) {
  if (!h || !d_sx || !d_sy || !init_pose) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  // converged mode: the first two chunks are enqueued here, ndt2d_align_finish keeps the loop fed
  return run_align(h, d_sx, d_sy, n, init_pose, -1, h->check_every, /*wait=*/false);
}

int32_t ndt2d_align_dev(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n,
                        const double init_pose[3], ndt2d_result* out) {
  if (!h || !d_sx || !d_sy || !init_pose || !out) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const int32_t st = run_align(h, d_sx, d_sy, n, init_pose, -1, h->check_every);
  if (st != NDT_OK) return st;
  return ndt2d_align_finish(h, out);
}

int32_t ndt2d_align(ndt2d_handle* h, const float* sx, const float* sy, size_t n, const double init_pose[3],
                    ndt2d_result* out) {
  if (!h || !sx || !sy || !init_pose || !out || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  const int32_t st = ensure_points(&h->d_sx, &h->d_sy, &h->scap, n);
  if (st != NDT_OK) return st;
  HIP_TRY(hipMemcpyAsync(h->d_sx, sx, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_sy, sy, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  const int32_t rs = run_align(h, h->d_sx, h->d_sy, n, init_pose, -1, h->check_every, /*wait=*/true, /*own_source=*/true);
  if (rs != NDT_OK) return rs;
  return ndt2d_align_finish(h, out);
}

}  // extern "C"


// ---- multi-start alignment (ndt2d_multi_start.hpp) ---------------------------------------------
namespace {

template <int MODE, int THREADS, bool SHARED>
const void* multi_kernel(int nh, bool four) {
  if (four) return (const void*)&k_iterate_multi<MODE, 1, THREADS, SHARED, 4>;     // overlapping grids: one start per workgroup
  if constexpr (THREADS <= 256) {           // a 1024-thread workgroup fills its CU with one start
    if (nh == 2) return (const void*)&k_iterate_multi<MODE, 2, THREADS, SHARED>;
    if (nh == 4) return (const void*)&k_iterate_multi<MODE, 4, THREADS, SHARED>;
  }
  return (const void*)&k_iterate_multi<MODE, 1, THREADS, SHARED>;
}

template <int MODE, int THREADS, bool SHARED>
const void* body_kernel(int, bool four) {   // the split chain evaluates one start per workgroup
  if (four) return (const void*)&k_multi_body<MODE, 1, THREADS, SHARED, 4>;
  return (const void*)&k_multi_body<MODE, 1, THREADS, SHARED>;
}

// m alignments against the cached grid in one launch chain: of one scan from m initial poses (shared), or
// of m scans (each with its initial pose).  sxs / sys / ns have one entry when shared, m otherwise.
int32_t multi_align(ndt2d_handle* h, const float* const* sxs, const float* const* sys, const size_t* ns, bool shared,
                    const double* init_poses, int32_t m, ndt2d_result* results) {
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  TraceRange range(shared ? "ndt2d_align_multi_start" : "ndt2d_align_multi_scan");
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_chunk_run(h); if (fs != NDT_OK) return fs; }
  if (h->n_valid < 1) {
    for (int32_t k = 0; k < m; ++k) {
      std::memset(&results[k], 0, sizeof(ndt2d_result));
      for (int j = 0; j < 3; ++j) results[k].pose[j] = init_poses[3 * k + j];
      results[k].status = NDT_TOO_FEW_CELLS;
    }
    return NDT_OK;
  }
  { const int32_t st = ensure_multi_buffers(h); if (st != NDT_OK) return st; }
  size_t n_max = 0;
  for (int32_t k = 0; k < (shared ? 1 : m); ++k) n_max = ns[k] > n_max ? ns[k] : n_max;
  if (!h->d_dyn_multi) {
    HIP_TRY(hipMalloc((void**)&h->d_dyn_multi, sizeof(AlignDynMulti)));
    HIP_TRY(hipMemsetAsync(h->d_dyn_multi, 0, sizeof(AlignDynMulti), h->stream));
  }
  const bool newton = h->prm.hessian_mode == NDT_HESSIAN_NEWTON;
  const bool wide = h->use_wide && n_max >= h->wide_threshold;
  const int fixed = h->prm.fixed_iterations;
  const int K = fixed > 0 ? fixed : h->prm.max_iterations;
  const bool converged_mode = fixed == 0;
  // grid = (256 workgroups) x (subsets of nh starts): up to six workgroups per CU carry one start
  // each, more starts double up inside the workgroups; a 1024-thread workgroup fills a CU alone
  const bool split = m >= h->split_from;
  // fused chain: six one-start workgroups fit a CU (78 VGPRs), more starts double up inside the workgroups;
  // split chain: one start per evaluation workgroup (measured best: 31.8 us per step at 64 starts, 33.0 with four)
  const bool four = h->prm.overlap_grids == 4;     // Biber's four overlapping grids: every point scores against all four (NG = 4)
  const int nh = (split || wide || m <= 6 || four) ? 1 : (m <= 12 ? 2 : 4);
  const int subsets = (m + nh - 1) / nh;
  // From kSplitFrom starts on the chain alternates two kernels per iteration (one workgroup per start solves,
  // then everybody evaluates): the 256-fold redundant prologues of the fused kernel cost more than the
  // second kernel boundary there.
  const void* func;
  if (!split) {
    if (shared)
      func = wide ? (newton ? multi_kernel<1, kIterThreadsWide, true>(nh, four) : multi_kernel<0, kIterThreadsWide, true>(nh, four))
                  : (newton ? multi_kernel<1, kIterThreads, true>(nh, four) : multi_kernel<0, kIterThreads, true>(nh, four));
    else
      func = wide ? (newton ? multi_kernel<1, kIterThreadsWide, false>(nh, four) : multi_kernel<0, kIterThreadsWide, false>(nh, four))
                  : (newton ? multi_kernel<1, kIterThreads, false>(nh, four) : multi_kernel<0, kIterThreads, false>(nh, four));
  } else {
    if (shared)
      func = wide ? (newton ? body_kernel<1, kIterThreadsWide, true>(nh, four) : body_kernel<0, kIterThreadsWide, true>(nh, four))
                  : (newton ? body_kernel<1, kIterThreads, true>(nh, four) : body_kernel<0, kIterThreads, true>(nh, four));
    else
      func = wide ? (newton ? body_kernel<1, kIterThreadsWide, false>(nh, four) : body_kernel<0, kIterThreadsWide, false>(nh, four))
                  : (newton ? body_kernel<1, kIterThreads, false>(nh, four) : body_kernel<0, kIterThreads, false>(nh, four));
  }
  __atomic_store_n(&h->h_flag[0], 0, __ATOMIC_RELAXED);
  __atomic_store_n(&h->h_flag[1], 0, __ATOMIC_RELAXED);
  h->call_seq = h->call_seq == 0x7fffffff ? 1 : h->call_seq + 1;
  StartPoses sp{};
  StartScans sc{};
  for (int k = 0; k < m; ++k) {
    for (int j = 0; j < 3; ++j) sp.p[k][j] = init_poses[3 * k + j];
    sc.sx[k] = sxs[shared ? 0 : k]; sc.sy[k] = sys[shared ? 0 : k]; sc.n[k] = (int)ns[shared ? 0 : k];
  }
  // AlignCall.n doubles as the "armed" word of the chain: the scan size when shared, any non-zero value otherwise
  hipLaunchKernelGGL(k_begin_multi, dim3(1), dim3(64), 0, h->stream, h->d_call, h->d_dyn_multi, sxs[0], sys[0], (int)n_max, sp, sc,
                     (int)m, fixed, converged_mode ? h->h_state_multi : (IterState*)nullptr,
                     converged_mode ? h->h_flag : (int*)nullptr, h->call_seq);
  HIP_TRY(hipGetLastError());
  const int launches = converged_mode ? h->check_every + (h->check_every & 1) : K + 1;
  hipGraphExec_t exec = nullptr;
  const int key = 0x10000 | (shared ? 0 : 0x20000) | (split ? 0x40000 : 0) | (nh << 5) | (subsets << 8) | h->prm.hessian_mode | (wide ? 16 : 0);
  if (!split)
    HIP_TRY(h->graphs.get(func, dim3(kMaxBlocks, subsets), dim3(wide ? kIterThreadsWide : kIterThreads), (void*)h->d_static,
                          (void*)h->d_call, (void*)h->d_dyn_multi, launches, key, h->stream, &exec));
  else {
    // launch shapes in powers of two (slots >= m are born finished: their workgroups return at once): a caller whose m
    // varies from call to call replays one of three cached graphs (16, 32, 64 starts) instead of instantiating new ones
    int mg = 1;
    while (mg < m) mg <<= 1;
    HIP_TRY(h->graphs.get2((const void*)&k_multi_solve, dim3(mg), dim3(kBlock), func, dim3(kMaxBlocks, mg),
                           dim3(wide ? kIterThreadsWide : kIterThreads), (void*)h->d_static, (void*)h->d_call,
                           (void*)h->d_dyn_multi, launches, (key & ~(0xff << 8)) | (mg << 8) | (mg << 20), h->stream, &exec));
  }
  if (converged_mode) {
    bool seen = false;
    HIP_TRY(run_chunks_until_flag(exec, h->stream, h->h_flag, launches, K + 1, h->call_seq, &seen));
    HIP_TRY(hipGetLastError());
    if (!seen) { set_error("the multi-start loop did not report its end"); return NDT_ERR_HIP; }
  } else {
    HIP_TRY(hipGraphLaunch(exec, h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_state_multi, h->d_dyn_multi->state[K & 1], kMaxStarts * sizeof(IterState),
                           hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  for (int k = 0; k < m; ++k) state_to_result(h->h_state_multi[k], &results[k]);
  return NDT_OK;
}

}  // namespace

extern "C" int32_t ndt2d_align_multi_start_dev(ndt2d_handle* h, const float* d_sx, const float* d_sy, size_t n,
                                               const double* init_poses, int32_t m, ndt2d_result* results) {
  if (!h || !d_sx || !d_sy || !init_poses || !results || m < 1 || m > kMaxStarts) return NDT_ERR_INVALID_ARG;
  if (n == 0 || n > kMaxSourcePoints) return NDT_ERR_INVALID_ARG;
  return multi_align(h, &d_sx, &d_sy, &n, /*shared=*/true, init_poses, m, results);
}

extern "C" int32_t ndt2d_align_multi_scan_dev(ndt2d_handle* h, const float* const* d_sx, const float* const* d_sy,
                                              const size_t* n, const double* init_poses, int32_t m, ndt2d_result* results) {
  if (!h || !d_sx || !d_sy || !n || !init_poses || !results || m < 1 || m > kMaxStarts) return NDT_ERR_INVALID_ARG;
  for (int32_t k = 0; k < m; ++k)
    if (!d_sx[k] || !d_sy[k] || n[k] == 0 || n[k] > kMaxSourcePoints) return NDT_ERR_INVALID_ARG;
  return multi_align(h, d_sx, d_sy, n, /*shared=*/false, init_poses, m, results);
}

#include "ndt2d_batch_api.hpp"
#include "ndt2d_multi_api.hpp"
#include "ndt3d_api.hpp"
#include "ndt3d_batch_api.hpp"
#include "ndt3d_multi_api.hpp"
#include "ndt_map_io.hpp"
