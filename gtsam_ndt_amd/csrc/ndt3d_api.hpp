// C-ABI of the 3D SE(3) variant (included at the end of ndt2d_api.hip: one translation unit).
#pragma once
#include <vector>

#include "ndt3d_kernels.hpp"
#include "ndt3d_build.hpp"
#include "ndt3d_multi.hpp"

struct ndt3d_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  ndt3d_params prm{};
  ndt::Grid3Dev grid{};
  size_t cell_capacity = 0;
  bool has_target = false;
  int n_valid = 0;
  unsigned int* d_bounds = nullptr;   // [6]
  int* d_counters = nullptr;          // counter shards of ndt3d_load_map's finalise (the builds keep theirs in d_tiles)
  int publish_seq = 0;                // k_build_publish's flag value of the build in flight (h_small + 192)
  float* d_parts3 = nullptr;          // [256][8]: per-workgroup partial bounding boxes (k_bounds3_parts)
  void* h_small = nullptr;            // pinned 256 B (counter shards at 0, the outside count at 128)
  float *d_t[3] = {nullptr, nullptr, nullptr}; size_t tcap = 0;
  float *d_b[3] = {nullptr, nullptr, nullptr}; size_t bcap = 0;      // binned build scratch
  unsigned int* d_tiles = nullptr; size_t tile_cap = 0;
  int last_ntile = 0;                 // tiles of the grid the handle holds (0: none): the launch bound of a single-sync build
  unsigned int* h_pub3 = nullptr;     // pinned [64 + 16]: the accumulator block's first 64 words of a single-sync build, flag at [64]
  bool one_round_trip = true;
  size_t tiles_clean = 0;             // leading words of d_tiles known to be zero (the last build's publish cleared them)
  unsigned char* d_split3 = nullptr; size_t split3_cap = 0;   // shared tiles' hand-off (ndt3d_build.hpp Split3Bufs): the slab pool
  float *d_s[3] = {nullptr, nullptr, nullptr}; size_t scap = 0;
  ndt::AlignStatic3* d_static = nullptr;
  ndt::AlignCall3* d_call = nullptr;
  ndt::AlignDyn3* d_dyn = nullptr;
  ndt::AlignStatic3* h_static = nullptr;
  hipEvent_t upload_ev = nullptr;     // recorded after the last upload from h_static
  ndt::IterState3* h_state = nullptr;
  unsigned long long* d_outside = nullptr;   // [1] points of the last build / update outside the extent
  int* h_flag = nullptr;              // pinned: raised by the launch that ends a converged-mode loop
  int call_seq = 0;                   // alignments enqueued so far
  ndt::ChainGraphCache graphs;
  hipGraphExec_t graph_exec = nullptr;   // selected by ensure_graph3, owned by `graphs`
  bool host_result = false;           // result already in h_state (no device work was enqueued)
  // an alignment in flight (ndt3d_align_dev_async ... ndt3d_align_finish): 0 none, 1 fixed-K chain with
  // the state copy enqueued behind it, 2 converged-mode chunk loop
  int in_flight = 0;
  ndt::ChunkRun chunk_run;
  ndt::AlignDynMulti3* d_dyn_multi = nullptr;   // multi-scan / multi-start chains (ndt3d_multi.hpp), on first use
  ndt::IterState3* h_state_multi = nullptr;     // pinned [kMaxStarts3]
};

namespace {

int32_t ensure3(float** d, size_t* cap, size_t n) {
  if (n <= *cap) return NDT_OK;
  for (int a = 0; a < 3; ++a) { if (d[a]) (void)hipFree(d[a]); d[a] = nullptr; }
  *cap = 0;
  const size_t want = n + n / 4 + 1024;
  for (int a = 0; a < 3; ++a) HIP_TRY(hipMalloc((void**)&d[a], want * sizeof(float)));
  *cap = want;
  return NDT_OK;
}

int32_t upload_static3(ndt3d_handle* h) {
  // as upload_static in 2D: the copy is left in flight (whatever reads d_static is ordered behind it on the same
  // stream); the pinned source is only rewritten once the previous copy has left it
  HIP_TRY(hipEventSynchronize(h->upload_ev));
  ndt::AlignStatic3* c = h->h_static;
  c->grid = h->grid;
  ndt::SolveParams& p = c->prm;
  p.d1 = (float)h->prm.d1; p.d2 = (float)h->prm.d2;
  p.hessian_mode = h->prm.hessian_mode; p.max_iterations = h->prm.max_iterations; p.min_hits = h->prm.min_hits; p.line_search = h->prm.line_search;
  p.eps_trans = h->prm.eps_trans; p.eps_rot = h->prm.eps_rot;
  p.step_max_trans = h->prm.step_max_trans; p.step_max_rot = h->prm.step_max_rot;
  p.step_scale = h->prm.step_scale > 0.0 ? h->prm.step_scale : 1.0;
  HIP_TRY(hipMemcpyAsync(h->d_static, c, sizeof(ndt::AlignStatic3), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipEventRecord(h->upload_ev, h->stream));
  return NDT_OK;
}

// The device words and buffers of a binned build of n points over at most `ntile` tiles.
// One block of device words carries everything a build adds into, so that ONE fill clears it (round 2: four) and one
// publish brings the results back:  counter shards [32] | outside count (u64) | pad to 64 (the arrival count of k_tile_count3's
// workgroups at word 41, a device-decided geometry from word kGeom3Word) | tile totals [ntile] | tickets of the shared tiles [ntile] |
// tile starts [ntile + 1] | scatter cursors [ntile] | number of (tile, share) workgroups [1] | their list [wg_bound]
struct Build3Bufs {
  int* d_cnt; unsigned long long* d_out;
  unsigned int *d_total, *d_ticket, *d_start, *d_cursor, *d_wgtotal, *d_wgmap;
  size_t wg_bound, zero_words;
  ndt::Split3Bufs sb;
};
int32_t ensure_build3_bufs(ndt3d_handle* h, size_t n, int ntile, bool binned, Build3Bufs* B) {
  using namespace ndt;
  B->wg_bound = (size_t)ntile + n / (size_t)kTile3SubMin + 1;
  const size_t tneed = 64 + 4 * (size_t)ntile + 4 + 1 + B->wg_bound;
  if (tneed > h->tile_cap) {
    if (h->d_tiles) (void)hipFree(h->d_tiles);
    h->d_tiles = nullptr; h->tile_cap = 0;
    HIP_TRY(hipMalloc((void**)&h->d_tiles, tneed * sizeof(unsigned int)));
    h->tile_cap = tneed;
    h->tiles_clean = 0;
  }
  B->d_cnt = reinterpret_cast<int*>(h->d_tiles);
  B->d_out = reinterpret_cast<unsigned long long*>(h->d_tiles + 32);
  B->d_total = h->d_tiles + 64;
  B->d_ticket = B->d_total + ntile;
  B->d_start = B->d_ticket + ntile;
  B->d_cursor = B->d_start + ntile + 1;
  B->d_wgtotal = B->d_cursor + ntile;
  B->d_wgmap = B->d_wgtotal + 1;
  B->zero_words = 64 + 2 * (size_t)ntile;
  B->sb = Split3Bufs{};
  if (!binned) return NDT_OK;
  { const int32_t st = ensure3(h->d_b, &h->bcap, n); if (st != NDT_OK) return st; }
  // the shared tiles' slabs: one per workgroup of the tile kernel's launch (only the shares of shared tiles use theirs)
  const size_t slabs = B->wg_bound;
  const size_t need = slabs * kSlabWords * sizeof(unsigned long long);
  if (need > h->split3_cap) {
    if (h->d_split3) (void)hipFree(h->d_split3);
    h->d_split3 = nullptr; h->split3_cap = 0;
    HIP_TRY(hipMalloc((void**)&h->d_split3, need + need / 4));
    h->split3_cap = need + need / 4;
  }
  B->sb.pool = reinterpret_cast<unsigned long long*>(h->d_split3);
  B->sb.capacity = (unsigned int)(slabs > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : slabs);
  return NDT_OK;
}

// a2 + a3 for n points into the grid whose geometry and storage are set: binned LDS build, or
// scattered global atomics for maps beyond the tile histogram.  merge = add to the cached sums
// (incremental submap update) instead of starting from zero.
int32_t accumulate3(ndt3d_handle* h, const float* dx, const float* dy, const float* dz, size_t n, bool merge,
                    unsigned long long* h_outside, const ndt::Rigid3F* move = nullptr) {
  using namespace ndt;
  Grid3Dev& g = h->grid;
  const size_t ncell = (size_t)g.W * g.H * g.D;
  const int ntx = (g.W + (1 << kT3x) - 1) >> kT3x, nty = (g.H + (1 << kT3y) - 1) >> kT3y, ntz = (g.D + (1 << kT3z) - 1) >> kT3z;
  const long long ntile_ll = (long long)ntx * nty * ntz;
  const bool binned = ntile_ll <= kBinMaxTiles && n <= 0xFFFFFFFFull;
  const int ntile = binned ? (int)ntile_ll : 0;
  Build3Bufs B{};
  { const int32_t bs = ensure_build3_bufs(h, n, ntile, binned, &B); if (bs != NDT_OK) return bs; }
  int* d_cnt = B.d_cnt;
  unsigned long long* d_out = B.d_out;
  unsigned int *d_total = B.d_total, *d_ticket = B.d_ticket, *d_start = B.d_start, *d_cursor = B.d_cursor, *d_wgtotal = B.d_wgtotal,
               *d_wgmap = B.d_wgmap;
  const size_t wg_bound = B.wg_bound;
  const Split3Bufs sb = B.sb;
  // (the publish of the build before cleared the block, unless this one needs more of it or that one did not finish)
  const bool clean = h->tiles_clean >= B.zero_words;
  h->tiles_clean = 0;
  if (!clean) HIP_TRY(hipMemsetAsync(h->d_tiles, 0, B.zero_words * sizeof(unsigned int), h->stream));
  if (binned) {
    // binned build (ndt3d_build.hpp)
    const BinGeom3 bg{g.ox, g.oy, g.oz, g.inv_c, g.W, g.H, g.D, ntx, nty, ntile};
    Move3Args mv{};
    if (move) { mv.T = *move; mv.use = 1; }
    size_t nb = (n + kBinThreads * 4 - 1) / (kBinThreads * 4);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_tile_count3, dim3((unsigned)nb), dim3(kBinThreads), ntile * sizeof(unsigned int), h->stream, dx, dy,
                       dz, n, bg, d_total, d_out, Geom3Args{}, Scan3Out{h->d_tiles + 41, d_start, d_cursor, d_wgtotal, d_wgmap}, mv);
    hipLaunchKernelGGL(k_tile_scatter3, dim3((unsigned)nb), dim3(kBinThreads), 2 * ntile * sizeof(unsigned int), h->stream,
                       dx, dy, dz, n, bg, d_cursor, h->d_b[0], h->d_b[1], h->d_b[2], (const GeomDev3*)nullptr, mv);
    // (no fill of the grid's sums: the workgroup that finishes a tile writes every voxel's sums, empty ones included)
    hipLaunchKernelGGL(k_tile_accumulate3, dim3((unsigned)wg_bound), dim3(kBinThreads), 0, h->stream, h->d_b[0], h->d_b[1],
                       h->d_b[2], d_start, g, ntx, nty, merge ? 1 : 0, h->prm.min_points, h->prm.eig_ratio, d_cnt, d_ticket, sb,
                       (const unsigned int*)d_wgtotal, (const unsigned int*)d_wgmap, (const GeomDev3*)nullptr, (const Grid3Dev*)nullptr);
    HIP_TRY(hipGetLastError());
    h->last_ntile = ntile;
  } else {
    if (move) {                                      // this path takes the points as they are: move them first
      const int32_t st = ensure3(h->d_t, &h->tcap, n);
      if (st != NDT_OK) return st;
      hipLaunchKernelGGL(k_transform_points3, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, dx, dy, dz, n, *move,
                         h->d_t[0], h->d_t[1], h->d_t[2]);
      HIP_TRY(hipGetLastError());
      dx = h->d_t[0]; dy = h->d_t[1]; dz = h->d_t[2];
    }
    if (!merge) HIP_TRY(hipMemsetAsync(g.acc, 0, ncell * sizeof(CellAcc3), h->stream));
    hipLaunchKernelGGL(k_accumulate3, dim3(stream_blocks(n)), dim3(kBlock), 0, h->stream, dx, dy, dz, n, g, d_out);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_finalise3, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, g,
                       h->prm.min_points, h->prm.eig_ratio, d_cnt);
    HIP_TRY(hipGetLastError());
  }
  // counter shards and outside count to the host through pinned memory and a flag (k_build_publish, as the 2D build)
  int* hc = (int*)h->h_small;
  unsigned long long* ho = (unsigned long long*)((char*)h->h_small + 128);
  {
    int* flag = reinterpret_cast<int*>(static_cast<char*>(h->h_small) + 192);
    h->publish_seq = h->publish_seq == 0x7fffffff ? 1 : h->publish_seq + 1;
    hipLaunchKernelGGL(k_build_publish_clear3, dim3(1), dim3(256), 0, h->stream, h->d_tiles, (unsigned int*)hc, 34, flag, h->publish_seq,
                       (int)B.zero_words);
    HIP_TRY(hipGetLastError());
    bool seen = false;
    const int want = h->publish_seq;
    HIP_TRY(spin_until(h->stream, [&]() { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want; }, &seen));
    if (!seen) {       // a second of silence: the kernel's own stores are the only copy (it clears the block behind them)
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) { set_error("the voxel-grid build did not report its end"); return NDT_ERR_HIP; }
    }
  }
  h->tiles_clean = B.zero_words;
  if (h_outside) *h_outside = *ho;
  int n_valid_sum = 0, n_over_sum = 0;
  sum_count_shards(hc, &n_valid_sum, &n_over_sum);
  h->n_valid = n_valid_sum;
  if (n_over_sum > 0) { set_error("a target cell holds more than 2^20 points"); return NDT_ERR_CAPACITY; }
  return NDT_OK;
}

// Voxel-grid geometry for a bounding box (oracle/ndt3d.py grid_geometry3) and storage for its voxels.
int32_t setup_geometry3(ndt3d_handle* h, const float lo[3], const float hi[3]) {
  using namespace ndt;
  const double c = h->prm.cell_size;
  Grid3Dev& g = h->grid;
  g.cell = c;
  g.inv_c = (float)(1.0 / c);
  float o[3];
  int dims[3];
  double ncell_d = 1.0;
  for (int a = 0; a < 3; ++a) {
    const float mn = lo[a], mx = hi[a];
    o[a] = (float)((std::floor((double)mn / c) - 1.0) * c);
    const volatile float f = (mx - o[a]) * g.inv_c;
    const double k = std::floor((double)f);
    if (!(k >= 0.0) || k > 1e7) { set_error("target extent too large for the cell size"); return NDT_ERR_CAPACITY; }
    dims[a] = (int)k + 2;
    ncell_d *= dims[a];
  }
  if (ncell_d > (double)kMaxCells) { set_error("3D target needs more than 2^27 cells"); return NDT_ERR_CAPACITY; }
  g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
  g.W = dims[0]; g.H = dims[1]; g.D = dims[2]; g.pad = 0;
  g.fix_scale = std::ldexp(1.0, kFixShift) / c;
  const size_t ncell = (size_t)g.W * g.H * g.D;
  if (ncell > h->cell_capacity) {
    void* old[] = {g.rec, g.acc};
    for (void* p : old) if (p) (void)hipFree(p);
    g.rec = nullptr; g.acc = nullptr; h->cell_capacity = 0;
    const size_t want = ncell + ncell / 8;
    HIP_TRY(hipMalloc((void**)&g.rec, 4 * want * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&g.acc, want * sizeof(CellAcc3)));
    h->cell_capacity = want;
  }
  return NDT_OK;
}

// ndt3d_set_target with ONE host round trip (the 3D twin of the 2D build's set_target_single_sync): a handle that already
// holds a grid enqueues the whole build at once - bounding-box partials (its workgroup 0 clears the accumulators), the count
// kernel whose prologue reduces the box and decides the grid (it must fit the handle's storage and a launch bound of twice
// the cached grid's tiles), scan, scatter, tile kernel with the geometry read from device memory - and one publish of the
// results AND the box.  If
// the grid does not fit, ok = 0 makes every kernel return and the caller builds the usual way with the box it now has.
// The host recomputes the geometry from the same box afterwards and compares.
int32_t set_target3_single_sync(ndt3d_handle* h, const float* dx, const float* dy, const float* dz, size_t n, bool* done,
                                unsigned int* hb_out, bool* have_bounds) {
  using namespace ndt;
  *done = false; *have_bounds = false;
  if (!h->one_round_trip || h->cell_capacity == 0 || h->last_ntile <= 0 || n > 0xFFFFFFFFull || !h->grid.rec || !h->grid.acc) return NDT_OK;
  long long tb = 2ll * h->last_ntile + 16;
  if (tb > kBinMaxTiles) tb = kBinMaxTiles;
  const int tile_bound = (int)tb;
  if (!h->h_pub3) HIP_TRY(hipHostMalloc((void**)&h->h_pub3, 80 * sizeof(unsigned int), hipHostMallocDefault));
  if (!h->d_parts3) HIP_TRY(hipMalloc((void**)&h->d_parts3, 256 * 8 * sizeof(float)));
  Build3Bufs B{};
  { const int32_t bs = ensure_build3_bufs(h, n, tile_bound, true, &B); if (bs != NDT_OK) return bs; }
  h->tiles_clean = 0;                                      // (k_bounds3_parts clears the block for this build)
  int sbk = stream_blocks(n);
  if (sbk > 256) sbk = 256;
  const GeomDev3* dg = reinterpret_cast<const GeomDev3*>(h->d_tiles + kGeom3Word);
  Grid3Dev* dgrid = &h->d_static->grid;
  HIP_TRY(hipEventSynchronize(h->upload_ev));              // (an upload of d_static still in flight would overwrite the header)
  hipLaunchKernelGGL(k_bounds3_parts, dim3(sbk), dim3(kBlock), 0, h->stream, dx, dy, dz, n, h->d_parts3, h->d_tiles, (int)B.zero_words,
                     kGeom3Word, (int)(sizeof(GeomDev3) / 4));
  Geom3Args ga{};
  ga.parts = h->d_parts3; ga.nparts = sbk; ga.tile_bound = tile_bound; ga.cell = h->prm.cell_size;
  ga.cell_capacity = (unsigned long long)h->cell_capacity; ga.grid = dgrid;
  ga.out = reinterpret_cast<GeomDev3*>(h->d_tiles + kGeom3Word);
  const BinGeom3 none{};
  size_t nb = (n + kBinThreads * 4 - 1) / (kBinThreads * 4);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_tile_count3, dim3((unsigned)nb), dim3(kBinThreads), tile_bound * sizeof(unsigned int), h->stream, dx, dy, dz, n,
                     none, B.d_total, B.d_out, ga, Scan3Out{h->d_tiles + 41, B.d_start, B.d_cursor, B.d_wgtotal, B.d_wgmap}, Move3Args{});
  hipLaunchKernelGGL(k_tile_scatter3, dim3((unsigned)nb), dim3(kBinThreads), 2 * tile_bound * sizeof(unsigned int), h->stream, dx, dy, dz,
                     n, none, B.d_cursor, h->d_b[0], h->d_b[1], h->d_b[2], dg, Move3Args{});
  hipLaunchKernelGGL(k_tile_accumulate3, dim3((unsigned)B.wg_bound), dim3(kBinThreads), 0, h->stream, h->d_b[0], h->d_b[1], h->d_b[2],
                     B.d_start, h->grid, 0, 0, 0, h->prm.min_points, h->prm.eig_ratio, B.d_cnt, B.d_ticket, B.sb,
                     (const unsigned int*)B.d_wgtotal, (const unsigned int*)B.d_wgmap, dg, (const Grid3Dev*)dgrid);
  HIP_TRY(hipGetLastError());
  unsigned int* hp = h->h_pub3;
  {
    int* flag = reinterpret_cast<int*>(hp + 64);
    h->publish_seq = h->publish_seq == 0x7fffffff ? 1 : h->publish_seq + 1;
    hipLaunchKernelGGL(k_build_publish_clear3, dim3(1), dim3(256), 0, h->stream, h->d_tiles, hp, 64, flag, h->publish_seq, (int)B.zero_words);
    HIP_TRY(hipGetLastError());
    bool seen = false;
    const int want = h->publish_seq;
    HIP_TRY(spin_until(h->stream, [&]() { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want; }, &seen));
    if (!seen) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) { set_error("the voxel-grid build did not report its end"); return NDT_ERR_HIP; }
    }
  }
  h->tiles_clean = B.zero_words;
  const GeomDev3* hg = reinterpret_cast<const GeomDev3*>(hp + kGeom3Word);
  for (int j = 0; j < 6; ++j) hb_out[j] = hg->bounds[j];
  *have_bounds = true;
  if (!hg->ok) return NDT_OK;                              // does not fit storage or bound (or no finite point): the usual way
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = ordered_to_float(hg->bounds[2 * a]); hi[a] = ordered_to_float(hg->bounds[2 * a + 1]); }
  // the host's view of the same geometry, from the same box; the storage is large enough, nothing is reallocated
  if (setup_geometry3(h, lo, hi) != NDT_OK) return NDT_OK;
  const Grid3Dev& g = h->grid;
  if (g.W != hg->bin.W || g.H != hg->bin.H || g.D != hg->bin.D || g.ox != hg->bin.ox || g.oy != hg->bin.oy || g.oz != hg->bin.oz) return NDT_OK;
  h->last_ntile = hg->bin.ntile;
  int n_valid_sum = 0, n_over_sum = 0;
  sum_count_shards(reinterpret_cast<const int*>(hp), &n_valid_sum, &n_over_sum);
  h->n_valid = n_valid_sum;
  if (n_over_sum > 0) { set_error("a target cell holds more than 2^20 points"); return NDT_ERR_CAPACITY; }
  *done = true;
  return NDT_OK;
}

int32_t set_target3_impl(ndt3d_handle* h, const float* dx, const float* dy, const float* dz, size_t n) {
  using namespace ndt;
  TraceRange range("ndt3d_set_target: voxel grid build");
  h->has_target = false;
  unsigned int* hb = (unsigned int*)h->h_small;
  unsigned int fast_bounds[6];
  bool done = false, have_bounds = false;
  { const int32_t fs = set_target3_single_sync(h, dx, dy, dz, n, &done, fast_bounds, &have_bounds); if (fs != NDT_OK) return fs; }
  if (done) {
    h->has_target = true;
    return upload_static3(h);
  }
  if (have_bounds) {
    for (int j = 0; j < 6; ++j) hb[j] = fast_bounds[j];    // the single-sync attempt measured the box already
  } else {
    // bounding box: one partial per workgroup, then one wave reduces them into pinned host memory and raises a flag
    if (!h->d_parts3) HIP_TRY(hipMalloc((void**)&h->d_parts3, 256 * 8 * sizeof(float)));
    int sb = stream_blocks(n);
    if (sb > 256) sb = 256;
    int* flag = reinterpret_cast<int*>(static_cast<char*>(h->h_small) + 192);
    h->publish_seq = h->publish_seq == 0x7fffffff ? 1 : h->publish_seq + 1;
    hipLaunchKernelGGL(k_bounds3_parts, dim3(sb), dim3(kBlock), 0, h->stream, dx, dy, dz, n, h->d_parts3, (unsigned int*)nullptr, 0, 0, 0);
    hipLaunchKernelGGL(k_bounds3_publish, dim3(1), dim3(64), 0, h->stream, (const float*)h->d_parts3, sb, hb, flag, h->publish_seq);
    HIP_TRY(hipGetLastError());
    bool seen = false;
    const int want = h->publish_seq;
    HIP_TRY(spin_until(h->stream, [&]() { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want; }, &seen));
    if (!seen) {                                           // safety net: the atomic form with a copy each way
      for (int a = 0; a < 3; ++a) { hb[2 * a] = 0xFFFFFFFFu; hb[2 * a + 1] = 0u; }
      HIP_TRY(hipMemcpyAsync(h->d_bounds, hb, 24, hipMemcpyHostToDevice, h->stream));
      hipLaunchKernelGGL(k_bounds3, dim3(sb > kBoundsBlocks ? kBoundsBlocks : sb), dim3(kBlock), 0, h->stream, dx, dy, dz, n, h->d_bounds);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(hb, h->d_bounds, 24, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
  }
  if (hb[0] == 0xFFFFFFFFu || hb[1] == 0u) { set_error("target has no finite point"); return NDT_ERR_INVALID_ARG; }
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = ordered_to_float(hb[2 * a]); hi[a] = ordered_to_float(hb[2 * a + 1]); }
  { const int32_t gs = setup_geometry3(h, lo, hi); if (gs != NDT_OK) return gs; }
  const int32_t as = accumulate3(h, dx, dy, dz, n, /*merge=*/false, nullptr);
  if (as != NDT_OK) return as;
  h->has_target = true;
  return upload_static3(h);
}

int32_t ensure_graph3(ndt3d_handle* h, int launches) {
  using namespace ndt;
  const bool newton = h->prm.hessian_mode == NDT_HESSIAN_NEWTON;
  HIP_TRY(h->graphs.get(newton ? (const void*)&k_iterate3<1> : (const void*)&k_iterate3<0>, dim3(kMaxBlocks), dim3(kBlock),
                        (void*)h->d_static, (void*)h->d_call, (void*)h->d_dyn, launches, h->prm.hessian_mode, h->stream,
                        &h->graph_exec));
  return NDT_OK;
}

// waits for the alignment in flight, if any; the final state is in h->h_state afterwards
int32_t finish_align3(ndt3d_handle* h) {
  using namespace ndt;
  const int kind = h->in_flight;
  h->in_flight = 0;
  if (kind == 1) {
    HIP_TRY(hipStreamSynchronize(h->stream));
  } else if (kind == 2) {
    bool seen = false;
    HIP_TRY(chunk_run_finish(h->chunk_run, h->stream, h->h_flag, &seen));
    HIP_TRY(hipGetLastError());
    if (!seen) { ndt::set_error("the Gauss-Newton loop did not report its end"); return NDT_ERR_HIP; }
  }
  return NDT_OK;
}

// enqueues the loop; finish_align3 leaves the final state in h->h_state
int32_t begin_align3(ndt3d_handle* h, const float* dx, const float* dy, const float* dz, size_t n, const double* pose,
                     int fixed_override) {
  using namespace ndt;
  TraceRange range("ndt3d_align: Gauss-Newton loop");
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  // a converged-mode loop needs the host to keep it fed: finish it.  A fixed-K chain in flight is
  // simply followed on the stream (the caller gave up its result by not fetching it).
  if (h->in_flight == 2) { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  if (n == 0 || n > kMaxSourcePoints) return NDT_ERR_INVALID_ARG;
  if (h->n_valid < 1) {
    std::memset(h->h_state, 0, sizeof(IterState3));
    for (int j = 0; j < 6; ++j) h->h_state->pose[j] = pose[j];
    h->h_state->status = NDT_TOO_FEW_CELLS;
    return NDT_OK;
  }
  const int fixed = fixed_override >= 0 ? fixed_override : h->prm.fixed_iterations;
  __atomic_store_n(&h->h_flag[0], 0, __ATOMIC_RELAXED);
  __atomic_store_n(&h->h_flag[1], 0, __ATOMIC_RELAXED);
  h->call_seq = h->call_seq == 0x7fffffff ? 1 : h->call_seq + 1;
  hipLaunchKernelGGL(k_begin3, dim3(1), dim3(64), 0, h->stream, h->d_call, h->d_dyn, dx, dy, dz, (int)n, pose[0], pose[1],
                     pose[2], pose[3], pose[4], pose[5], fixed, fixed > 0 ? (IterState3*)nullptr : h->h_state,
                     fixed > 0 ? (int*)nullptr : h->h_flag, h->call_seq);
  const int K = fixed > 0 ? fixed : h->prm.max_iterations;
  if (fixed > 0) {
    const int32_t gs = ensure_graph3(h, K + 1);
    if (gs != NDT_OK) return gs;
    HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->h_state, &h->d_dyn->state[K & 1], sizeof(IterState3), hipMemcpyDeviceToHost, h->stream));
    h->in_flight = 1;
  } else {
    const int chunk = 8;
    const int32_t gs = ensure_graph3(h, chunk);
    if (gs != NDT_OK) return gs;
    h->chunk_run.drain = true;
    h->chunk_run.seq = h->call_seq;
    HIP_TRY(chunk_run_begin(h->chunk_run, h->graph_exec, h->stream, chunk, K + 1));
    h->in_flight = 2;
  }
  return NDT_OK;
}

int32_t run_align3(ndt3d_handle* h, const float* dx, const float* dy, const float* dz, size_t n, const double* pose,
                   int fixed_override) {
  const int32_t st = begin_align3(h, dx, dy, dz, n, pose, fixed_override);
  return st != NDT_OK ? st : finish_align3(h);
}

void unpack_h21(const double* s, double* H);

void state3_to_result(const ndt::IterState3& s, ndt3d_result* out) {
  std::memset(out, 0, sizeof(*out));
  for (int j = 0; j < 6; ++j) { out->pose[j] = s.pose[j]; out->g[j] = s.g[j]; }
  unpack_h21(s.H, out->H);
  out->score = s.score; out->iterations = s.iter; out->n_hit = s.n_hit; out->status = s.status;
}

// m alignments against the cached voxel grid in one launch chain (ndt3d_multi.hpp)
int32_t multi_align3(ndt3d_handle* h, const float* const* sxs, const float* const* sys, const float* const* szs, const size_t* ns,
                     bool shared, const double* init_poses, int32_t m, ndt3d_result* results) {
  using namespace ndt;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  TraceRange range(shared ? "ndt3d_align_multi_start" : "ndt3d_align_multi_scan");
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  if (h->n_valid < 1) {
    for (int32_t k = 0; k < m; ++k) {
      std::memset(&results[k], 0, sizeof(ndt3d_result));
      for (int j = 0; j < 6; ++j) results[k].pose[j] = init_poses[6 * k + j];
      results[k].status = NDT_TOO_FEW_CELLS;
    }
    return NDT_OK;
  }
  if (!h->h_state_multi) HIP_TRY(hipHostMalloc((void**)&h->h_state_multi, kMaxStarts3 * sizeof(IterState3), hipHostMallocDefault));
  if (!h->d_dyn_multi) {
    HIP_TRY(hipMalloc((void**)&h->d_dyn_multi, sizeof(AlignDynMulti3)));
    HIP_TRY(hipMemsetAsync(h->d_dyn_multi, 0, sizeof(AlignDynMulti3), h->stream));
  }
  const bool newton = h->prm.hessian_mode == NDT_HESSIAN_NEWTON;
  const int fixed = h->prm.fixed_iterations;
  const int K = fixed > 0 ? fixed : h->prm.max_iterations;
  const bool converged_mode = fixed == 0;
  __atomic_store_n(&h->h_flag[0], 0, __ATOMIC_RELAXED);
  __atomic_store_n(&h->h_flag[1], 0, __ATOMIC_RELAXED);
  h->call_seq = h->call_seq == 0x7fffffff ? 1 : h->call_seq + 1;
  StartPoses3 sp{};
  StartScans3 sc{};
  for (int k = 0; k < m; ++k) {
    for (int j = 0; j < 6; ++j) sp.p[k][j] = init_poses[6 * k + j];
    const int q = shared ? 0 : k;
    sc.sx[k] = sxs[q]; sc.sy[k] = sys[q]; sc.sz[k] = szs[q]; sc.n[k] = (int)ns[q];
  }
  hipLaunchKernelGGL(k_begin_multi3_scans, dim3(1), dim3(64), 0, h->stream, h->d_dyn_multi, sc, (int)m);
  hipLaunchKernelGGL(k_begin_multi3, dim3(1), dim3(64), 0, h->stream, h->d_call, h->d_dyn_multi, sp, (int)m, fixed,
                     converged_mode ? h->h_state_multi : (IterState3*)nullptr, converged_mode ? h->h_flag : (int*)nullptr,
                     h->call_seq);
  HIP_TRY(hipGetLastError());
  const int chunk = 8;
  const int steps = converged_mode ? chunk : K + 1;
  hipGraphExec_t exec = nullptr;
  const void* fs = newton ? (const void*)&k_multi_solve3<1> : (const void*)&k_multi_solve3<0>;
  const void* fb = newton ? (const void*)&k_multi_body3<1> : (const void*)&k_multi_body3<0>;
  // launch shapes in powers of two (slots >= m are born finished: their workgroups return at once), so that a caller
  // whose m varies from call to call replays one of seven cached graphs instead of instantiating a new one each time
  int mg = 1;
  while (mg < m) mg <<= 1;
  HIP_TRY(h->graphs.get2(fs, dim3(mg), dim3(kBlock), fb, dim3(kMaxBlocks, mg), dim3(kBlock), (void*)h->d_static, (void*)h->d_call,
                         (void*)h->d_dyn_multi, steps, 0x100000 | (mg << 8) | h->prm.hessian_mode, h->stream, &exec));
  if (converged_mode) {
    bool seen = false;
    HIP_TRY(run_chunks_until_flag(exec, h->stream, h->h_flag, steps, K + 1, h->call_seq, &seen));
    HIP_TRY(hipGetLastError());
    if (!seen) { set_error("the 3D multi-scan loop did not report its end"); return NDT_ERR_HIP; }
  } else {
    HIP_TRY(hipGraphLaunch(exec, h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_state_multi, h->d_dyn_multi->state[K & 1], kMaxStarts3 * sizeof(IterState3), hipMemcpyDeviceToHost,
                           h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  for (int k = 0; k < m; ++k) state3_to_result(h->h_state_multi[k], &results[k]);
  return NDT_OK;
}

void unpack_h21(const double* s, double* H) {
  H[0] = s[0]; H[1] = s[1]; H[2] = s[2]; H[7] = s[3]; H[8] = s[4]; H[14] = s[5];
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) H[6 * r + 3 + k] = s[6 + 3 * r + k];
  H[21] = s[15]; H[22] = s[16]; H[23] = s[17]; H[28] = s[18]; H[29] = s[19]; H[35] = s[20];
  for (int r = 0; r < 6; ++r) for (int c = 0; c < r; ++c) H[6 * r + c] = H[6 * c + r];
}

}  // namespace

extern "C" {

void ndt3d_default_params(ndt3d_params* p) {
  if (!p) return;
  ndt2d_default_params(p);
  p->cell_size = 1.0;
  p->min_points = 5;
  p->step_max_trans = 1.0;
  p->min_hits = 6;
}

int32_t ndt3d_create(const ndt3d_params* p, int32_t device_id, ndt3d_handle** out) {
  if (!out) return NDT_ERR_INVALID_ARG;
  *out = nullptr;
  const int32_t st = check_params(p);
  if (st != NDT_OK) return st;
  if (p->overlap_grids == 4) { set_error("overlapping grids are a 2D option"); return NDT_ERR_INVALID_ARG; }
  const int ndev = ndt_device_count();
  if (ndev <= 0) { set_error("no HIP device visible: this library has no CPU fallback"); return NDT_ERR_NO_DEVICE; }
  if (device_id < 0 || device_id >= ndev) return NDT_ERR_INVALID_ARG;
  ndt3d_handle* h = new (std::nothrow) ndt3d_handle();
  if (!h) return NDT_ERR_ALLOC;
  h->device = device_id;
  h->prm = *p;
  auto fail = [&](int32_t code) { ndt3d_destroy(h); return code; };
  if (hipSetDevice(device_id) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipMalloc((void**)&h->d_bounds, 32) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_counters, ndt::kCountInts * sizeof(int)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_outside, sizeof(unsigned long long)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_static, sizeof(ndt::AlignStatic3)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_call, sizeof(ndt::AlignCall3)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMalloc((void**)&h->d_dyn, sizeof(ndt::AlignDyn3)) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipHostMalloc((void**)&h->h_static, sizeof(ndt::AlignStatic3), hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipEventCreateWithFlags(&h->upload_ev, hipEventDisableTiming) != hipSuccess) return fail(NDT_ERR_HIP);
  if (hipHostMalloc((void**)&h->h_state, sizeof(ndt::IterState3), hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipHostMalloc((void**)&h->h_flag, 64, hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  *h->h_flag = 0;
  if (hipHostMalloc(&h->h_small, 256, hipHostMallocDefault) != hipSuccess) return fail(NDT_ERR_ALLOC);
  if (hipMemset(h->d_dyn, 0, sizeof(ndt::AlignDyn3)) != hipSuccess) return fail(NDT_ERR_HIP);
  *out = h;
  return NDT_OK;
}

int32_t ndt3d_destroy(ndt3d_handle* h) {
  if (!h) return NDT_OK;
  (void)hipSetDevice(h->device);
  (void)finish_align3(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->graphs.clear();
  void* dev[] = {h->d_parts3, h->d_bounds, h->d_counters, h->d_outside, h->d_static, h->d_call, h->d_dyn, h->d_t[0], h->d_t[1], h->d_t[2],
                 h->d_s[0], h->d_s[1], h->d_s[2], h->d_b[0], h->d_b[1], h->d_b[2], h->d_tiles, h->d_split3, h->grid.rec, h->grid.acc, h->d_dyn_multi};
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = {h->h_static, h->h_state, h->h_small, h->h_flag, h->h_state_multi, h->h_pub3};
  for (void* p : host) if (p) (void)hipHostFree(p);
  if (h->upload_ev) (void)hipEventDestroy(h->upload_ev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return NDT_OK;
}

int32_t ndt3d_set_tuning(ndt3d_handle* h, int32_t knob, int64_t value) {
  if (!h || knob != NDT_TUNE_SINGLE_SYNC_BUILD) return NDT_ERR_INVALID_ARG;
  h->one_round_trip = value != 0;
  return NDT_OK;
}

int32_t ndt3d_wait_stream(ndt3d_handle* h, void* producer_stream) {
  if (!h) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(ndt::order_after(h->stream, (hipStream_t)producer_stream));
  return NDT_OK;
}

int32_t ndt3d_add_target_points(ndt3d_handle* h, const float* x, const float* y, const float* z, size_t n, size_t* n_outside) {
  if (!h || !x || !y || !z || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  const int32_t st = ensure3(h->d_t, &h->tcap, n);
  if (st != NDT_OK) return st;
  const float* src[3] = {x, y, z};
  for (int a = 0; a < 3; ++a) HIP_TRY(hipMemcpyAsync(h->d_t[a], src[a], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  unsigned long long outside = 0;
  const int32_t fs = accumulate3(h, h->d_t[0], h->d_t[1], h->d_t[2], n, /*merge=*/true, &outside);
  if (n_outside) *n_outside = (size_t)outside;
  if (fs != NDT_OK) { h->has_target = false; return fs; }
  return NDT_OK;          // geometry, storage and parameters are unchanged: the device context stays as it is
}

int32_t ndt3d_range_image_to_points_dev(const float* d_ranges, int32_t n_elev, int32_t n_azim, const double* elevations,
                                        double azimuth0, double azimuth_inc, double range_min, double range_max, float* d_x,
                                        float* d_y, float* d_z, void* stream) {
  if (!d_ranges || !elevations || !d_x || !d_y || !d_z || n_elev < 1 || n_elev > ndt::kMaxRings || n_azim < 1 ||
      !std::isfinite(azimuth0) || !std::isfinite(azimuth_inc))
    return NDT_ERR_INVALID_ARG;
  ndt::RingTable rings{};
  for (int e = 0; e < n_elev; ++e) {
    if (!std::isfinite(elevations[e])) return NDT_ERR_INVALID_ARG;
    rings.cs[2 * e] = std::cos(elevations[e]); rings.cs[2 * e + 1] = std::sin(elevations[e]);
  }
  const size_t n = (size_t)n_elev * (size_t)n_azim;
  hipLaunchKernelGGL(ndt::k_range_image_to_points, dim3(stream_blocks(n)), dim3(ndt::kBlock), 0, (hipStream_t)stream, d_ranges,
                     n_elev, n_azim, rings, azimuth0, azimuth_inc, (float)range_min, (float)range_max, d_x, d_y, d_z);
  HIP_TRY(hipGetLastError());
  return NDT_OK;
}

int32_t ndt3d_reserve_target(ndt3d_handle* h, const double lo[3], const double hi[3]) {
  if (!h || !lo || !hi) return NDT_ERR_INVALID_ARG;
  float l[3], u[3];
  for (int a = 0; a < 3; ++a) {
    if (!(lo[a] <= hi[a]) || !std::isfinite(lo[a]) || !std::isfinite(hi[a])) return NDT_ERR_INVALID_ARG;
    l[a] = (float)lo[a]; u[a] = (float)hi[a];
  }
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  h->has_target = false;
  { const int32_t gs = setup_geometry3(h, l, u); if (gs != NDT_OK) return gs; }
  const size_t ncell = (size_t)h->grid.W * h->grid.H * h->grid.D;
  HIP_TRY(hipMemsetAsync(h->grid.acc, 0, ncell * sizeof(ndt::CellAcc3), h->stream));
  HIP_TRY(hipMemsetAsync(h->grid.rec, 0, 4 * ncell * sizeof(float4), h->stream));
  h->n_valid = 0;
  h->has_target = true;
  return upload_static3(h);
}

int32_t ndt3d_add_target_points_dev(ndt3d_handle* h, const float* d_x, const float* d_y, const float* d_z, size_t n,
                                    const double pose[6], size_t* n_outside, void* stream) {
  if (!h || !d_x || !d_y || !d_z || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  if (stream) HIP_TRY(ndt::order_after(h->stream, (hipStream_t)stream));
  const float* p[3] = {d_x, d_y, d_z};
  ndt::Rigid3F T;
  if (pose) {       // the points are moved on the way into the build (count and scatter passes), not by a kernel of their own
    const double ca = std::cos(pose[3]), sa = std::sin(pose[3]), cb = std::cos(pose[4]), sb = std::sin(pose[4]),
                 cg = std::cos(pose[5]), sg = std::sin(pose[5]);
    const double R[9] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa,
                         sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa,
                         -sb, cb * sa, cb * ca};
    for (int j = 0; j < 9; ++j) T.r[j] = (float)R[j];
    for (int j = 0; j < 3; ++j) T.t[j] = (float)pose[j];
  }
  unsigned long long outside = 0;
  const int32_t fs = accumulate3(h, p[0], p[1], p[2], n, /*merge=*/true, &outside, pose ? &T : nullptr);
  if (n_outside) *n_outside = (size_t)outside;
  if (fs != NDT_OK) { h->has_target = false; return fs; }
  return NDT_OK;
}

int32_t ndt3d_set_target_dev(ndt3d_handle* h, const float* d_x, const float* d_y, const float* d_z, size_t n, void* stream) {
  if (!h || !d_x || !d_y || !d_z || n == 0) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  if (stream) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   // producer of the device arrays
  return set_target3_impl(h, d_x, d_y, d_z, n);
}

int32_t ndt3d_set_target(ndt3d_handle* h, const float* x, const float* y, const float* z, size_t n) {
  if (!h || !x || !y || !z || n == 0) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  const int32_t st = ensure3(h->d_t, &h->tcap, n);
  if (st != NDT_OK) return st;
  const float* src[3] = {x, y, z};
  for (int a = 0; a < 3; ++a) HIP_TRY(hipMemcpyAsync(h->d_t[a], src[a], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  return set_target3_impl(h, h->d_t[0], h->d_t[1], h->d_t[2], n);
}

int32_t ndt3d_get_grid_info(ndt3d_handle* h, ndt3d_grid_info* info) {
  if (!h || !info) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  info->ox = h->grid.ox; info->oy = h->grid.oy; info->oz = h->grid.oz; info->inv_cell = h->grid.inv_c;
  info->width = h->grid.W; info->height = h->grid.H; info->depth = h->grid.D; info->n_valid = h->n_valid;
  return NDT_OK;
}

int32_t ndt3d_get_grid(ndt3d_handle* h, int32_t* count, float* mean_xyz, float* icov6) {
  if (!h) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  const size_t nc = (size_t)h->grid.W * h->grid.H * h->grid.D;
  float4* rec = new (std::nothrow) float4[4 * nc];
  ndt::CellAcc3* acc = count ? new (std::nothrow) ndt::CellAcc3[nc] : nullptr;
  int32_t rc = (!rec || (count && !acc)) ? NDT_ERR_ALLOC : NDT_OK;
  if (rc == NDT_OK && hipMemcpyAsync(rec, h->grid.rec, 4 * nc * sizeof(float4), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = NDT_ERR_HIP;
  if (rc == NDT_OK && acc && hipMemcpyAsync(acc, h->grid.acc, nc * sizeof(ndt::CellAcc3), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = NDT_ERR_HIP;
  if (rc == NDT_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = NDT_ERR_HIP;
  if (rc == NDT_OK) {
    for (size_t k = 0; k < nc; ++k) {
      const float4 a = rec[4 * k], b = rec[4 * k + 1], c = rec[4 * k + 2];
      const bool valid = a.w > 0.f;
      if (count) count[k] = (int32_t)acc[k].n;
      if (mean_xyz) { mean_xyz[3 * k] = valid ? a.x : 0.f; mean_xyz[3 * k + 1] = valid ? a.y : 0.f; mean_xyz[3 * k + 2] = valid ? a.z : 0.f; }
      if (icov6) {
        icov6[6 * k] = b.x; icov6[6 * k + 1] = b.y; icov6[6 * k + 2] = b.z; icov6[6 * k + 3] = b.w;
        icov6[6 * k + 4] = c.x; icov6[6 * k + 5] = c.y;
      }
    }
  }
  delete[] rec; delete[] acc;
  return rc;
}

static int32_t upload_source3(ndt3d_handle* h, const float* sx, const float* sy, const float* sz, size_t n) {
  { const int32_t fs = finish_align3(h); if (fs != NDT_OK) return fs; }
  const int32_t st = ensure3(h->d_s, &h->scap, n);
  if (st != NDT_OK) return st;
  const float* src[3] = {sx, sy, sz};
  for (int a = 0; a < 3; ++a) HIP_TRY(hipMemcpyAsync(h->d_s[a], src[a], n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  return NDT_OK;
}

int32_t ndt3d_evaluate(ndt3d_handle* h, const float* sx, const float* sy, const float* sz, size_t n,
                       const double pose[6], ndt3d_eval* out) {
  if (!h || !sx || !sy || !sz || !pose || !out || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  int32_t st = upload_source3(h, sx, sy, sz, n);
  if (st != NDT_OK) return st;
  st = run_align3(h, h->d_s[0], h->d_s[1], h->d_s[2], n, pose, 1);
  if (st != NDT_OK) return st;
  std::memset(out, 0, sizeof(*out));
  unpack_h21(h->h_state->H, out->H);
  for (int j = 0; j < 6; ++j) out->g[j] = h->h_state->g[j];
  out->score = h->h_state->score;
  out->n_hit = h->h_state->n_hit;
  return NDT_OK;
}

int32_t ndt3d_evaluate_dev(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                           const double pose[6], ndt3d_eval* out) {
  if (!h || !d_sx || !d_sy || !d_sz || !pose || !out || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  const int32_t st = run_align3(h, d_sx, d_sy, d_sz, n, pose, 1);
  if (st != NDT_OK) return st;
  std::memset(out, 0, sizeof(*out));
  unpack_h21(h->h_state->H, out->H);
  for (int j = 0; j < 6; ++j) out->g[j] = h->h_state->g[j];
  out->score = h->h_state->score;
  out->n_hit = h->h_state->n_hit;
  return NDT_OK;
}

int32_t ndt3d_align_dev_async(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                              const double init_pose[6]) {
  if (!h || !d_sx || !d_sy || !d_sz || !init_pose) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  return begin_align3(h, d_sx, d_sy, d_sz, n, init_pose, -1);
}

int32_t ndt3d_align_finish(ndt3d_handle* h, ndt3d_result* out) {
  if (!h || !out) return NDT_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const int32_t st = finish_align3(h);
  if (st != NDT_OK) return st;
  const ndt::IterState3& s = *h->h_state;
  std::memset(out, 0, sizeof(*out));
  for (int j = 0; j < 6; ++j) { out->pose[j] = s.pose[j]; out->g[j] = s.g[j]; }
  unpack_h21(s.H, out->H);
  out->score = s.score; out->iterations = s.iter; out->n_hit = s.n_hit; out->status = s.status;
  return NDT_OK;
}

// Per-iteration trace, as ndt2d_align_trace: one plain k_iterate3 launch and one state fetch per iteration.
int32_t ndt3d_align_trace(ndt3d_handle* h, const float* sx, const float* sy, const float* sz, size_t n,
                          const double init_pose[6], ndt3d_result* rows, int32_t capacity, int32_t* n_rows, ndt3d_result* out) {
  using namespace ndt;
  if (!h || !sx || !sy || !sz || !init_pose || !rows || capacity < 1 || !n_rows || n == 0 || n > kMaxSourcePoints)
    return NDT_ERR_INVALID_ARG;
  *n_rows = 0;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  TraceRange range("ndt3d_align_trace");
  HIP_TRY(hipSetDevice(h->device));
  auto to_row = [](const IterState3& s, ndt3d_result* r) {
    std::memset(r, 0, sizeof(*r));
    for (int j = 0; j < 6; ++j) { r->pose[j] = s.pose[j]; r->g[j] = s.g[j]; }
    unpack_h21(s.H, r->H);
    r->score = s.score; r->iterations = s.iter; r->n_hit = s.n_hit; r->status = s.status;
  };
  if (h->n_valid < 1) {
    std::memset(&rows[0], 0, sizeof(ndt3d_result));
    for (int j = 0; j < 6; ++j) rows[0].pose[j] = init_pose[j];
    rows[0].status = NDT_TOO_FEW_CELLS;
    if (out) *out = rows[0];
    return NDT_OK;
  }
  { const int32_t us = upload_source3(h, sx, sy, sz, n); if (us != NDT_OK) return us; }
  const int fixed = h->prm.fixed_iterations;
  const int K = fixed > 0 ? fixed : h->prm.max_iterations;
  const bool newton = h->prm.hessian_mode == NDT_HESSIAN_NEWTON;
  h->call_seq = h->call_seq == 0x7fffffff ? 1 : h->call_seq + 1;
  hipLaunchKernelGGL(k_begin3, dim3(1), dim3(64), 0, h->stream, h->d_call, h->d_dyn, h->d_s[0], h->d_s[1], h->d_s[2], (int)n,
                     init_pose[0], init_pose[1], init_pose[2], init_pose[3], init_pose[4], init_pose[5], fixed,
                     (IterState3*)nullptr, (int*)nullptr, h->call_seq);
  for (int k = 0; k <= K; ++k) {
    if (newton) hipLaunchKernelGGL((k_iterate3<1>), dim3(kMaxBlocks), dim3(kBlock), 0, h->stream, h->d_static, h->d_call, h->d_dyn, k & 1);
    else hipLaunchKernelGGL((k_iterate3<0>), dim3(kMaxBlocks), dim3(kBlock), 0, h->stream, h->d_static, h->d_call, h->d_dyn, k & 1);
    HIP_TRY(hipGetLastError());
    if (k == 0) continue;                                  // launch 0 only evaluates
    HIP_TRY(hipMemcpyAsync(h->h_state, &h->d_dyn->state[k & 1], sizeof(IterState3), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (*n_rows < capacity) to_row(*h->h_state, &rows[(*n_rows)++]);
    if (h->h_state->done) break;
  }
  if (out) to_row(*h->h_state, out);
  return NDT_OK;
}

int32_t ndt3d_align_multi_scan_dev(ndt3d_handle* h, const float* const* d_sx, const float* const* d_sy, const float* const* d_sz,
                                   const size_t* n, const double* init_poses, int32_t m, ndt3d_result* results) {
  if (!h || !d_sx || !d_sy || !d_sz || !n || !init_poses || !results || m < 1 || m > ndt::kMaxStarts3) return NDT_ERR_INVALID_ARG;
  for (int32_t k = 0; k < m; ++k)
    if (!d_sx[k] || !d_sy[k] || !d_sz[k] || n[k] == 0 || n[k] > kMaxSourcePoints) return NDT_ERR_INVALID_ARG;
  return multi_align3(h, d_sx, d_sy, d_sz, n, /*shared=*/false, init_poses, m, results);
}

int32_t ndt3d_align_multi_start_dev(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                                    const double* init_poses, int32_t m, ndt3d_result* results) {
  if (!h || !d_sx || !d_sy || !d_sz || !init_poses || !results || m < 1 || m > ndt::kMaxStarts3) return NDT_ERR_INVALID_ARG;
  if (n == 0 || n > kMaxSourcePoints) return NDT_ERR_INVALID_ARG;
  return multi_align3(h, &d_sx, &d_sy, &d_sz, &n, /*shared=*/true, init_poses, m, results);
}

void* ndt3d_stream(ndt3d_handle* h) { return h ? (void*)h->stream : nullptr; }

int32_t ndt3d_align_dev(ndt3d_handle* h, const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                        const double init_pose[6], ndt3d_result* out) {
  if (!h || !d_sx || !d_sy || !d_sz || !init_pose || !out) return NDT_ERR_INVALID_ARG;
  const int32_t st = ndt3d_align_dev_async(h, d_sx, d_sy, d_sz, n, init_pose);
  if (st != NDT_OK) return st;
  return ndt3d_align_finish(h, out);
}

int32_t ndt3d_align(ndt3d_handle* h, const float* sx, const float* sy, const float* sz, size_t n,
                    const double init_pose[6], ndt3d_result* out) {
  if (!h || !sx || !sy || !sz || !init_pose || !out || n == 0) return NDT_ERR_INVALID_ARG;
  if (!h->has_target) return NDT_ERR_NO_TARGET;
  HIP_TRY(hipSetDevice(h->device));
  const int32_t st = upload_source3(h, sx, sy, sz, n);
  if (st != NDT_OK) return st;
  return ndt3d_align_dev(h, h->d_s[0], h->d_s[1], h->d_s[2], n, init_pose, out);
}

}  // extern "C"
