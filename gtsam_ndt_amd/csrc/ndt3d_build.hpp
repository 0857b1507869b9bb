// Binned 3D voxel-grid build: the 3D twin of ndt2d_build.hpp with 4 x 4 x 4-voxel tiles.  The
// atomic path (k_accumulate3) suffers from same-address contention on floor / ceiling voxels
// (thousands of points per voxel); here that contention is confined to LDS.  The tile is small
// because a lidar scan puts tens of thousands of points into the few tiles around the sensor and
// one workgroup (one CU's LDS atomic unit) owns a tile: 8 x 8 x 4 tiles took 0.235 ms on the
// densest one (131k-point scan), 4 x 4 x 4 spreads it over 4x as many CUs (set_target 0.40 ->
// 0.28 ms); 2 x 2 x 4 gains little more (0.24) and quarters the map volume the 8192-tile
// histogram covers.
#pragma once
#include "ndt2d_build.hpp"
#include "ndt3d_kernels.hpp"

namespace ndt {

constexpr int kT3x = 2, kT3y = 2, kT3z = 2;                        // log2 tile edge: 4 x 4 x 4
constexpr int kTile3Cells = 1 << (kT3x + kT3y + kT3z);             // 64
#ifndef NDT_TILE3_SPLIT
#define NDT_TILE3_SPLIT 8
#define NDT_TILE3_SUBMIN 1024
#endif
constexpr int kTile3Split = NDT_TILE3_SPLIT;                       // workgroups that may share a tile (k_tile_accumulate3)
constexpr int kTile3SubMin = NDT_TILE3_SUBMIN;                     // ... one per this many points of the tile

struct BinGeom3 {
  float ox, oy, oz, inv_c;
  int W, H, D, ntx, nty, ntile;
};

__device__ __forceinline__ int tile_of3(const BinGeom3& g, float px, float py, float pz) {
  const float fx = (px - g.ox) * g.inv_c, fy = (py - g.oy) * g.inv_c, fz = (pz - g.oz) * g.inv_c;
  const bool in = (fx >= 0.f) & (fx < (float)g.W) & (fy >= 0.f) & (fy < (float)g.H) & (fz >= 0.f) & (fz < (float)g.D);
  return in ? ((((int)fz >> kT3z) * g.nty + ((int)fy >> kT3y)) * g.ntx + ((int)fx >> kT3x)) : -1;
}

// ---- a build whose geometry is decided on the device (one host round trip instead of two: ndt3d_api.hpp
// set_target3_single_sync, the 3D twin of the 2D build's single-sync path) ---------------------------------------------
// The decision lives in the pad of the build's accumulator block (words kGeom3Word ..), so that ONE publish brings the
// counters, the outside count, the bounding box and the geometry back.
struct GeomDev3 {
  BinGeom3 bin;               // 10 words
  int ok;                     // 0: no finite point, or the grid fits neither the handle's storage nor the launch bounds
  unsigned int bounds[6];     // ordered-float bounding box, for the host
};
constexpr int kGeom3Word = 42;
static_assert(sizeof(GeomDev3) == 17 * 4 && kGeom3Word + 17 <= 64, "GeomDev3 sits in the pad of the accumulator block");

// a1 in 3D on the device: the arithmetic of setup_geometry3() on the host (oracle/ndt3d.py grid_geometry3), which
// recomputes it from the same bounds afterwards and compares.  One lane.
__device__ inline bool decide_geometry3(const float mn[3], const float mx[3], double c, unsigned long long cell_capacity,
                                        int tile_bound, BinGeom3* bin, Grid3Dev* grid) {
#pragma clang fp contract(off)
  const float inv_c = (float)(1.0 / c);
  float o[3];
  int dims[3];
  double ncell_d = 1.0;
  for (int a = 0; a < 3; ++a) {
    o[a] = (float)((floor((double)mn[a] / c) - 1.0) * c);
    const float f = (mx[a] - o[a]) * inv_c;
    const double k = floor((double)f);
    if (!(k >= 0.0) || k > 1e7) return false;
    dims[a] = (int)k + 2;
    ncell_d *= dims[a];
  }
  if (ncell_d > (double)cell_capacity) return false;       // (the handle's storage never exceeds the 2^27 cells of setup_geometry3)
  const int ntx = (dims[0] + (1 << kT3x) - 1) >> kT3x, nty = (dims[1] + (1 << kT3y) - 1) >> kT3y, ntz = (dims[2] + (1 << kT3z) - 1) >> kT3z;
  if ((long long)ntx * nty * ntz > (long long)tile_bound) return false;
  bin->ox = o[0]; bin->oy = o[1]; bin->oz = o[2]; bin->inv_c = inv_c;
  bin->W = dims[0]; bin->H = dims[1]; bin->D = dims[2]; bin->ntx = ntx; bin->nty = nty; bin->ntile = ntx * nty * ntz;
  grid->ox = o[0]; grid->oy = o[1]; grid->oz = o[2]; grid->inv_c = inv_c;
  grid->W = dims[0]; grid->H = dims[1]; grid->D = dims[2]; grid->pad = 0;
  grid->cell = c;
  grid->fix_scale = 4194304.0 / c;
  static_assert(kFixShift == 22, "fix_scale literal");
  return true;
}

// The geometry decision is the prologue of k_tile_count3 (as k_chunk_sort's in 2D): EVERY workgroup reduces the
// bounding-box partials of k_bounds3_parts (a few KB) and applies the rule - the same inputs, the same arithmetic, the same
// grid everywhere - and workgroup 0 also writes it where the later kernels, the alignments and the host read it.  A
// one-workgroup kernel of its own for this cost 4 us of the build plus a kernel boundary.
// the rigid motion of a submap update (k_transform_points3's arithmetic, bit for bit), applied on the way in by both passes
// over the cloud instead of by a kernel of its own (use = 0: none)
struct Move3Args {
  Rigid3F T;
  int use;
};

struct Geom3Args {
  const float* parts;          // null: the geometry comes with the launch (BinGeom3 argument)
  int nparts;
  int tile_bound;
  double cell;
  unsigned long long cell_capacity;
  Grid3Dev* grid;              // the device context's grid header
  GeomDev3* out;               // in the accumulator block (which k_bounds3_parts cleared, all but these words)
};

// wave 0 of a workgroup; returns ok, the geometry in *bin (valid in lane 0 only: broadcast through LDS by the caller)
__device__ __forceinline__ bool reduce_and_decide3(const Geom3Args& ga, BinGeom3* bin, bool write) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x & 63; i < ga.nparts; i += 64) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], ga.parts[8 * i + 2 * a]); mx[a] = fmaxf(mx[a], ga.parts[8 * i + 2 * a + 1]); }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
  bool ok = false;
  if ((threadIdx.x & 63) == 0) {
    bool none = false;
    unsigned int b[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool na = !(mn[a] <= mx[a]);                   // no finite point: k_bounds3's "empty" encoding
      none = none || na;
      b[2 * a] = na ? 0xFFFFFFFFu : float_to_ordered(mn[a]);
      b[2 * a + 1] = na ? 0u : float_to_ordered(mx[a]);
    }
    Grid3Dev scratch_grid;                                 // (only workgroup 0 writes the shared header)
    ok = !none && decide_geometry3(mn, mx, ga.cell, ga.cell_capacity, ga.tile_bound, bin, write ? ga.grid : &scratch_grid);
    if (write) {
#pragma unroll
      for (int j = 0; j < 6; ++j) ga.out->bounds[j] = b[j];
      ga.out->bin = *bin;
      ga.out->ok = ok ? 1 : 0;
    }
  }
  return ok;
}

// workgroups that share a tile of `points` points
__device__ __forceinline__ int tile3_subs(unsigned int points) {
  const int nsub = (int)((points + kTile3SubMin - 1) / kTile3SubMin);
  return nsub < 1 ? 1 : (nsub > kTile3Split ? kTile3Split : nsub);
}
static_assert(kBinMaxTiles <= (1 << 24) && kTile3Split <= 256, "wg_map packs tile | share << 24");

// What follows the count: the exclusive scan of the tile totals (starts and scatter cursors) and, beside it, of the
// workgroups each tile gets: wg_map lists the (tile, share) pairs one after the other - tile | share << 24 - so that
// k_tile_accumulate3 is launched with one workgroup per entry (at most ntile + n / kTile3SubMin) instead of
// ntile x kTile3Split, most of which had nothing to do and still had to be dispatched: 2904 workgroups for the 363 tiles
// of a config-5 scan, the last of them starting 15 us into the kernel (in-kernel clocks).
// Run by ONE workgroup of kBinThreads threads - the workgroup of k_tile_count3 that finishes last (a one-workgroup kernel
// of its own was 3.9 us of the build plus a kernel boundary); the totals were added by agent-scope atomics and are read
// with agent-scope loads.  ntile <= kBinMaxTiles.
struct Scan3Out {
  unsigned int* done;          // arrivals of k_tile_count3's workgroups (in the accumulator block: cleared with it)
  unsigned int* tile_start;    // [ntile + 1]
  unsigned int* tile_cursor;   // [ntile]
  unsigned int* wg_total;      // [1]
  unsigned int* wg_map;        // [ntile + n / kTile3SubMin + 1]
};
__device__ __forceinline__ void tile3_scan_block(const unsigned int* __restrict__ tile_total, int ntile, const Scan3Out& so) {
  __shared__ unsigned int s_wv[2][kBinThreads / 64];
  const int per = (ntile + kBinThreads - 1) / kBinThreads;
  const int t0 = threadIdx.x * per;
  unsigned int local = 0, lwg = 0;
  for (int k = 0; k < per; ++k)
    if (t0 + k < ntile) {
      const unsigned int c = __hip_atomic_load(tile_total + t0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      local += c; lwg += (unsigned int)tile3_subs(c);
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int inc = local, iwg = lwg;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int v = __shfl_up(inc, d, 64), w = __shfl_up(iwg, d, 64);
    if (lane >= d) { inc += v; iwg += w; }
  }
  if (lane == 63) { s_wv[0][wave] = inc; s_wv[1][wave] = iwg; }
  __syncthreads();
  unsigned int base = 0, bwg = 0;
  for (int w = 0; w < wave; ++w) { base += s_wv[0][w]; bwg += s_wv[1][w]; }
  unsigned int run = base + inc - local, rwg = bwg + iwg - lwg;
  for (int k = 0; k < per; ++k) {
    if (t0 + k < ntile) {
      const unsigned int c = __hip_atomic_load(tile_total + t0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      so.tile_start[t0 + k] = run;
      so.tile_cursor[t0 + k] = run;
      run += c;
      const int ns = tile3_subs(c);
      for (int sb = 0; sb < ns; ++sb) so.wg_map[rwg + sb] = (unsigned int)(t0 + k) | ((unsigned int)sb << 24);
      rwg += (unsigned int)ns;
    }
  }
  if (threadIdx.x == kBinThreads - 1) { so.tile_start[ntile] = run; *so.wg_total = rwg; }
}

__global__ __launch_bounds__(kBinThreads) void k_tile_count3(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ z, size_t n, BinGeom3 g,
                                                              unsigned int* __restrict__ tile_total,
                                                              unsigned long long* __restrict__ n_outside, Geom3Args ga,
                                                              Scan3Out so, Move3Args mv) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_hist[];
  __shared__ int s_last;
  if (ga.parts) {                                          // geometry decided here (the LDS covers the host's tile bound)
    __shared__ BinGeom3 s_bin;
    __shared__ int s_ok;
    if (threadIdx.x < 64) {
      BinGeom3 bin{};
      const bool ok = reduce_and_decide3(ga, &bin, blockIdx.x == 0);
      if (threadIdx.x == 0) { s_bin = bin; s_ok = ok ? 1 : 0; }
    }
    __syncthreads();
    if (!s_ok) return;                                     // uniform: the host repeats the build the usual way
    g = s_bin;
  }
  for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) s_hist[t] = 0u;
  __syncthreads();
  unsigned int outside = 0;
  const size_t stride = (size_t)gridDim.x * kBinThreads;
  for (size_t i = (size_t)blockIdx.x * kBinThreads + threadIdx.x; i < n; i += 4 * stride) {      // four points in flight
    float px[4], py[4], pz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t ii = i + u * stride;
      px[u] = ii < n ? x[ii] : NAN; py[u] = ii < n ? y[ii] : NAN; pz[u] = ii < n ? z[ii] : NAN;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (mv.use) apply_rigid3(mv.T, px[u], py[u], pz[u]);
      const int t = tile_of3(g, px[u], py[u], pz[u]);
      if (t >= 0) atomicAdd(&s_hist[t], 1u);
      else if (i + u * stride < n) outside++;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) {
    const unsigned int c = s_hist[t];
    if (c) atomicAdd(&tile_total[t], c);
  }
  if (n_outside && outside) atomicAdd(n_outside, (unsigned long long)outside);
  // the workgroup whose adds come last scans the totals (every adding wave drains its atomics, the workgroup meets, one
  // lane takes the ticket: MI355X_MICROARCH.md "Valid forms", agent atomics on both sides)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(so.done, 1u) == gridDim.x - 1u ? 1 : 0;
  __syncthreads();
  if (!s_last) return;                               // uniform
  tile3_scan_block(tile_total, g.ntile, so);
}

__global__ __launch_bounds__(kBinThreads) void k_tile_scatter3(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ z, size_t n, BinGeom3 g,
                                                                unsigned int* __restrict__ tile_cursor,
                                                                float* __restrict__ bx, float* __restrict__ by,
                                                                float* __restrict__ bz, const GeomDev3* __restrict__ dg,
                                                                Move3Args mv) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_mem[];
  if (dg) { if (!dg->ok) return; g = dg->bin; }
  unsigned int* s_hist = s_mem;
  unsigned int* s_base = s_mem + g.ntile;
  const size_t chunk = (size_t)kBinThreads * 4;
  for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
    for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) s_hist[t] = 0u;
    __syncthreads();
    float px[4], py[4], pz[4];
    int tile[4];
    unsigned int rank[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t i = base + (size_t)u * kBinThreads + threadIdx.x;
      px[u] = i < n ? x[i] : NAN; py[u] = i < n ? y[i] : NAN; pz[u] = i < n ? z[i] : NAN;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (mv.use) apply_rigid3(mv.T, px[u], py[u], pz[u]);            // (the moved point is what is stored below)
      tile[u] = tile_of3(g, px[u], py[u], pz[u]);
      rank[u] = tile[u] >= 0 ? atomicAdd(&s_hist[tile[u]], 1u) : 0u;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) {
      const unsigned int c = s_hist[t];
      s_base[t] = c ? atomicAdd(&tile_cursor[t], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (tile[u] >= 0) {
        const unsigned int dst = s_base[tile[u]] + rank[u];
        bx[dst] = px[u]; by[dst] = py[u]; bz[dst] = pz[u];
      }
    }
    __syncthreads();
  }
}

// What the workgroups that share a tile hand to the one that finishes it: every one of them leaves its 64 voxels' sums as
// a slab in a pool (write-through stores; slab = the workgroup's number), the last to arrive (a ticket per tile) adds the
// others' slabs to its own sums.  Round 2 added each share to the grid's sums with ten global atomics per voxel - 640 per
// workgroup, up to eight workgroups on the same 640 words, performed one after the other at the memory side - and read
// the totals back; that also needed the grid's sums cleared by a fill launch before every build.
constexpr int kCopies3 = 8;                                       // private copies of a tile's sums in k_tile_accumulate3's point loop
constexpr int kSlabWords = 9 * kTile3Cells + kTile3Cells / 2;       // 64-bit words: 9 sums per voxel, then the counts as u32 pairs
struct Split3Bufs {
  unsigned long long* pool;        // [capacity][kSlabWords]: slab b belongs to workgroup b of k_tile_accumulate3
  unsigned int capacity;           // slabs: the launch's workgroups (ntile + n / kTile3SubMin + 1)
};

#if defined(NDT_BUILD_PHASE_CLOCKS)
__device__ unsigned long long g_tile3_stamps[4096][8];       // tools-only: 100 MHz clock per phase, per workgroup
#define NDT_STAMP3(k) do { if (threadIdx.x == 0) g_tile3_stamps[blockIdx.x & 4095][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NDT_STAMP3(k) do { } while (0)
#endif

// per tile: LDS sums, then the finalise of k_finalise3 (shared device function below)
__device__ __forceinline__ bool finalise_sums3(const CellAcc3& c, double cx, double cy, double cz, double fix_scale,
                                               int min_points, double eig_ratio, float4& ra, float4& rb, float4& rc);

__global__ __launch_bounds__(kBinThreads) void k_tile_accumulate3(const float* __restrict__ bx, const float* __restrict__ by,
                                                                   const float* __restrict__ bz,
                                                                   const unsigned int* __restrict__ tile_start, Grid3Dev g,
                                                                   int ntx, int nty, int merge, int min_points,
                                                                   double eig_ratio, int* __restrict__ counters,
                                                                   unsigned int* __restrict__ tile_ticket, Split3Bufs sb,
                                                                   const unsigned int* __restrict__ wg_total,
                                                                   const unsigned int* __restrict__ wg_map,
                                                                   const GeomDev3* __restrict__ dg, const Grid3Dev* __restrict__ dgrid) {
  // The point loop adds into kCopies3 private copies of the tile's sums, chosen by lane: neighbouring points of a scan fall
  // into the same voxel, so a wave's 64 atomics went to one or two LDS words and were performed one after the other - the
  // fullest tiles' loops took 12-18 us (in-kernel clocks, tools/quick_tile3_stamps.py) and held up every other workgroup
  // on their CU's LDS.  With a copy per lane % 8 at most eight lanes meet on a word
  // (16 copies: 83 KB of LDS, one workgroup per CU - measured no faster).
  __shared__ __attribute__((aligned(16))) unsigned long long s_part[9 * kTile3Cells * kCopies3];    // [(sum, voxel)][copy]
  __shared__ unsigned int s_npart[kTile3Cells * kCopies3];
  __shared__ __attribute__((aligned(16))) unsigned long long s_sum[9][kTile3Cells];
  __shared__ unsigned int s_n[kTile3Cells];
  __shared__ int s_last;
  if (dg) {        // geometry from the device; storage (g.rec, g.acc) from the launch
    if (!dg->ok) return;
    ntx = dg->bin.ntx; nty = dg->bin.nty;
    g.ox = dgrid->ox; g.oy = dgrid->oy; g.oz = dgrid->oz; g.inv_c = dgrid->inv_c;
    g.W = dgrid->W; g.H = dgrid->H; g.D = dgrid->D; g.cell = dgrid->cell; g.fix_scale = dgrid->fix_scale;
  }
  if (blockIdx.x >= __builtin_amdgcn_readfirstlane(*wg_total)) return;       // (the launch covers the host's bound)
  const unsigned int entry = __builtin_amdgcn_readfirstlane(wg_map[blockIdx.x]);
  const int tile = (int)(entry & 0xFFFFFFu), sub = (int)(entry >> 24);
  const int tx0 = (tile % ntx) << kT3x, ty0 = ((tile / ntx) % nty) << kT3y, tz0 = (tile / (ntx * nty)) << kT3z;
  // A lidar scan puts thousands of points into the few tiles around the sensor, and one workgroup per tile left the
  // build waiting for the fullest one.  A tile of more than kTile3SubMin points is shared by up to kTile3Split workgroups
  // (grid.y): each sums its share in LDS and leaves it as a slab (Split3Bufs); the last one to arrive adds the slabs up
  // (exact integers: any split gives the same bits) and finalises the tile.
  const unsigned int p0 = tile_start[tile], p1 = tile_start[tile + 1];
  const int nsub = tile3_subs(p1 - p0);              // (sub < nsub: tile3_scan_block listed the shares by the same rule)
  NDT_STAMP3(0);
  const bool split = nsub > 1;
  // init: zeros, or the cached sums of this tile's voxels (merge = incremental submap update; a shared tile's cached
  // sums are added by the workgroup that finishes it)
  for (int c = threadIdx.x; c < kTile3Cells; c += kBinThreads) {
    const int ix = tx0 + (c & ((1 << kT3x) - 1)), iy = ty0 + ((c >> kT3x) & ((1 << kT3y) - 1)), iz = tz0 + (c >> (kT3x + kT3y));
    CellAcc3 a = {};
    if (merge && !split && ix < g.W && iy < g.H && iz < g.D) a = g.acc[((size_t)iz * g.H + iy) * g.W + ix];
    s_n[c] = a.n;
#pragma unroll
    for (int j = 0; j < 3; ++j) s_sum[j][c] = (unsigned long long)a.s[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) s_sum[3 + j][c] = (unsigned long long)a.ss[j];
  }
  for (int e = threadIdx.x; e < 9 * kTile3Cells * kCopies3; e += kBinThreads) s_part[e] = 0ull;
  for (int e = threadIdx.x; e < kTile3Cells * kCopies3; e += kBinThreads) s_npart[e] = 0u;
  __syncthreads();
  NDT_STAMP3(1);
  const int copy = threadIdx.x & (kCopies3 - 1);
  const unsigned int share = (p1 - p0 + nsub - 1) / nsub;
  const unsigned int q0 = p0 + sub * share, q1 = q0 + share < p1 ? q0 + share : p1;
  // 8 points in flight per thread: a one-point loop pays the memory latency on every trip
  constexpr int kU = 8;
  for (unsigned int i = q0 + threadIdx.x; i < q1; i += kBinThreads * kU) {
    float qx[kU], qy[kU], qz[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const unsigned int ii = i + u * kBinThreads;
      qx[u] = ii < q1 ? bx[ii] : 0.f;
      qy[u] = ii < q1 ? by[ii] : 0.f;
      qz[u] = ii < q1 ? bz[ii] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (i + u * kBinThreads < q1) {
        const float px = qx[u], py = qy[u], pz = qz[u];
        const int ix = (int)((px - g.ox) * g.inv_c), iy = (int)((py - g.oy) * g.inv_c), iz = (int)((pz - g.oz) * g.inv_c);
        const int ux = fix_coord(px, cell_centre(g.ox, ix, g.cell), g.fix_scale);
        const int uy = fix_coord(py, cell_centre(g.oy, iy, g.cell), g.fix_scale);
        const int uz = fix_coord(pz, cell_centre(g.oz, iz, g.cell), g.fix_scale);
        const int c = ((((iz - tz0) << kT3y) + (iy - ty0)) << kT3x) + (ix - tx0);
        unsigned long long* w = s_part + c * kCopies3 + copy;
        constexpr int kS = kTile3Cells * kCopies3;          // stride between the nine sums
        atomicAdd(&s_npart[c * kCopies3 + copy], 1u);
        atomicAdd(w, (unsigned long long)(long long)ux);
        atomicAdd(w + kS, (unsigned long long)(long long)uy);
        atomicAdd(w + 2 * kS, (unsigned long long)(long long)uz);
        atomicAdd(w + 3 * kS, prod64(ux, ux));
        atomicAdd(w + 4 * kS, prod64(ux, uy));
        atomicAdd(w + 5 * kS, prod64(ux, uz));
        atomicAdd(w + 6 * kS, prod64(uy, uy));
        atomicAdd(w + 7 * kS, prod64(uy, uz));
        atomicAdd(w + 8 * kS, prod64(uz, uz));
      }
    }
  }
  __syncthreads();
  // the copies -> the tile's sums (every (sum, voxel) word has one owner thread; the cached sums of a merge are in s_sum)
  for (int e = threadIdx.x; e < 9 * kTile3Cells; e += kBinThreads) {
    unsigned long long tot = 0ull;
#pragma unroll
    for (int k = 0; k < kCopies3; ++k) tot += s_part[e * kCopies3 + ((k + threadIdx.x) & (kCopies3 - 1))];   // (rotated: no two lanes on a bank)
    (&s_sum[0][0])[e] += tot;
  }
  if (threadIdx.x < kTile3Cells) {
    unsigned int tot = 0u;
#pragma unroll
    for (int k = 0; k < kCopies3; ++k) tot += s_npart[threadIdx.x * kCopies3 + ((k + threadIdx.x) & (kCopies3 - 1))];
    s_n[threadIdx.x] += tot;
  }
  __syncthreads();
  NDT_STAMP3(2);
  int nover = 0;
  if (split) {
    // [r3] No fences (MI355X_MICROARCH.md "Valid forms"): every handed-over word is stored write-through (agent-scope
    // atomic stores) and loaded with agent-scope loads, every storing wave drains its stores, the workgroup meets, one
    // lane takes the tile's ticket.  A __threadfence() here is a write-back AND an invalidate of the whole L2.
    // A share's slab is the one with its workgroup's number (the list of k_tile_count3's scan gives every (tile, share) its
    // own): no cursor to take, no table to look the others' slabs up in - the shares of a tile are neighbours in the list.
    const unsigned int slab = blockIdx.x;
    const bool have = slab < sb.capacity;              // (always: the host sizes the pool for the launch)
    if (have) {
      unsigned long long* dst = sb.pool + (size_t)slab * kSlabWords;
      const unsigned long long* src = &s_sum[0][0];
      for (int e = threadIdx.x; e < 9 * kTile3Cells; e += kBinThreads) __hip_atomic_store(dst + e, src[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x < kTile3Cells / 2)
        __hip_atomic_store(dst + 9 * kTile3Cells + threadIdx.x,
                           (unsigned long long)s_n[2 * threadIdx.x] | ((unsigned long long)s_n[2 * threadIdx.x + 1] << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&tile_ticket[tile], 1u) == (unsigned)(nsub - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;                             // uniform
    // the last of the tile's workgroups: the other shares (every thread adds its own words of every slab - no LDS atomics;
    // all loads of a thread are issued before the first add)
    {
      constexpr int kPer = (kSlabWords + kBinThreads - 1) / kBinThreads;       // 3 words per thread and slab
      unsigned long long w[kTile3Split][kPer];
      unsigned int idx[kTile3Split];
#pragma unroll
      for (int o = 0; o < kTile3Split; ++o)
        idx[o] = (o < nsub && o != sub) ? blockIdx.x - (unsigned)sub + (unsigned)o : 0xFFFFFFFEu;
#pragma unroll
      for (int o = 0; o < kTile3Split; ++o) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
          const int e = threadIdx.x + j * kBinThreads;
          w[o][j] = 0ull;
          if (idx[o] < sb.capacity && e < kSlabWords)
            w[o][j] = __hip_atomic_load(sb.pool + (size_t)idx[o] * kSlabWords + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (idx[o] != 0xFFFFFFFEu && idx[o] >= sb.capacity && threadIdx.x == 0) nover = 1;   // a pool smaller than the launch: reported, not overrun
      }
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        const int e = threadIdx.x + j * kBinThreads;
        unsigned long long tot = 0ull;
        unsigned int lo = 0u, hi = 0u;
#pragma unroll
        for (int o = 0; o < kTile3Split; ++o) { tot += w[o][j]; lo += (unsigned int)w[o][j]; hi += (unsigned int)(w[o][j] >> 32); }
        if (e < 9 * kTile3Cells) (&s_sum[0][0])[e] += tot;
        else if (e < kSlabWords) { s_n[2 * (e - 9 * kTile3Cells)] += lo; s_n[2 * (e - 9 * kTile3Cells) + 1] += hi; }
      }
    }
    __syncthreads();
    if (merge) {                                     // the cached sums of a submap update, once per tile
      for (int c = threadIdx.x; c < kTile3Cells; c += kBinThreads) {
        const int ix = tx0 + (c & ((1 << kT3x) - 1)), iy = ty0 + ((c >> kT3x) & ((1 << kT3y) - 1)), iz = tz0 + (c >> (kT3x + kT3y));
        if (ix < g.W && iy < g.H && iz < g.D) {
          const CellAcc3 a = g.acc[((size_t)iz * g.H + iy) * g.W + ix];
          s_n[c] += a.n;
#pragma unroll
          for (int j = 0; j < 3; ++j) s_sum[j][c] += (unsigned long long)a.s[j];
#pragma unroll
          for (int j = 0; j < 6; ++j) s_sum[3 + j][c] += (unsigned long long)a.ss[j];
        }
      }
      __syncthreads();
    }
  }
  NDT_STAMP3(3);
  int nvalid = 0;
  for (int c = threadIdx.x; c < kTile3Cells; c += kBinThreads) {
    const int ix = tx0 + (c & ((1 << kT3x) - 1)), iy = ty0 + ((c >> kT3x) & ((1 << kT3y) - 1)), iz = tz0 + (c >> (kT3x + kT3y));
    if (ix < g.W && iy < g.H && iz < g.D) {
      const size_t k = ((size_t)iz * g.H + iy) * g.W + ix;
      CellAcc3 a;
#pragma unroll
      for (int j = 0; j < 3; ++j) a.s[j] = (long long)s_sum[j][c];
#pragma unroll
      for (int j = 0; j < 6; ++j) a.ss[j] = (long long)s_sum[3 + j][c];
      a.n = s_n[c]; a.pad = 0u;
      float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
      if (a.n > kMaxCellCount) nover++;
      else if (finalise_sums3(a, cell_centre(g.ox, ix, g.cell), cell_centre(g.oy, iy, g.cell), cell_centre(g.oz, iz, g.cell),
                              g.fix_scale, min_points, eig_ratio, ra, rb, rc))
        nvalid++;
      g.acc[k] = a;
      g.rec[4 * k] = ra; g.rec[4 * k + 1] = rb; g.rec[4 * k + 2] = rc;
    }
  }
  NDT_STAMP3(4);
  block_count_add(counters, nvalid, nover);       // one add per workgroup, sharded (ndt_device.hpp)
  NDT_STAMP3(5);
}

// The end of a 3D build on the host's side (as k_build_publish) - and, behind the flag, the accumulator block is cleared for
// the NEXT build, which then starts without a fill launch (the block's words are dead once they are on the host).  One
// workgroup; wave 0 publishes.
__global__ __launch_bounds__(256) void k_build_publish_clear3(unsigned int* __restrict__ block, unsigned int* __restrict__ host_dst,
                                                               int nwords, int* __restrict__ host_flag, int seq, int clear_words) {
  if (threadIdx.x < 64) {
    for (int i = threadIdx.x; i < nwords; i += 64) __hip_atomic_store(host_dst + i, block[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();                                   // (every published word has been read)
  for (int i = threadIdx.x; i < clear_words; i += 256) block[i] = 0u;
}

}  // namespace ndt

#if defined(NDT_BUILD_PHASE_CLOCKS)
extern "C" int ndt_exp_read_tile3_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndt::g_tile3_stamps), sizeof(ndt::g_tile3_stamps));
}
#endif
