// Chunk-sorted target-grid build (rows a1-a3 of SURVEY.md section 8a): the default path of ndt2d_set_target and
// ndt2d_add_target_points* since round 3.
//
// The binned build of round 1 (ndt2d_build.hpp: count -> scan -> scatter -> accumulate) pays two full passes and
// 80 000 global atomics to learn where a tile's points go before it moves a single one, then moves them with 4-byte
// scattered stores.  Here nothing global is counted:
//   k_bounds_parts    the bounding box, one partial per workgroup (plain stores; the reduce is in k_geometry or
//                     k_bounds_reduce - no same-address atomics at the tail);
//   k_chunk_sort      each workgroup sorts ITS chunk of the cloud by 32 x 32-cell tile in LDS (histogram, scan,
//                     placement) and writes the chunk back in that order with fully coalesced 8-byte stores, plus one
//                     table entry per (tile, chunk): where the tile's run starts in the chunk and how long it is.
//                     Optionally moves the points into the map frame on the way in (the submap update);
//   k_tile_gather     one 1024-thread workgroup per tile walks the tile's row of the table, reads its runs (short
//                     contiguous reads), sums into LDS (cell slots padded to a stride of 33 so that a wall along y does
//                     not put a whole wave on one bank), finalises and writes sums and records with coalesced stores.
//                     In merge mode (submap update) tiles that received no point return at once.  A tile with many
//                     points is SHARED by up to kGatherSplit workgroups (grid.y): each sums its share of the runs, all
//                     but the last to finish leave their non-empty cells as a short list in global memory, the last
//                     one (a ticket per tile) adds the lists to its own sums and finalises - exact integers, so any
//                     split gives the same bits.  One workgroup per tile left the kernel waiting for the fullest
//                     tiles (11 000 of a million points on one CU's LDS atomic unit) with a third of the CUs idle, and
//                     a submap update on the 16 CUs of the 16 tiles a scan touches.
// The sums are the same exact integers as on every other path, so the records are bit-identical
// (tests/test_gpu_ndt2d.py::test_binned_build_equals_atomic_build pins all three builds against each other).
// The table has tiles x chunks entries: the path is taken while that is <= 2^20 and a row fits the gather kernel's LDS
// (4096 chunks); beyond it the round-1 path stays.
#pragma once
#include "ndt2d_build.hpp"

namespace ndt {

constexpr int kSortThreads = 256;
constexpr int kGatherThreads = 1024;
constexpr int kSortMaxChunks = 4096;                 // a table row is staged in the gather kernel's LDS
constexpr size_t kSortMaxTable = size_t(1) << 20;    // tiles x chunks
constexpr int kBoundsParts = 256;                    // workgroups of k_bounds_parts (one partial each)
constexpr int kTileStride = kTile + 1;               // LDS cell slots per tile row: 33, see above
constexpr int kTileSlots = kTileStride * kTile;
constexpr int kGatherSplit = 8;                      // workgroups that may share a tile (grid.y of k_tile_gather)

// what the workgroups that share a tile hand to the one that finishes it (all in one allocation of the handle)
struct SplitBufs {
  unsigned int* ticket;      // [tiles]: arrivals per tile; [tiles]: the pool's cursor.  Cleared by k_chunk_sort
  unsigned int* touched;     // [tiles]: the call number `seq` where this call's cloud put a point into the tile (k_chunk_sort;
                             // never cleared: an older number means "not this time") - lets the surplus workgroups of tiles
                             // that received nothing leave after one scalar load instead of after the tile's table row
  unsigned int seq;
  uint2* part;               // [tiles][kGatherSplit]: (first entry, entries) of a workgroup's list in the pool
  CellAcc* pool;             // the lists: a cell's sums with its LDS slot in `pad`; one entry per (workgroup, non-empty
                             // cell), so never more entries than points
  int tiles;                 // launch bound on the number of tiles (the cursor sits behind the tickets)
  int split_points;          // a tile is shared by ceil(points / split_points) workgroups, at most kGatherSplit
};

// ---- bounds: one partial per workgroup -------------------------------------------------------------------------
// parts[b] = (xmin, xmax, ymin, ymax) over the finite points workgroup b saw (+inf / -inf when it saw none).
// VEC: x and y are 16-byte aligned - four points per load.
// zero (may be null): the accumulators of a build whose geometry is decided on the device (counter shards, outside count)
// are cleared here, one kernel before anything adds to them.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void k_bounds_parts(const float* __restrict__ x, const float* __restrict__ y, size_t n,
                                                          float4* __restrict__ parts, GeomDev* __restrict__ zero) {
  if (zero && blockIdx.x == 0 && threadIdx.x < kCountInts + 2) {
    if (threadIdx.x < kCountInts) zero->counters[threadIdx.x] = 0;
    else if (threadIdx.x == kCountInts) zero->n_outside = 0ull;
    else zero->ok = 0;
  }
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  auto take = [&](float a, float b) {
    if (isfinite(a) && isfinite(b)) {
      xmin = fminf(xmin, a); xmax = fmaxf(xmax, a);
      ymin = fminf(ymin, b); ymax = fmaxf(ymax, b);
    }
  };
  const size_t stride = (size_t)gridDim.x * kBlock;
  const size_t t0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (VEC) {
    const size_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    for (size_t i = t0; i < n4; i += 4 * stride) {          // 16 points in flight per thread
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t ii = i + u * stride;
        a[u] = ii < n4 ? x4[ii] : make_float4(NAN, NAN, NAN, NAN);
        b[u] = ii < n4 ? y4[ii] : make_float4(NAN, NAN, NAN, NAN);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { take(a[u].x, b[u].x); take(a[u].y, b[u].y); take(a[u].z, b[u].z); take(a[u].w, b[u].w); }
    }
    for (size_t i = (n4 << 2) + t0; i < n; i += stride) take(x[i], y[i]);
  } else {
    for (size_t i = t0; i < n; i += 8 * stride) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t ii = i + u * stride;
        a[u] = ii < n ? x[ii] : NAN;
        b[u] = ii < n ? y[ii] : NAN;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) take(a[u], b[u]);
    }
  }
  xmin = wave_min(xmin); xmax = wave_max(xmax);
  ymin = wave_min(ymin); ymax = wave_max(ymax);
  __shared__ float s_b[kBlock / 64][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_b[wave][0] = xmin; s_b[wave][1] = xmax; s_b[wave][2] = ymin; s_b[wave][3] = ymax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) {
      xmin = fminf(xmin, s_b[w][0]); xmax = fmaxf(xmax, s_b[w][1]);
      ymin = fminf(ymin, s_b[w][2]); ymax = fmaxf(ymax, s_b[w][3]);
    }
    parts[blockIdx.x] = make_float4(xmin, xmax, ymin, ymax);
  }
}

// the partials of k_bounds_parts -> the ordered-float bounding box (one wave)
__device__ __forceinline__ void reduce_bounds_parts(const float4* __restrict__ parts, int nparts, unsigned int out[4]) {
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  for (int i = threadIdx.x & 63; i < nparts; i += 64) {
    const float4 p = parts[i];
    xmin = fminf(xmin, p.x); xmax = fmaxf(xmax, p.y); ymin = fminf(ymin, p.z); ymax = fmaxf(ymax, p.w);
  }
  xmin = wave_min(xmin); xmax = wave_max(xmax); ymin = wave_min(ymin); ymax = wave_max(ymax);
  // no finite point: the "empty" encoding k_bounds leaves behind (min = 0xFFFFFFFF, max = 0)
  const bool none = !(xmin <= xmax);
  out[0] = none ? 0xFFFFFFFFu : float_to_ordered(xmin); out[1] = none ? 0u : float_to_ordered(xmax);
  out[2] = none ? 0xFFFFFFFFu : float_to_ordered(ymin); out[3] = none ? 0u : float_to_ordered(ymax);
}

__global__ __launch_bounds__(64) void k_bounds_reduce(const float4* __restrict__ parts, int nparts, unsigned int* __restrict__ out) {
  unsigned int b[4];
  reduce_bounds_parts(parts, nparts, b);
  if (threadIdx.x == 0) { out[0] = b[0]; out[1] = b[1]; out[2] = b[2]; out[3] = b[3]; }
}

// a1 on the device (oracle/ndt2d.py grid_geometry; the arithmetic of setup_geometry() on the host, which recomputes it
// from the same bounds afterwards and compares): the grid of a bounding box, if it fits the handle's storage and the
// launch bounds the host chose.  One lane.  grid (may be null) receives the device context's copy.
__device__ inline bool decide_geometry(const unsigned int b[4], double c, unsigned long long cell_capacity, int tile_bound,
                                       BinGeom* bin, GridDev* grid) {
  if (b[0] == 0xFFFFFFFFu || b[1] == 0u) return false;          // no finite point
  const float xmin = ordered_to_float(b[0]), xmax = ordered_to_float(b[1]);
  const float ymin = ordered_to_float(b[2]), ymax = ordered_to_float(b[3]);
  const float inv_c = (float)(1.0 / c);
  const double bx = floor((double)xmin / c), by = floor((double)ymin / c);
  const float ox = (float)((bx - 1.0) * c), oy = (float)((by - 1.0) * c);
  const float fx = (xmax - ox) * inv_c, fy = (ymax - oy) * inv_c;
  const double kx = floor((double)fx), ky = floor((double)fy);
  if (!(kx >= 0.0 && ky >= 0.0 && (kx + 2.0) * (ky + 2.0) <= (double)cell_capacity)) return false;
  const int W = (int)kx + 2, H = (int)ky + 2;
  const int ntx = (W + kTile - 1) >> kTileShift, nty = (H + kTile - 1) >> kTileShift;
  if ((long long)ntx * nty > (long long)tile_bound) return false;
  bin->ox = ox; bin->oy = oy; bin->inv_c = inv_c; bin->W = W; bin->H = H; bin->ntx = ntx; bin->ntile = ntx * nty;
  if (grid) {
    grid->ox = ox; grid->oy = oy; grid->inv_c = inv_c; grid->cell32 = (float)c;
    grid->W = W; grid->H = H; grid->ngrid = 1; grid->pad = 0;
    const double sh[4][2] = {{0.0, 0.0}, {0.5, 0.0}, {0.0, 0.5}, {0.5, 0.5}};
#pragma unroll
    for (int q = 0; q < kMaxGrids; ++q) {
      grid->gx[q] = (float)((bx - 1.0 - sh[q][0]) * c);
      grid->gy[q] = (float)((by - 1.0 - sh[q][1]) * c);
    }
    grid->cell = c;
    grid->fix_scale = 4194304.0 / c;
    static_assert(kFixShift == 22, "fix_scale literal");
  }
  return true;
}

// A build whose geometry is decided on the device: k_chunk_sort does it in its prologue - EVERY workgroup reduces the
// partial boxes (4 KB) and applies the rule (the same inputs, the same arithmetic: the same grid everywhere), workgroup 0
// also writes the result where the gather kernel, the alignments and the host read it.  A kernel of its own for this
// one wave cost 5 us of the build (launch, a dependent chain of loads and float64 divisions, a kernel boundary).
struct GeomArgs {
  const float4* parts;        // null: the geometry comes with the launch (BinGeom argument)
  int nparts;
  int tile_bound;
  double cell;
  unsigned long long cell_capacity;
  GridDev* grid;              // the device context's grid header
  GeomDev* out;               // bin, ok, bounds (the accumulators in it were cleared by k_bounds_parts)
};

#if defined(NDT_BUILD_PHASE_CLOCKS)
__device__ unsigned long long g_sort_stamps[4096][2];        // tools-only: start / end clock of every k_chunk_sort workgroup
#endif
// ---- chunk sort ---------------------------------------------------------------------------------------------------
struct MoveArgs {          // the rigid motion of k_transform_points, applied on the way in (use = 0: none)
  float cs, sn, tx, ty;
  int use;
};

// table[t * nchunks + chunk] = start | len << 16: the run of tile t inside this chunk's sorted copy.
// P points per thread, chunk = 256 * P points (<= 4096: start and len fit 16 bits).
template <int P>
__global__ __launch_bounds__(kSortThreads) void k_chunk_sort(const float* __restrict__ x, const float* __restrict__ y, size_t n,
                                                              BinGeom g, int nchunks, MoveArgs mv, float2* __restrict__ binned,
                                                              unsigned int* __restrict__ table,
                                                              unsigned long long* __restrict__ n_outside, GeomArgs ga,
                                                              unsigned int* __restrict__ ticket, int n_ticket,
                                                              unsigned int* __restrict__ touched, unsigned int seq) {
#pragma clang fp contract(off)      // the motion must round as k_transform_points does (products, then sums)
  constexpr int C = kSortThreads * P;
  static_assert(C <= 4096, "start and length of a run are 16-bit fields");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  // the chunk's points are requested first: they do not depend on the geometry, and the loads fly while the prologue
  // below reduces the partial boxes and decides the grid
  float px[P], py[P];
  const size_t base = (size_t)blockIdx.x * C;
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const size_t i = base + (size_t)u * kSortThreads + threadIdx.x;
    px[u] = i < n ? x[i] : NAN;
    py[u] = i < n ? y[i] : NAN;
  }
  if (ga.parts) {                                          // geometry decided here (see GeomArgs)
    __shared__ BinGeom s_bin;
    __shared__ int s_ok;
    if (threadIdx.x < 64) {
      unsigned int b[4];
      reduce_bounds_parts(ga.parts, ga.nparts, b);
      if (threadIdx.x == 0) {
        BinGeom bin{};
        const bool ok = decide_geometry(b, ga.cell, ga.cell_capacity, ga.tile_bound, &bin, blockIdx.x == 0 ? ga.grid : nullptr);
        s_bin = bin;
        s_ok = ok ? 1 : 0;
        if (blockIdx.x == 0) {
          ga.out->bin = bin;
          ga.out->bounds[0] = b[0]; ga.out->bounds[1] = b[1]; ga.out->bounds[2] = b[2]; ga.out->bounds[3] = b[3];
          ga.out->ok = ok ? 1 : 0;
        }
      }
    }
    __syncthreads();
    if (!s_ok) return;                                     // uniform: the host repeats the build the slow way
    g = s_bin;
  }
  float2* s_pts = reinterpret_cast<float2*>(s_raw);                       // [C]
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(s_raw + (size_t)C * sizeof(float2));   // [ntile]
  __shared__ unsigned int s_wave[kSortThreads / 64];
  __shared__ unsigned int s_total;
  const int chunk = blockIdx.x;
  if (chunk == 0)                     // the tickets and the pool cursor of the gather kernel that follows
    for (int t = threadIdx.x; t < n_ticket; t += kSortThreads) ticket[t] = 0u;
#if defined(NDT_BUILD_PHASE_CLOCKS)
  if (threadIdx.x == 0) g_sort_stamps[blockIdx.x & 4095][0] = __builtin_amdgcn_s_memrealtime();
#endif
  for (int t = threadIdx.x; t < g.ntile; t += kSortThreads) s_hist[t] = 0u;
  __syncthreads();
  int tile[P];
  unsigned int rank[P];
  unsigned int outside = 0;
#pragma unroll
  for (int u = 0; u < P; ++u) {
    if (mv.use) {
      const float a = mv.cs * px[u], b = mv.sn * py[u], c = mv.sn * px[u], d = mv.cs * py[u];
      px[u] = (a - b) + mv.tx;
      py[u] = (c + d) + mv.ty;
    }
    tile[u] = tile_of(g, px[u], py[u]);
    rank[u] = tile[u] >= 0 ? atomicAdd(&s_hist[tile[u]], 1u) : 0u;
    outside += (tile[u] < 0) & (base + (size_t)u * kSortThreads + threadIdx.x < n);
  }
  __syncthreads();
  // exclusive scan of the histogram; the table entries of this chunk on the way
  const int per = (g.ntile + kSortThreads - 1) / kSortThreads;
  const int t0 = threadIdx.x * per;
  unsigned int local = 0;
  for (int k = 0; k < per; ++k) if (t0 + k < g.ntile) local += s_hist[t0 + k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int inc = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int v = __shfl_up(inc, d, 64);
    if (lane >= d) inc += v;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  unsigned int run = inc - local;
  for (int w = 0; w < wave; ++w) run += s_wave[w];
  for (int k = 0; k < per; ++k) {
    const int t = t0 + k;
    if (t < g.ntile) {
      const unsigned int len = s_hist[t];
      s_hist[t] = run;
      table[(size_t)t * nchunks + chunk] = run | (len << 16);
      if (len) touched[t] = seq;            // (every chunk that has points in the tile stores the same number)
      run += len;
    }
  }
  if (threadIdx.x == kSortThreads - 1) s_total = run;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < P; ++u)
    if (tile[u] >= 0) s_pts[s_hist[tile[u]] + rank[u]] = make_float2(px[u], py[u]);
  __syncthreads();
  const unsigned int total = s_total;
  float2* out = binned + base;
  for (unsigned int j = threadIdx.x; j < total; j += kSortThreads) out[j] = s_pts[j];
  if (n_outside) {
    outside = (unsigned int)wave_sum((float)outside);        // <= 64 * P: exact in float
    if (lane == 0 && outside) atomicAdd(n_outside, (unsigned long long)outside);
  }
#if defined(NDT_BUILD_PHASE_CLOCKS)
  if (threadIdx.x == 0) g_sort_stamps[blockIdx.x & 4095][1] = __builtin_amdgcn_s_memrealtime();
#endif
}

#if defined(NDT_BUILD_PHASE_CLOCKS)
__device__ unsigned long long g_gather_stamps[1024][8];      // tools-only: 100 MHz clock per phase, per workgroup
__device__ unsigned long long g_gather_wave_end[1024][16];   // ... and the end of every wave
#define NDT_STAMP(k) do { if (threadIdx.x == 0) g_gather_stamps[blockIdx.x & 1023][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NDT_STAMP(k) do { } while (0)
#endif

// ---- gather + accumulate + finalise: one workgroup per tile ------------------------------------------------------
// counters (sharded, ndt_device.hpp) += valid cells (merge: the CHANGE in valid cells of this tile), overflowed cells.
__global__ __launch_bounds__(kGatherThreads) void k_tile_gather(const float2* __restrict__ binned, const unsigned int* __restrict__ table,
                                                                 int nchunks, int chunk_points, GridDev g, int q, int ntx, int merge,
                                                                 int min_points, double eig_ratio, int* __restrict__ counters,
                                                                 const GeomDev* __restrict__ dg, const GridDev* __restrict__ dgrid,
                                                                 SplitBufs sb, unsigned int* __restrict__ zero_next) {
  // LDS: the five 64-bit sums and the counts of the tile's cells (SoA, slot = ly * 33 + lx), the tile's row of the
  // run table; the finished float32 records are staged over the sums at the end (s_rec) for whole-line stores.
  __shared__ __attribute__((aligned(16))) unsigned long long s_sum[5][kTileSlots];
  __shared__ unsigned int s_n[kTileSlots];
  __shared__ unsigned int s_runs[kSortMaxChunks];
  __shared__ unsigned int s_total, s_list, s_last;
  float4* const s_rec = reinterpret_cast<float4*>(&s_sum[0][0]);      // [1024 cells][2], 32 KB of the 42 KB
  static_assert(sizeof(float4) * 2 * kTileCells <= sizeof(unsigned long long) * 5 * kTileSlots, "the records overlay the sums");
  const int tile = blockIdx.x;
  // The grid's scalars, picked field by field: indexing g.gx[q] with a run-time q (or copying *dgrid over g) makes the
  // compiler keep the whole struct in scratch memory and reach the cell arrays through flat loads - seen in the ISA.
  float4* const rec = g.rec;
  CellAcc* const acc = g.acc;
  int W = g.W, H = g.H;
  float inv_c = g.inv_c;
  double cell_size = g.cell, fix_scale = g.fix_scale;
  float ox = q == 0 ? g.gx[0] : q == 1 ? g.gx[1] : q == 2 ? g.gx[2] : g.gx[3];
  float oy = q == 0 ? g.gy[0] : q == 1 ? g.gy[1] : q == 2 ? g.gy[2] : g.gy[3];
  if (dg) {       // geometry from the device (the launch covers the host's bound on the number of tiles); storage from g
    if (!dg->ok || tile >= dg->bin.ntile) return;
    ntx = dg->bin.ntx;
    W = dgrid->W; H = dgrid->H; inv_c = dgrid->inv_c; cell_size = dgrid->cell; fix_scale = dgrid->fix_scale;
    ox = dgrid->gx[0]; oy = dgrid->gy[0];        // (a device-decided geometry is a single grid)
  }
  NDT_STAMP(0);
  const int sub = blockIdx.y;
  // the NEXT build's accumulators (counter shards + outside count: the other half of a ping-pong pair, which nothing adds
  // to in this call) are cleared here instead of by two fill launches in front of every build
  if (zero_next && tile == 0 && sub == 0 && threadIdx.x < kCountInts + 2) zero_next[threadIdx.x] = 0u;
  if (__builtin_amdgcn_readfirstlane(sb.touched[tile]) != sb.seq && (sub > 0 || merge)) return;    // nothing arrived (uniform)
  if (threadIdx.x == 0) { s_total = 0u; s_list = 0u; }
  __syncthreads();
  {
    unsigned int pts = 0;
    const unsigned int* row = table + (size_t)tile * nchunks;
    for (int k = threadIdx.x; k < nchunks; k += kGatherThreads) { const unsigned int e = row[k]; s_runs[k] = e; pts += e >> 16; }
    pts = (unsigned int)wave_sum_u32(pts);
    if ((threadIdx.x & 63) == 0 && pts) atomicAdd(&s_total, pts);
  }
  __syncthreads();
  const unsigned int tile_points = s_total;                   // the same in every workgroup of the tile
  if (merge && tile_points == 0u) return;   // submap update: this tile received no point, its cells stand as they are (uniform)
  int nsub = (int)((tile_points + (unsigned)sb.split_points - 1u) / (unsigned)sb.split_points);
  nsub = nsub < 1 ? 1 : (nsub > (int)gridDim.y ? (int)gridDim.y : nsub);
  if (sub >= nsub) return;                                    // uniform
  const bool shared_tile = nsub > 1;
  const int tx0 = (tile % ntx) << kTileShift, ty0 = (tile / ntx) << kTileShift;
  const size_t gbase = (size_t)q * W * H;
  // one cell per thread: cell (lx, ly) of the tile lives in LDS slot ly * 33 + lx
  const int lx = threadIdx.x & (kTile - 1), ly = threadIdx.x >> kTileShift;
  const int slot = ly * kTileStride + lx;
  const int ix = tx0 + lx, iy = ty0 + ly;
  const bool in_grid = ix < W && iy < H;
  const size_t cell = gbase + (size_t)iy * W + ix;
  {
    CellAcc a = {0, 0, 0, 0, 0, 0u, 0u};
    if (merge && !shared_tile && in_grid) a = acc[cell];       // (a shared tile's cached sums are added by the workgroup that finishes it)
    s_n[slot] = a.n;
    s_sum[0][slot] = (unsigned long long)a.sx; s_sum[1][slot] = (unsigned long long)a.sy;
    s_sum[2][slot] = (unsigned long long)a.sxx; s_sum[3][slot] = (unsigned long long)a.sxy;
    s_sum[4][slot] = (unsigned long long)a.syy;
  }
  __syncthreads();
  NDT_STAMP(1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto add_point = [&](float2 p) {
    const float fx = (p.x - ox) * inv_c, fy = (p.y - oy) * inv_c;
    const int cx = (int)fx, cy = (int)fy;              // in range: the point was binned by the same formula
    const int ux = fix_coord(p.x, cell_centre(ox, cx, cell_size), fix_scale);
    const int uy = fix_coord(p.y, cell_centre(oy, cy, cell_size), fix_scale);
    const int c = (cy - ty0) * kTileStride + (cx - tx0);
    atomicAdd(&s_n[c], 1u);
    atomicAdd(&s_sum[0][c], (unsigned long long)(long long)ux);
    atomicAdd(&s_sum[1][c], (unsigned long long)(long long)uy);
    atomicAdd(&s_sum[2][c], prod64(ux, ux));
    atomicAdd(&s_sum[3][c], prod64(ux, uy));
    atomicAdd(&s_sum[4][c], prod64(uy, uy));
  };
  // A wave per run, four runs in flight: wave w takes the runs w, w + 16, ... of this tile.  The loads of four runs
  // are issued before any point is summed - one load per lane and run covers a run of up to 64 points, which is
  // nearly every run of a large cloud (a tile's share of a 4096-point chunk); a loop with one dependent load per
  // trip ran at the memory latency (10 us per tile, seen with in-kernel clocks).  Longer runs go on below, again with
  // four loads in flight.
  // (workgroup `sub` of the nsub that share the tile takes the runs sub, sub + nsub, ...; wave w every 16th of those)
  const int kstep = 16 * nsub;
  for (int k0 = sub + nsub * wave; k0 < nchunks; k0 += 4 * kstep) {
    const float2* r[4];
    unsigned int len[4];
    float2 p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + kstep * j;
      const unsigned int e = kk < nchunks ? s_runs[kk] : 0u;
      len[j] = e >> 16;
      r[j] = binned + (size_t)kk * chunk_points + (e & 0xFFFFu);
      p[j] = make_float2(0.f, 0.f);
      if ((unsigned)lane < len[j]) p[j] = r[j][lane];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((unsigned)lane < len[j]) add_point(p[j]);
    // the rest of a long run: every lane takes a CONTIGUOUS slice of it, so the lanes of one instruction are a slice
    // apart.  Long runs come from clouds in spatial order (a bearing-ordered scan puts whole chunks into one tile), where
    // neighbours in the array share a cell: lane = point mod 64 put a wave's 64 atomics on one or two LDS words.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (len[j] > 64u) {                                            // uniform
        const unsigned int slice = (len[j] - 64u + 63u) >> 6;
        const unsigned int first = 64u + (unsigned)lane * slice;
        constexpr int kF = 4;                                        // loads in flight per lane (8: the same time)
        for (unsigned int i0 = 0; i0 < slice; i0 += kF) {            // uniform trip count per run
          float2 t[kF];
#pragma unroll
          for (int u = 0; u < kF; ++u) {
            t[u] = make_float2(0.f, 0.f);
            if (i0 + u < slice && first + i0 + u < len[j]) t[u] = r[j][first + i0 + u];
          }
#pragma unroll
          for (int u = 0; u < kF; ++u)
            if (i0 + u < slice && first + i0 + u < len[j]) add_point(t[u]);
        }
      }
    }
  }
  NDT_STAMP(2);
  __syncthreads();
  if (shared_tile) {
    // This workgroup's non-empty cells as a list in the pool: the cell's sums, its LDS slot in `pad`.
    const unsigned int my_n = s_n[slot];
    unsigned int my_idx = 0;
    if (my_n) my_idx = atomicAdd(&s_list, 1u);
    __syncthreads();
    const unsigned int my_count = s_list;
    if (threadIdx.x == 0) s_total = my_count ? atomicAdd(&sb.ticket[sb.tiles], my_count) : 0u;       // the pool's cursor
    __syncthreads();
    const unsigned int my_base = s_total;
    // The hand-off avoids agent-scope fences: a release fence writes back every dirty line of the XCD's L2 (the sort's
    // output sits there: 7-9 us per workgroup, measured) - instead EVERY handed-off byte is stored write-through (sc1,
    // agent-scope atomic stores) and loaded with sc1 loads, every storing wave drains its stores, the workgroup meets,
    // and one lane takes the tile's ticket (MI355X_MICROARCH.md, "Valid forms" and the table under it: signal = one
    // lane's agent-scope atomic add for all the workgroup's stores, the consumer is the workgroup whose add came last).
    if (my_n) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(sb.pool + my_base + my_idx);
#pragma unroll
      for (int j = 0; j < 5; ++j) __hip_atomic_store(dst + j, s_sum[j][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 5, (unsigned long long)my_n | ((unsigned long long)(unsigned int)slot << 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);                                   // (n, pad = the cell's LDS slot)
    }
    if (threadIdx.x == 0)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(sb.part + (size_t)tile * kGatherSplit + sub),
                         (unsigned long long)my_base | ((unsigned long long)my_count << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&sb.ticket[tile], 1u) == (unsigned int)(nsub - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;                                       // uniform
    // the other workgroups' lists, one wave per list (read one after the other they cost two dependent memory round
    // trips each: 7 lists were 9 us)
    if (wave < nsub && wave != sub) {
      const int o = wave;
      const unsigned long long pw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(sb.part + (size_t)tile * kGatherSplit + o),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // written before o's ticket
      const unsigned int base = (unsigned int)(pw & 0xffffffffull), count = (unsigned int)(pw >> 32);
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(sb.pool + base);
      for (unsigned int i = lane; i < count; i += 64) {
        // agent-scope loads (sc1: served by L2, never by a stale line of this CU's L1)
        unsigned long long w[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) w[j] = __hip_atomic_load(src + 6 * (size_t)i + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int en = (unsigned int)(w[5] & 0xffffffffull), es = (unsigned int)(w[5] >> 32);
        atomicAdd(&s_n[es], en);                               // (other waves add other lists to the same cells)
#pragma unroll
        for (int j = 0; j < 5; ++j) atomicAdd(&s_sum[j][es], w[j]);
      }
    }
    __syncthreads();
    if (merge && in_grid) {                                    // the cached sums of a submap update, once per tile
      const CellAcc a = acc[cell];
      s_n[slot] += a.n;
      s_sum[0][slot] += (unsigned long long)a.sx; s_sum[1][slot] += (unsigned long long)a.sy;
      s_sum[2][slot] += (unsigned long long)a.sxx; s_sum[3][slot] += (unsigned long long)a.sxy;
      s_sum[4][slot] += (unsigned long long)a.syy;
    }
    __syncthreads();                                           // the write-back below reads other threads' slots
  }
  NDT_STAMP(3);
  int nvalid = 0, nover = 0;
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in_grid) {
    const unsigned int n = s_n[slot];
    const long long sx = (long long)s_sum[0][slot], sy = (long long)s_sum[1][slot], sxx = (long long)s_sum[2][slot],
                    sxy = (long long)s_sum[3][slot], syy = (long long)s_sum[4][slot];
    if (merge) nvalid -= rec[2 * cell + 1].z > 0.f ? 1 : 0;       // a valid record carries its point count there
    if (n > kMaxCellCount) nover++;
    else if ((int)n >= min_points &&
             finalise_sums((int)n, sx, sy, sxx, sxy, syy, cell_centre(ox, ix, cell_size), cell_centre(oy, iy, cell_size),
                           fix_scale, min_points, eig_ratio, ra, rb))
      nvalid++;
  }
  NDT_STAMP(4);
  // Write-back in whole lines.  A lane that stores its own cell's 48-byte sums and 32-byte record writes 16-byte
  // pieces 48 / 32 bytes apart: a third / a half of every line per instruction, and the 80 KB of a tile took 15 us to
  // drain (in-kernel clocks).  Here consecutive lanes store consecutive 16-byte pieces of a grid row's 32 cells:
  // the sums straight from the SoA arrays in LDS, the records after a pass through LDS.
  {
    const size_t row0 = gbase + (size_t)ty0 * W + tx0;           // the tile's first cell
    const int cols = W - tx0 < kTile ? W - tx0 : kTile;           // cells of a tile row that lie inside the grid
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pc = threadIdx.x + j * kGatherThreads;            // piece 0 .. 3071: (row, cell, third)
      const int row = pc / 96, r = pc - row * 96, cx = r / 3, third = r - cx * 3;
      if (ty0 + row < H && cx < cols) {
        const int sl = row * kTileStride + cx;
        ulonglong2 v;
        v.x = s_sum[2 * third][sl];
        v.y = third < 2 ? s_sum[2 * third + 1][sl] : (unsigned long long)s_n[sl];      // (n, pad = 0)
        *reinterpret_cast<ulonglong2*>(reinterpret_cast<char*>(acc + row0 + (size_t)row * W) + (size_t)r * 16) = v;
      }
    }
    __syncthreads();                       // every lane has read its sums: the records may overlay them
    s_rec[2 * threadIdx.x] = ra;           // thread t is cell (t & 31, t >> 5): records in cell order
    s_rec[2 * threadIdx.x + 1] = rb;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = threadIdx.x + j * kGatherThreads;            // piece 0 .. 2047: (row, cell, half)
      const int row = pc >> 6, r = pc & 63;
      if (ty0 + row < H && (r >> 1) < cols)
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(rec + 2 * (row0 + (size_t)row * W)) + (size_t)r * 16) = s_rec[pc];
    }
  }
  NDT_STAMP(5);
  block_count_add(counters, nvalid, nover);       // one add per workgroup, sharded (ndt_device.hpp)
#if defined(NDT_BUILD_PHASE_CLOCKS)
  if (lane == 0) g_gather_wave_end[blockIdx.x & 1023][wave] = __builtin_amdgcn_s_memrealtime();
#endif
}

// The end of a build on the host's side: the build's few result words (counter shards, outside count; the device-decided
// geometry) go straight into pinned host memory and a flag follows them - the host spins on the flag (as the alignments
// do) instead of paying a copy launch and a stream synchronisation's wake-up for 136 bytes.  One wave.
__global__ __launch_bounds__(64) void k_build_publish(const unsigned int* __restrict__ src, unsigned int* __restrict__ host_dst,
                                                       int nwords, int* __restrict__ host_flag, int seq) {
  for (int i = threadIdx.x; i < nwords; i += 64) __hip_atomic_store(host_dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace ndt

#if defined(NDT_BUILD_PHASE_CLOCKS)
extern "C" int ndt_exp_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndt::g_gather_stamps), sizeof(ndt::g_gather_stamps));
}
extern "C" int ndt_exp_read_wave_ends(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndt::g_gather_wave_end), sizeof(ndt::g_gather_wave_end));
}
extern "C" int ndt_exp_read_sort_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndt::g_sort_stamps), sizeof(ndt::g_sort_stamps));
}
#endif
