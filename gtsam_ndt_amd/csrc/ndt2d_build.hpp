// Binned target-grid build (row a2+a3 of SURVEY.md section 8a), the fast path of ndt2d_set_target.
//
// k_accumulate issues six scattered global atomics per point and runs at the chip's
// scattered-atomic rate (~18 G atomics/s: 328 us for 1M points).  Here the points are first
// bucketed by 32 x 32-cell tile (count -> scan -> scatter, all traffic coalesced or in short
// runs), then one workgroup per tile accumulates its points into LDS with LDS integer atomics,
// finalises the tile's cells and writes records and sums with coalesced stores.  The sums are
// the same exact integers, so the result is bit-identical to the atomic path (tests pin it).
#pragma once
#include "ndt2d_kernels.hpp"

namespace ndt {

constexpr int kTileShift = 5;                       // 32 x 32 cells per tile
constexpr int kTile = 1 << kTileShift;
constexpr int kTileCells = kTile * kTile;
constexpr int kBinMaxTiles = 8192;                  // LDS histogram capacity of the bucket kernels
constexpr int kBinThreads = 256;
constexpr int kBinPerThread = 8;                    // points per thread and chunk in count/scatter

struct BinGeom {
  float ox, oy, inv_c;
  int W, H, ntx, ntile;
};

// Geometry decided on the device (k_geometry below): the build kernels of one ndt2d_set_target call are then
// enqueued without waiting for the bounding box to reach the host.  ok = 0: the geometry does not fit the
// handle's storage or the launch bounds the host chose - every kernel that is given this struct returns at
// once and the host, which reads it back with the results, repeats the build the slow way.
struct GeomDev {
  BinGeom bin;
  int ok;
  unsigned int bounds[4];     // the ordered-float bounding box (k_bounds accumulates here), for the host
  int counters[kCountInts];   // valid cells, overflowed cells, in shards (ndt_device.hpp: block_count_add)
  unsigned long long n_outside;
};

// start of a single-sync build: everything the kernels accumulate into, in one launch instead of four memsets
__global__ __launch_bounds__(1024) void k_build_init(GeomDev* __restrict__ geom, unsigned int* __restrict__ tile_total, int tile_bound) {
  for (int t = threadIdx.x; t < tile_bound; t += 1024) tile_total[t] = 0u;
  if (threadIdx.x == 0) {
    geom->ok = 0;
    geom->bounds[0] = 0xFFFFFFFFu; geom->bounds[1] = 0u; geom->bounds[2] = 0xFFFFFFFFu; geom->bounds[3] = 0u;
    for (int k = 0; k < kCountInts; ++k) geom->counters[k] = 0;
    geom->n_outside = 0ull;
  }
}

// a1 on the device: oracle/ndt2d.py grid_geometry, the arithmetic of setup_geometry() on the host (the host
// recomputes it from the same bounds afterwards and compares).  One thread.
__global__ void k_geometry(double c, unsigned long long cell_capacity, int tile_bound, GridDev* __restrict__ grid,
                           GeomDev* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned int* bounds = out->bounds;
  GeomDev o = *out;
  if (!(bounds[0] == 0xFFFFFFFFu || bounds[1] == 0u)) {       // at least one finite point
    const float xmin = ordered_to_float(bounds[0]), xmax = ordered_to_float(bounds[1]);
    const float ymin = ordered_to_float(bounds[2]), ymax = ordered_to_float(bounds[3]);
    const float inv_c = (float)(1.0 / c);
    const double bx = floor((double)xmin / c), by = floor((double)ymin / c);
    const float ox = (float)((bx - 1.0) * c), oy = (float)((by - 1.0) * c);
    const float fx = (xmax - ox) * inv_c, fy = (ymax - oy) * inv_c;
    const double kx = floor((double)fx), ky = floor((double)fy);
    if (kx >= 0.0 && ky >= 0.0 && (kx + 2.0) * (ky + 2.0) <= (double)cell_capacity) {
      const int W = (int)kx + 2, H = (int)ky + 2;
      const int ntx = (W + kTile - 1) >> kTileShift, nty = (H + kTile - 1) >> kTileShift;
      if ((long long)ntx * nty <= (long long)tile_bound) {
        o.bin.ox = ox; o.bin.oy = oy; o.bin.inv_c = inv_c; o.bin.W = W; o.bin.H = H; o.bin.ntx = ntx; o.bin.ntile = ntx * nty;
        o.ok = 1;
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
    }
  }
  *out = o;
}

__device__ __forceinline__ int tile_of(const BinGeom& g, float px, float py) {
  const float fx = (px - g.ox) * g.inv_c, fy = (py - g.oy) * g.inv_c;
  const bool in = in_interior(fx, fy, g.W, g.H);      // ring cells stay empty (ndt2d_kernels.hpp)
  return in ? (((int)fy >> kTileShift) * g.ntx + ((int)fx >> kTileShift)) : -1;
}

// P1: per-tile totals.  LDS histogram per workgroup, one global atomic per touched tile.
__global__ __launch_bounds__(kBinThreads) void k_tile_count(const float* __restrict__ x, const float* __restrict__ y,
                                                             size_t n, BinGeom g, unsigned int* __restrict__ tile_total,
                                                             unsigned long long* __restrict__ n_outside,
                                                             const GeomDev* __restrict__ dg /* null: g is valid */) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_hist[];
  if (dg) { if (!dg->ok) return; g = dg->bin; }           // uniform
  for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) s_hist[t] = 0u;
  __syncthreads();
  unsigned int outside = 0;
  const size_t chunk = (size_t)kBinThreads * kBinPerThread;
  for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
    float px[kBinPerThread], py[kBinPerThread];
#pragma unroll
    for (int u = 0; u < kBinPerThread; ++u) {
      const size_t i = base + (size_t)u * kBinThreads + threadIdx.x;
      px[u] = i < n ? x[i] : NAN;
      py[u] = i < n ? y[i] : NAN;
    }
#pragma unroll
    for (int u = 0; u < kBinPerThread; ++u) {
      const size_t i = base + (size_t)u * kBinThreads + threadIdx.x;
      const int t = tile_of(g, px[u], py[u]);
      if (t >= 0) atomicAdd(&s_hist[t], 1u);
      else if (i < n) outside++;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) {
    const unsigned int c = s_hist[t];
    if (c) atomicAdd(&tile_total[t], c);
  }
  if (n_outside && outside) atomicAdd(n_outside, (unsigned long long)outside);
}

// exclusive scan of the tile totals (one workgroup; ntile <= kBinMaxTiles)
__global__ __launch_bounds__(1024) void k_tile_scan(const unsigned int* __restrict__ tile_total,
                                                     unsigned int* __restrict__ tile_start,
                                                     unsigned int* __restrict__ tile_cursor, int ntile,
                                                     const GeomDev* __restrict__ dg) {
  __shared__ unsigned int s_wave[16];
  if (dg) { if (!dg->ok) return; ntile = dg->bin.ntile; }
  const int per = (ntile + 1023) / 1024;
  const int t0 = threadIdx.x * per;
  unsigned int local = 0;
  for (int k = 0; k < per; ++k) if (t0 + k < ntile) local += tile_total[t0 + k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int inc = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int v = __shfl_up(inc, d, 64);
    if (lane >= d) inc += v;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  unsigned int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  unsigned int run = base + inc - local;
  for (int k = 0; k < per; ++k) {
    if (t0 + k < ntile) {
      tile_start[t0 + k] = run;
      tile_cursor[t0 + k] = run;
      run += tile_total[t0 + k];
    }
  }
  if (threadIdx.x == 1023) tile_start[ntile] = run;     // total number of binned points
}

// P2: scatter the points into tile order.  Rank within (workgroup, tile) from an LDS atomic,
// the workgroup's range in the tile from one global atomic per touched tile.
__global__ __launch_bounds__(kBinThreads) void k_tile_scatter(const float* __restrict__ x, const float* __restrict__ y,
                                                               size_t n, BinGeom g, unsigned int* __restrict__ tile_cursor,
                                                               float* __restrict__ bx, float* __restrict__ by,
                                                               const GeomDev* __restrict__ dg) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_mem[];
  if (dg) { if (!dg->ok) return; g = dg->bin; }
  unsigned int* s_hist = s_mem;
  unsigned int* s_base = s_mem + g.ntile;
  const size_t chunk = (size_t)kBinThreads * kBinPerThread;
  for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
    for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) s_hist[t] = 0u;
    __syncthreads();
    float px[kBinPerThread], py[kBinPerThread];
    int tile[kBinPerThread];
    unsigned int rank[kBinPerThread];
#pragma unroll
    for (int u = 0; u < kBinPerThread; ++u) {
      const size_t i = base + (size_t)u * kBinThreads + threadIdx.x;
      px[u] = i < n ? x[i] : NAN;
      py[u] = i < n ? y[i] : NAN;
    }
#pragma unroll
    for (int u = 0; u < kBinPerThread; ++u) {
      tile[u] = tile_of(g, px[u], py[u]);
      rank[u] = tile[u] >= 0 ? atomicAdd(&s_hist[tile[u]], 1u) : 0u;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) {
      const unsigned int c = s_hist[t];
      s_base[t] = c ? atomicAdd(&tile_cursor[t], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kBinPerThread; ++u) {
      if (tile[u] >= 0) {
        const unsigned int dst = s_base[tile[u]] + rank[u];
        bx[dst] = px[u];
        by[dst] = py[u];
      }
    }
    __syncthreads();
  }
}

// P3: one workgroup per tile.  LDS sums (+ the existing global sums when adding to a cached
// grid), finalise, coalesced write-back of sums and records.
__global__ __launch_bounds__(kBinThreads) void k_tile_accumulate(const float* __restrict__ bx, const float* __restrict__ by,
                                                                  const unsigned int* __restrict__ tile_start, GridDev g,
                                                                  int q, int ntx, int merge, int min_points,
                                                                  double eig_ratio, int* __restrict__ counters,
                                                                  const GeomDev* __restrict__ dg,
                                                                  const GridDev* __restrict__ dgrid) {
  __shared__ unsigned int s_n[kTileCells];
  __shared__ unsigned long long s_sum[5][kTileCells];
  const int tile = blockIdx.x;
  if (dg) {       // geometry from the device (the launch covers the host's bound on the number of tiles); storage from g
    if (!dg->ok || tile >= dg->bin.ntile) return;
    ntx = dg->bin.ntx;
    float4* rec = g.rec;
    CellAcc* acc = g.acc;
    g = *dgrid;
    g.rec = rec; g.acc = acc;
  }
  const int tx0 = (tile % ntx) << kTileShift, ty0 = (tile / ntx) << kTileShift;
  const size_t gbase = (size_t)q * g.W * g.H;
  const float ox = g.gx[q], oy = g.gy[q];
  // init: zeros, or the cached sums of this tile's cells
  for (int c = threadIdx.x; c < kTileCells; c += kBinThreads) {
    const int ix = tx0 + (c & (kTile - 1)), iy = ty0 + (c >> kTileShift);
    CellAcc a = {0, 0, 0, 0, 0, 0u, 0u};
    if (merge && ix < g.W && iy < g.H) a = g.acc[gbase + (size_t)iy * g.W + ix];
    s_n[c] = a.n;
    s_sum[0][c] = (unsigned long long)a.sx; s_sum[1][c] = (unsigned long long)a.sy;
    s_sum[2][c] = (unsigned long long)a.sxx; s_sum[3][c] = (unsigned long long)a.sxy;
    s_sum[4][c] = (unsigned long long)a.syy;
  }
  __syncthreads();
  const unsigned int p0 = tile_start[tile], p1 = tile_start[tile + 1];
  for (unsigned int i = p0 + threadIdx.x; i < p1; i += kBinThreads * 4) {
    float px[4], py[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned int ii = i + u * kBinThreads;
      px[u] = ii < p1 ? bx[ii] : 0.f;
      py[u] = ii < p1 ? by[ii] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u * kBinThreads < p1) {
        const float fx = (px[u] - ox) * g.inv_c, fy = (py[u] - oy) * g.inv_c;
        const int ix = (int)fx, iy = (int)fy;            // in range: the point was binned by the same formula
        const int ux = fix_coord(px[u], cell_centre(ox, ix, g.cell), g.fix_scale);
        const int uy = fix_coord(py[u], cell_centre(oy, iy, g.cell), g.fix_scale);
        const int c = ((iy - ty0) << kTileShift) + (ix - tx0);
        atomicAdd(&s_n[c], 1u);
        atomicAdd(&s_sum[0][c], (unsigned long long)(long long)ux);
        atomicAdd(&s_sum[1][c], (unsigned long long)(long long)uy);
        atomicAdd(&s_sum[2][c], prod64(ux, ux));
        atomicAdd(&s_sum[3][c], prod64(ux, uy));
        atomicAdd(&s_sum[4][c], prod64(uy, uy));
      }
    }
  }
  __syncthreads();
  int nvalid = 0, nover = 0;
  for (int c = threadIdx.x; c < kTileCells; c += kBinThreads) {
    const int ix = tx0 + (c & (kTile - 1)), iy = ty0 + (c >> kTileShift);
    if (ix < g.W && iy < g.H) {
      const size_t k = gbase + (size_t)iy * g.W + ix;
      const unsigned int n = s_n[c];
      CellAcc a;
      a.sx = (long long)s_sum[0][c]; a.sy = (long long)s_sum[1][c]; a.sxx = (long long)s_sum[2][c];
      a.sxy = (long long)s_sum[3][c]; a.syy = (long long)s_sum[4][c]; a.n = n; a.pad = 0u;
      float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n > kMaxCellCount) nover++;
      else if ((int)n >= min_points &&
               finalise_sums((int)n, a.sx, a.sy, a.sxx, a.sxy, a.syy, cell_centre(ox, ix, g.cell),
                             cell_centre(oy, iy, g.cell), g.fix_scale, min_points, eig_ratio, ra, rb))
        nvalid++;
      g.acc[k] = a;
      g.rec[2 * k] = ra;
      g.rec[2 * k + 1] = rb;
    }
  }
  block_count_add(counters, nvalid, nover);       // one add per workgroup, sharded (ndt_device.hpp)
}

}  // namespace ndt
