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

__device__ __forceinline__ int tile_of(const BinGeom& g, float px, float py) {
  const float fx = (px - g.ox) * g.inv_c, fy = (py - g.oy) * g.inv_c;
  const bool in = in_interior(fx, fy, g.W, g.H);      // ring cells stay empty (ndt2d_kernels.hpp)
  return in ? (((int)fy >> kTileShift) * g.ntx + ((int)fx >> kTileShift)) : -1;
}

// P1: per-tile totals.  LDS histogram per workgroup, one global atomic per touched tile.
__global__ __launch_bounds__(kBinThreads) void k_tile_count(const float* __restrict__ x, const float* __restrict__ y,
                                                             size_t n, BinGeom g, unsigned int* __restrict__ tile_total,
                                                             unsigned long long* __restrict__ n_outside) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_hist[];
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
                                                     unsigned int* __restrict__ tile_cursor, int ntile) {
  __shared__ unsigned int s_wave[16];
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
                                                               float* __restrict__ bx, float* __restrict__ by) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_mem[];
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
                                                                  double eig_ratio, int* __restrict__ counters) {
  __shared__ unsigned int s_n[kTileCells];
  __shared__ unsigned long long s_sum[5][kTileCells];
  const int tile = blockIdx.x;
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
  // one counter atomic per wave
  const int lane = threadIdx.x & 63;
  nvalid = (int)wave_sum((float)nvalid);
  nover = (int)wave_sum((float)nover);
  if (lane == 0) {
    if (nvalid) atomicAdd(&counters[0], nvalid);
    if (nover) atomicAdd(&counters[1], nover);
  }
}

}  // namespace ndt
