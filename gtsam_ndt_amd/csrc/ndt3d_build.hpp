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

__global__ __launch_bounds__(kBinThreads) void k_tile_count3(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ z, size_t n, BinGeom3 g,
                                                              unsigned int* __restrict__ tile_total,
                                                              unsigned long long* __restrict__ n_outside) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_hist[];
  for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) s_hist[t] = 0u;
  __syncthreads();
  unsigned int outside = 0;
  for (size_t i = (size_t)blockIdx.x * kBinThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kBinThreads) {
    const int t = tile_of3(g, x[i], y[i], z[i]);
    if (t >= 0) atomicAdd(&s_hist[t], 1u);
    else outside++;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < g.ntile; t += kBinThreads) {
    const unsigned int c = s_hist[t];
    if (c) atomicAdd(&tile_total[t], c);
  }
  if (n_outside && outside) atomicAdd(n_outside, (unsigned long long)outside);
}

__global__ __launch_bounds__(kBinThreads) void k_tile_scatter3(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ z, size_t n, BinGeom3 g,
                                                                unsigned int* __restrict__ tile_cursor,
                                                                float* __restrict__ bx, float* __restrict__ by,
                                                                float* __restrict__ bz) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_mem[];
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

// per tile: LDS sums, then the finalise of k_finalise3 (shared device function below)
__device__ __forceinline__ bool finalise_sums3(const CellAcc3& c, double cx, double cy, double cz, double fix_scale,
                                               int min_points, double eig_ratio, float4& ra, float4& rb, float4& rc);

__global__ __launch_bounds__(kBinThreads) void k_tile_accumulate3(const float* __restrict__ bx, const float* __restrict__ by,
                                                                   const float* __restrict__ bz,
                                                                   const unsigned int* __restrict__ tile_start, Grid3Dev g,
                                                                   int ntx, int nty, int merge, int min_points,
                                                                   double eig_ratio, int* __restrict__ counters,
                                                                   unsigned int* __restrict__ tile_ticket) {
  __shared__ unsigned int s_n[kTile3Cells];
  __shared__ unsigned long long s_sum[9][kTile3Cells];
  __shared__ int s_last;
  const int tile = blockIdx.x, sub = blockIdx.y;
  const int tx0 = (tile % ntx) << kT3x, ty0 = ((tile / ntx) % nty) << kT3y, tz0 = (tile / (ntx * nty)) << kT3z;
  // A lidar scan puts thousands of points into the few tiles around the sensor, and one workgroup per tile left the
  // build waiting for the fullest one.  A tile of more than kTile3SubMin points is shared by up to kTile3Split workgroups
  // (grid.y): each sums its share in LDS, adds it to the grid's sums with atomics (exact integers: any split gives
  // the same bits; the host zeroes the sums first unless this is a submap update), and the last one to arrive
  // finalises the tile.
  const unsigned int p0 = tile_start[tile], p1 = tile_start[tile + 1];
  int nsub = (int)((p1 - p0 + kTile3SubMin - 1) / kTile3SubMin);
  nsub = nsub < 1 ? 1 : (nsub > kTile3Split ? kTile3Split : nsub);
  if (sub >= nsub) return;                           // uniform
  const bool split = nsub > 1;
  // init: zeros, or the cached sums of this tile's voxels (merge = incremental submap update; a shared tile adds
  // to them in place instead)
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
  __syncthreads();
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
        atomicAdd(&s_n[c], 1u);
        atomicAdd(&s_sum[0][c], (unsigned long long)(long long)ux);
        atomicAdd(&s_sum[1][c], (unsigned long long)(long long)uy);
        atomicAdd(&s_sum[2][c], (unsigned long long)(long long)uz);
        atomicAdd(&s_sum[3][c], prod64(ux, ux));
        atomicAdd(&s_sum[4][c], prod64(ux, uy));
        atomicAdd(&s_sum[5][c], prod64(ux, uz));
        atomicAdd(&s_sum[6][c], prod64(uy, uy));
        atomicAdd(&s_sum[7][c], prod64(uy, uz));
        atomicAdd(&s_sum[8][c], prod64(uz, uz));
      }
    }
  }
  __syncthreads();
  if (split) {
    for (int c = threadIdx.x; c < kTile3Cells; c += kBinThreads) {
      const int ix = tx0 + (c & ((1 << kT3x) - 1)), iy = ty0 + ((c >> kT3x) & ((1 << kT3y) - 1)), iz = tz0 + (c >> (kT3x + kT3y));
      if (s_n[c] != 0u && ix < g.W && iy < g.H && iz < g.D) {
        CellAcc3* a = &g.acc[((size_t)iz * g.H + iy) * g.W + ix];
        atomicAdd(&a->n, s_n[c]);
#pragma unroll
        for (int j = 0; j < 3; ++j) atomicAdd(reinterpret_cast<unsigned long long*>(&a->s[j]), s_sum[j][c]);
#pragma unroll
        for (int j = 0; j < 6; ++j) atomicAdd(reinterpret_cast<unsigned long long*>(&a->ss[j]), s_sum[3 + j][c]);
      }
    }
    // [r3] No fences: everything handed over is written with agent-scope atomic adds (performed at the memory side, never
    // in this CU's L1 or this XCD's L2) and read back with agent-scope atomic loads, so all the hand-off needs is that
    // every wave's adds have completed before the workgroup's ticket is taken (MI355X_MICROARCH.md "Valid forms": agent
    // atomics on both sides).  A __threadfence() here is a write-back AND an invalidate of the whole L2, in all 256
    // threads, twice per shared workgroup: 5-7 us of the fullest tiles' critical path.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&tile_ticket[tile], 1u) == (unsigned)(nsub - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;                             // uniform
    // the last of the tile's workgroups: every share is in the grid's sums - read them back for the finalise below
    for (int c = threadIdx.x; c < kTile3Cells; c += kBinThreads) {
      const int ix = tx0 + (c & ((1 << kT3x) - 1)), iy = ty0 + ((c >> kT3x) & ((1 << kT3y) - 1)), iz = tz0 + (c >> (kT3x + kT3y));
      if (ix < g.W && iy < g.H && iz < g.D) {
        const CellAcc3* a = &g.acc[((size_t)iz * g.H + iy) * g.W + ix];
        s_n[c] = __hip_atomic_load(&a->n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          s_sum[j][c] = (unsigned long long)__hip_atomic_load(&a->s[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 6; ++j)
          s_sum[3 + j][c] = (unsigned long long)__hip_atomic_load(&a->ss[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
  int nvalid = 0, nover = 0;
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
      if (!split) g.acc[k] = a;
      g.rec[4 * k] = ra; g.rec[4 * k + 1] = rb; g.rec[4 * k + 2] = rc;
    }
  }
  block_count_add(counters, nvalid, nover);       // one add per workgroup, sharded (ndt_device.hpp)
}

}  // namespace ndt
