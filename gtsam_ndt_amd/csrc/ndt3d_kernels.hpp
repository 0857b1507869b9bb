// 3D NDT SE(3) variant (BASELINE config 5).  Same structure as the
// 2D path: exact fixed-point per-cell sums, one launch per Gauss-Newton iteration with the
// reduction of the previous launch's block partials and the 6x6 solve in every workgroup's
// prologue, kernel boundary as the only inter-workgroup synchronisation.
// Pose = (tx, ty, tz, roll, pitch, yaw), R = Rz(yaw) Ry(pitch) Rx(roll); 21 + 6 + 2 = 29 sums.
#pragma once
#include "ndt2d_kernels.hpp"

namespace ndt {

constexpr int kSum3RowStride = 66;
constexpr int kNumAcc3 = 40;   // rows of the partials table.  Gauss-Newton uses 29: Htt(6) Htr(9) Hrr(6) g(6) score nhit
                               // (its waves own rows 8w .. 8w+7); Newton 38: + M(9) = sum w v p' (rows 10w .. 10w+9)
template <int MODE> struct Acc3 {
  static constexpr int kUsed = MODE == 1 ? 38 : 29;     // sums per thread
  static constexpr int kRows = MODE == 1 ? 40 : 32;     // partial rows read back (a multiple of the 4 waves)
  static constexpr int kRowsPerWave = kRows / 4;
};

struct CellAcc3 {   // 80 B
  long long s[3];
  long long ss[6];   // xx xy xz yy yz zz
  unsigned int n;
  unsigned int pad;
};

struct Grid3Dev {
  float ox, oy, oz, inv_c;
  int W, H, D, pad;
  double cell, fix_scale;
  float4* rec;    // one 64-byte line per cell: rec[4k] = (mean_x, mean_y, mean_z, n; 0 = invalid),
                  // rec[4k+1] = Sigma^-1 (xx xy xz yy), rec[4k+2] = (yz zz 0 0), rec[4k+3] unused
  CellAcc3* acc;
};

struct IterState3 {
  double pose[6];
  double H[21];
  double g[6];
  double score;
  int n_hit, iter, status, done, have_partials, pad;
};   // 34 doubles + 6 ints = 296 B -> padded to 304 by alignment of the arrays below
static_assert(sizeof(IterState3) % 8 == 0, "IterState3 is copied as 8-byte words");

struct AlignStatic3 {
  Grid3Dev grid;
  SolveParams prm;
};
struct IterState3;
struct AlignCall3 {
  const float* sx; const float* sy; const float* sz;
  int n;
  int fixed_iterations;
  IterState3* host_state;    // as AlignCall: pinned host memory for the finishing launch, or null
  int* host_flag;
  int seq;                   // as AlignCall
  int pad;
};
struct LineSearch3 {      // LineSearch of ndt2d_kernels.hpp for a 6-vector pose
  double base[6];
  double step[6];
  double score;
  double alpha;
  int trials;
  int valid;
};
struct AlignDyn3 {
  IterState3 state[2];
  float partials[2][kNumAcc3][kMaxBlocks];
  LineSearch3 ls[2];
};

// A 3D scan moved into the map frame with the pose an alignment returned, before it is merged into the voxel
// grid (ndt3d_add_target_points_dev): R = Rz Ry Rx formed in float64 and rounded to float32 by the host,
// p' = ((r0 x + r1 y) + r2 z) + t per row with every float32 operation rounded separately (no contraction),
// so that a host restatement reproduces the points bit for bit.
struct Rigid3F { float r[9]; float t[3]; };
// the same arithmetic for kernels that move the points on the way in (the binned build of a submap update)
__device__ __forceinline__ void apply_rigid3(const Rigid3F& T, float& x, float& y, float& z) {
#pragma clang fp contract(off)
  const float px = x, py = y, pz = z;
  float o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float a = T.r[3 * r] * px, b = T.r[3 * r + 1] * py, c = T.r[3 * r + 2] * pz;
    o[r] = ((a + b) + c) + T.t[r];
  }
  x = o[0]; y = o[1]; z = o[2];
}
__global__ __launch_bounds__(kBlock) void k_transform_points3(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ z, size_t n, Rigid3F T,
                                                               float* __restrict__ ox, float* __restrict__ oy,
                                                               float* __restrict__ oz) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float px = x[i], py = y[i], pz = z[i];
  float* out[3] = {ox, oy, oz};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float a = T.r[3 * r] * px, b = T.r[3 * r + 1] * py, c = T.r[3 * r + 2] * pz;
    out[r][i] = ((a + b) + c) + T.t[r];
  }
}

// Range image of a spinning multi-beam lidar -> Cartesian SoA (the driver side of the 3D boundary): ring e (its
// elevation cos / sin from a table passed by value), azimuth column j at az0 + j * az_inc, one sincos per point from
// the column index; ranges outside [range_min, range_max] or non-finite become NaN points, which every kernel ignores.
constexpr int kMaxRings = 128;
struct RingTable { double cs[2 * kMaxRings]; };      // cos, sin of each ring's elevation: 2 KB of kernel arguments
__global__ __launch_bounds__(kBlock) void k_range_image_to_points(const float* __restrict__ r, int n_elev, int n_azim,
                                                                   RingTable rings, double az0, double az_inc,
                                                                   float range_min, float range_max,
                                                                   float* __restrict__ x, float* __restrict__ y,
                                                                   float* __restrict__ z) {
  const size_t n = (size_t)n_elev * n_azim;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int e = (int)(i / (size_t)n_azim), j = (int)(i - (size_t)e * n_azim);
    const float ri = r[i];
    double s, c;
    sincos(fma((double)j, az_inc, az0), &s, &c);
    const double ce = rings.cs[2 * e], se = rings.cs[2 * e + 1];
    const bool ok = isfinite(ri) & (ri >= range_min) & (ri <= range_max);
    x[i] = ok ? (float)((double)ri * (ce * c)) : NAN;
    y[i] = ok ? (float)((double)ri * (ce * s)) : NAN;
    z[i] = ok ? (float)((double)ri * se) : NAN;
  }
}

// ---------------------------------------------------------------------------- bounds
__global__ __launch_bounds__(kBlock) void k_bounds3(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ z, size_t n,
                                                     unsigned int* __restrict__ out /*[6]*/) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float p[3] = {x[i], y[i], z[i]};
    if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], p[a]); mx[a] = fmaxf(mx[a], p[a]); }
    }
  }
  __shared__ float s_b[kBlock / 64][6];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]);
    if ((threadIdx.x & 63) == 0) { s_b[wave][2 * a] = mn[a]; s_b[wave][2 * a + 1] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) {
      for (int w = 1; w < kBlock / 64; ++w) { mn[a] = fminf(mn[a], s_b[w][2 * a]); mx[a] = fmaxf(mx[a], s_b[w][2 * a + 1]); }
      atomicMin(&out[2 * a], float_to_ordered(mn[a]));
      atomicMax(&out[2 * a + 1], float_to_ordered(mx[a]));
    }
  }
}

// [r3] The same without same-line atomics and without a copy in either direction: one partial box per workgroup (plain
// stores), then one wave reduces the partials and writes the six ordered bounds straight into pinned host memory, a flag
// behind them (the host spins on it, as for the alignments).  k_bounds3's six atomics per workgroup all land on ONE
// 64-byte line and serialise at the memory side: 128 workgroups x 6 x ~10 ns were most of its 12 us.
// zero (may be null): the accumulator block of a build whose geometry is decided on the device is cleared here, one kernel
// before anything adds to it - words [0, zero_words) except [keep_from, keep_from + keep_words) (the geometry's own words).
__global__ __launch_bounds__(kBlock) void k_bounds3_parts(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, size_t n, float* __restrict__ parts /*[grid][8]*/,
                                                           unsigned int* __restrict__ zero, int zero_words, int keep_from, int keep_words) {
  if (zero && blockIdx.x == 0)
    for (int i = threadIdx.x; i < zero_words; i += kBlock)
      if (i < keep_from || i >= keep_from + keep_words) zero[i] = 0u;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += 4 * stride) {          // four points in flight
    float p[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t ii = i + u * stride;
      p[u][0] = ii < n ? x[ii] : NAN; p[u][1] = ii < n ? y[ii] : NAN; p[u][2] = ii < n ? z[ii] : NAN;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (isfinite(p[u][0]) && isfinite(p[u][1]) && isfinite(p[u][2])) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], p[u][a]); mx[a] = fmaxf(mx[a], p[u][a]); }
      }
  }
  __shared__ float s_b[kBlock / 64][6];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]);
    if ((threadIdx.x & 63) == 0) { s_b[wave][2 * a] = mn[a]; s_b[wave][2 * a + 1] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) {
      for (int w = 1; w < kBlock / 64; ++w) { mn[a] = fminf(mn[a], s_b[w][2 * a]); mx[a] = fmaxf(mx[a], s_b[w][2 * a + 1]); }
      parts[8 * blockIdx.x + 2 * a] = mn[a];
      parts[8 * blockIdx.x + 2 * a + 1] = mx[a];
    }
  }
}

__global__ __launch_bounds__(64) void k_bounds3_publish(const float* __restrict__ parts, int nparts, unsigned int* __restrict__ host_out /*[6]*/,
                                                         int* __restrict__ host_flag, int seq) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < nparts; i += 64) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], parts[8 * i + 2 * a]); mx[a] = fmaxf(mx[a], parts[8 * i + 2 * a + 1]); }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { mn[a] = wave_min(mn[a]); mx[a] = wave_max(mx[a]); }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool none = !(mn[a] <= mx[a]);                 // no finite point: k_bounds3's "empty" encoding
      __hip_atomic_store(host_out + 2 * a, none ? 0xFFFFFFFFu : float_to_ordered(mn[a]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_out + 2 * a + 1, none ? 0u : float_to_ordered(mx[a]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------- accumulate
__global__ __launch_bounds__(kBlock) void k_accumulate3(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, size_t n, Grid3Dev g,
                                                         unsigned long long* __restrict__ n_outside) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float px = x[i], py = y[i], pz = z[i];
    const float fx = (px - g.ox) * g.inv_c, fy = (py - g.oy) * g.inv_c, fz = (pz - g.oz) * g.inv_c;
    const bool in = (fx >= 0.f) & (fx < (float)g.W) & (fy >= 0.f) & (fy < (float)g.H) & (fz >= 0.f) & (fz < (float)g.D);
    if (!in && n_outside) atomicAdd(n_outside, 1ull);
    if (in) {
      const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
      const int ux = fix_coord(px, cell_centre(g.ox, ix, g.cell), g.fix_scale);
      const int uy = fix_coord(py, cell_centre(g.oy, iy, g.cell), g.fix_scale);
      const int uz = fix_coord(pz, cell_centre(g.oz, iz, g.cell), g.fix_scale);
      CellAcc3* c = g.acc + (((size_t)iz * g.H + iy) * g.W + ix);
      atomicAdd(&c->n, 1u);
      atomicAdd((unsigned long long*)&c->s[0], (unsigned long long)(long long)ux);
      atomicAdd((unsigned long long*)&c->s[1], (unsigned long long)(long long)uy);
      atomicAdd((unsigned long long*)&c->s[2], (unsigned long long)(long long)uz);
      atomicAdd((unsigned long long*)&c->ss[0], prod64(ux, ux));
      atomicAdd((unsigned long long*)&c->ss[1], prod64(ux, uy));
      atomicAdd((unsigned long long*)&c->ss[2], prod64(ux, uz));
      atomicAdd((unsigned long long*)&c->ss[3], prod64(uy, uy));
      atomicAdd((unsigned long long*)&c->ss[4], prod64(uy, uz));
      atomicAdd((unsigned long long*)&c->ss[5], prod64(uz, uz));
    }
  }
}

// ---------------------------------------------------------------------------- finalise
// Cyclic Jacobi, 6 sweeps over (0,1),(0,2),(1,2): oracle/ndt3d.py jacobi_eig3().
__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq,
                                           double* vp, double* vq) {
  if (fabs(apq) > 1e-300) {
    const double tau = (aqq - app) / (2.0 * apq);
    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
    app = app - t * apq;
    aqq = aqq + t * apq;
    apq = 0.0;
    const double rp = arp, rq = arq;
    arp = c * rp - s * rq;
    arq = s * rp + c * rq;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double a = vp[i], b = vq[i];
      vp[i] = c * a - s * b;
      vq[i] = s * a + c * b;
    }
  }
}

// sums -> (mean, n | Sigma^-1) for one voxel; false when the voxel is not usable
__device__ __forceinline__ bool finalise_sums3(const CellAcc3& c, double cx, double cy, double cz, double fix_scale,
                                               int min_points, double eig_ratio, float4& ra, float4& rb, float4& rc) {
  const int n = (int)c.n;
  if (n < min_points || n < 2) return false;
  const double inv_s = 1.0 / fix_scale, dn = (double)n;
  const double mx = fma((double)c.s[0] / dn, inv_s, cx);
  const double my = fma((double)c.s[1] / dn, inv_s, cy);
  const double mz = fma((double)c.s[2] / dn, inv_s, cz);
  const double den = 1.0 / (dn * fix_scale * fix_scale * (dn - 1.0));
  double axx = to_double(sub128(mul_s64(n, c.ss[0]), mul_s64(c.s[0], c.s[0]))) * den;
  double axy = to_double(sub128(mul_s64(n, c.ss[1]), mul_s64(c.s[0], c.s[1]))) * den;
  double axz = to_double(sub128(mul_s64(n, c.ss[2]), mul_s64(c.s[0], c.s[2]))) * den;
  double ayy = to_double(sub128(mul_s64(n, c.ss[3]), mul_s64(c.s[1], c.s[1]))) * den;
  double ayz = to_double(sub128(mul_s64(n, c.ss[4]), mul_s64(c.s[1], c.s[2]))) * den;
  double azz = to_double(sub128(mul_s64(n, c.ss[5]), mul_s64(c.s[2], c.s[2]))) * den;
  double v0[3] = {1, 0, 0}, v1[3] = {0, 1, 0}, v2[3] = {0, 0, 1};   // eigenvector columns
  for (int sweep = 0; sweep < 6; ++sweep) {
    jacobi_rot(axx, ayy, axy, axz, ayz, v0, v1);   // (p,q) = (0,1), r = 2
    jacobi_rot(axx, azz, axz, axy, ayz, v0, v2);   // (0,2), r = 1
    jacobi_rot(ayy, azz, ayz, axy, axz, v1, v2);   // (1,2), r = 0
  }
  const double lmax = fmax(axx, fmax(ayy, azz));
  if (!(lmax > 0.0)) return false;
  const double lim = eig_ratio * lmax;
  const double i0 = 1.0 / fmax(axx, lim), i1 = 1.0 / fmax(ayy, lim), i2 = 1.0 / fmax(azz, lim);
  const double cxx = i0 * v0[0] * v0[0] + i1 * v1[0] * v1[0] + i2 * v2[0] * v2[0];
  const double cxy = i0 * v0[0] * v0[1] + i1 * v1[0] * v1[1] + i2 * v2[0] * v2[1];
  const double cxz = i0 * v0[0] * v0[2] + i1 * v1[0] * v1[2] + i2 * v2[0] * v2[2];
  const double cyy = i0 * v0[1] * v0[1] + i1 * v1[1] * v1[1] + i2 * v2[1] * v2[1];
  const double cyz = i0 * v0[1] * v0[2] + i1 * v1[1] * v1[2] + i2 * v2[1] * v2[2];
  const double czz = i0 * v0[2] * v0[2] + i1 * v1[2] * v1[2] + i2 * v2[2] * v2[2];
  ra = make_float4((float)mx, (float)my, (float)mz, (float)n);
  rb = make_float4((float)cxx, (float)cxy, (float)cxz, (float)cyy);
  rc = make_float4((float)cyz, (float)czz, 0.f, 0.f);
  return true;
}

__global__ __launch_bounds__(kBlock) void k_finalise3(Grid3Dev g, int min_points, double eig_ratio,
                                                       int* __restrict__ counters /*[kCountShards][2]*/) {
  counters = count_shard(counters);
  const size_t ncell = (size_t)g.W * g.H * g.D;
  const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= ncell) return;
  const CellAcc3 c = g.acc[k];
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
  bool ok = false;
  if (c.n > kMaxCellCount) {
    atomicAdd(&counters[1], 1);
  } else if ((int)c.n >= min_points) {
    const unsigned int k32 = (unsigned int)k, w32 = (unsigned int)g.W, h32 = (unsigned int)g.H;
    const int ix = (int)(k32 % w32), iy = (int)((k32 / w32) % h32), iz = (int)(k32 / (w32 * h32));
    ok = finalise_sums3(c, cell_centre(g.ox, ix, g.cell), cell_centre(g.oy, iy, g.cell), cell_centre(g.oz, iz, g.cell),
                        g.fix_scale, min_points, eig_ratio, ra, rb, rc);
  }
  const unsigned long long valid_mask = __ballot(ok);      // one atomic per wave
  if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(valid_mask | (1ull << 63)) && valid_mask)
    atomicAdd(&counters[0], (int)__popcll(valid_mask));
  g.rec[4 * k] = ra;
  g.rec[4 * k + 1] = rb;
  g.rec[4 * k + 2] = rc;
}

// ---------------------------------------------------------------------------- solve (6x6)
__device__ __forceinline__ bool solve6(const double* A /*6x6 row-major, symmetric*/, const double* g, double* x) {
#if defined(NDT_EXP_DIAG_SOLVE3)      // tools only: what the 6x6 factorisation costs per launch (wrong steps, right timing)
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = -g[i] / fmax(fabs(A[7 * i]), 1e-12);
  return true;
#endif
  double dg[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) dg[i] = fmax(fabs(A[7 * i]), 1e-12);
  double lam = 0.0;
  for (int attempt = 0; attempt < 12; ++attempt) {
    double L[6][6], D[6], rD[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double p = A[7 * j] + lam * dg[j];
#pragma unroll
      for (int k = 0; k < j; ++k) p -= L[j][k] * L[j][k] * D[k];
      ok = ok && (p > 1e-12 * dg[j]);
      D[j] = p;
      rD[j] = fast_rcp(ok ? p : 1.0);
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        double a = A[6 * i + j];
#pragma unroll
        for (int k = 0; k < j; ++k) a -= L[i][k] * L[j][k] * D[k];
        L[i][j] = a * rD[j];
      }
    }
    if (ok) {
      double z[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double a = -g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) a -= L[i][k] * z[k];
        z[i] = a;
      }
      bool fin = true;
#pragma unroll
      for (int i = 5; i >= 0; --i) {
        double a = z[i] * rD[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) a -= L[k][i] * x[k];
        x[i] = a;
        fin = fin && isfinite(a);
      }
      if (fin) return true;
    }
    lam = (lam == 0.0) ? 1e-6 : lam * 10.0;
  }
  return false;
}

__device__ __forceinline__ bool gn_update3(double* pose, const double* A, const double* g, int n_hit, int& iter,
                                           int& status, const SolveParams& p, int fixed_iterations, double score,
                                           const LineSearch3* ls_in, LineSearch3* ls_out, bool write) {
  if (p.line_search > 0 && ls_in->valid && ls_in->trials < p.line_search &&
      (n_hit < p.min_hits || score < ls_in->score - kLineSearchTol * fabs(ls_in->score))) {
    LineSearch3 ls = *ls_in;
    ls.alpha *= 0.5;
    ls.trials += 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) pose[i] = ls.base[i] + ls.alpha * ls.step[i];
#pragma unroll
    for (int i = 3; i < 6; ++i) pose[i] = wrap_angle(ls.base[i] + ls.alpha * ls.step[i]);
    if (write) *ls_out = ls;
    iter += 1;
    status = 0;
    if (fixed_iterations > 0) return iter >= fixed_iterations;
    if (iter >= p.max_iterations) { status = 1; return true; }
    return false;
  }
  if (n_hit < p.min_hits) { status = 3; return true; }
  double d[6];
  if (!solve6(A, g, d)) { status = 2; return true; }
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] *= p.step_scale;
  const double nt2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const double nr2 = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
  double alpha = 1.0;
  if (nt2 > p.step_max_trans * p.step_max_trans) alpha = p.step_max_trans / sqrt(nt2);
  if (nr2 * alpha * alpha > p.step_max_rot * p.step_max_rot) alpha = p.step_max_rot / sqrt(nr2);
  if (p.line_search > 0 && write) {
#pragma unroll
    for (int i = 0; i < 6; ++i) { ls_out->base[i] = pose[i]; ls_out->step[i] = d[i] * alpha; }
    ls_out->score = score; ls_out->alpha = 1.0; ls_out->trials = 0; ls_out->valid = 1;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) pose[i] += d[i] * alpha;
#pragma unroll
  for (int i = 3; i < 6; ++i) pose[i] = wrap_angle(pose[i] + d[i] * alpha);
  iter += 1;
  status = 0;
  if (fixed_iterations > 0) return iter >= fixed_iterations;
  if (nt2 * alpha * alpha < p.eps_trans * p.eps_trans && nr2 * alpha * alpha < p.eps_rot * p.eps_rot) return true;
  if (iter >= p.max_iterations) { status = 1; return true; }
  return false;
}

// Newton form of the rotation block (MODE 1): A[3..5][3..5] += sum_ab (d2R / dk dl)[a][b] M[a][b] with
// M[3a + b] = sum w v_a p_b (rows 29..37 of the sums), at the pose the sums were taken at.
__device__ __forceinline__ void newton_rot_block3(const double* pose, const double* M, double* A) {
  // The six second derivatives follow from R and its first derivatives:
  // a roll derivative maps columns (1, 2) -> (col 2, -col 1), a yaw derivative rows (0, 1) ->
  // (-row 1, row 0), and d2Ry = -Ry + e_y e_y' (oracle/ndt3d.py rot_second_derivs states the products).
  double sa, ca, sb, cb, sg, cg;
  sincos_wrapped(pose[3], &sa, &ca);
  sincos_wrapped(pose[4], &sb, &cb);
  sincos_wrapped(pose[5], &sg, &cg);
  const double R[9] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa,
                       sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa,
                       -sb, cb * sa, cb * ca};
  const double Rb[9] = {-cg * sb, cg * cb * sa, cg * cb * ca, -sg * sb, sg * cb * sa, sg * cb * ca, -cb, -sb * sa, -sb * ca};
  auto cols = [&](const double* X) {                       // roll derivative of X: (0, X[:,2], -X[:,1]) . M
    double t2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) t2 += X[3 * a + 2] * M[3 * a + 1] - X[3 * a + 1] * M[3 * a + 2];
    return t2;
  };
  auto rows = [&](const double* X) {                       // yaw derivative of X: (-X[1,:], X[0,:], 0) . M
    double t2 = 0.0;
#pragma unroll
    for (int b = 0; b < 3; ++b) t2 += -X[3 + b] * M[b] + X[b] * M[3 + b];
    return t2;
  };
  const double Ra[9] = {0.0, R[2], -R[1], 0.0, R[5], -R[4], 0.0, R[8], -R[7]};
  double h_aa = 0.0, h_gg = 0.0, h_bb = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) h_aa -= R[3 * a + 1] * M[3 * a + 1] + R[3 * a + 2] * M[3 * a + 2];
#pragma unroll
  for (int b = 0; b < 3; ++b) h_gg -= R[b] * M[b] + R[3 + b] * M[3 + b];
#pragma unroll
  for (int k = 0; k < 9; ++k) h_bb -= R[k] * M[k];
  {
    const double u[3] = {-sg, cg, 0.0}, w3[3] = {0.0, ca, -sa};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) h_bb += u[a] * w3[b] * M[3 * a + b];
  }
  const double h_ab = cols(Rb), h_ag = rows(Ra), h_bg = rows(Rb);
  A[21] += h_aa; A[22] += h_ab; A[23] += h_ag; A[28] += h_bb; A[29] += h_bg; A[35] += h_gg;
  A[27] = A[22]; A[33] = A[23]; A[34] = A[29];
}

// Rotation, its three first-derivative matrices and the translation of a pose, float64 -> float32.
struct Rot3F {
  float R[9], Ra[9], Rb[9], Rg[9];
  float tx, ty, tz;
};
__device__ __forceinline__ void make_rot3(const double* pose, Rot3F& T) {
  double sa, ca, sb, cb, sg, cg;
  sincos_wrapped(pose[3], &sa, &ca);
  sincos_wrapped(pose[4], &sb, &cb);
  sincos_wrapped(pose[5], &sg, &cg);
  // R = Rz Ry Rx
  const double r00 = cg * cb, r01 = cg * sb * sa - sg * ca, r02 = cg * sb * ca + sg * sa;
  const double r10 = sg * cb, r11 = sg * sb * sa + cg * ca, r12 = sg * sb * ca - cg * sa;
  const double r20 = -sb, r21 = cb * sa, r22 = cb * ca;
  T.R[0] = (float)r00; T.R[1] = (float)r01; T.R[2] = (float)r02; T.R[3] = (float)r10; T.R[4] = (float)r11; T.R[5] = (float)r12;
  T.R[6] = (float)r20; T.R[7] = (float)r21; T.R[8] = (float)r22;
  // d/droll: columns 1,2 rotate: dR[:,1] = R[:,2], dR[:,2] = -R[:,1], dR[:,0] = 0
  T.Ra[0] = 0.f; T.Ra[1] = (float)r02; T.Ra[2] = (float)(-r01);
  T.Ra[3] = 0.f; T.Ra[4] = (float)r12; T.Ra[5] = (float)(-r11);
  T.Ra[6] = 0.f; T.Ra[7] = (float)r22; T.Ra[8] = (float)(-r21);
  // d/dpitch = Rz dRy Rx
  T.Rb[0] = (float)(-cg * sb); T.Rb[1] = (float)(cg * cb * sa); T.Rb[2] = (float)(cg * cb * ca);
  T.Rb[3] = (float)(-sg * sb); T.Rb[4] = (float)(sg * cb * sa); T.Rb[5] = (float)(sg * cb * ca);
  T.Rb[6] = (float)(-cb);      T.Rb[7] = (float)(-sb * sa);     T.Rb[8] = (float)(-sb * ca);
  // d/dyaw: rows rotate: dR[0,:] = -R[1,:], dR[1,:] = R[0,:], dR[2,:] = 0
  T.Rg[0] = (float)(-r10); T.Rg[1] = (float)(-r11); T.Rg[2] = (float)(-r12);
  T.Rg[3] = (float)r00;    T.Rg[4] = (float)r01;    T.Rg[5] = (float)r02;
  T.Rg[6] = 0.f; T.Rg[7] = 0.f; T.Rg[8] = 0.f;
  T.tx = (float)pose[0]; T.ty = (float)pose[1]; T.tz = (float)pose[2];
}

// a5 + a6 for one point: p = the untransformed source point, pp = R p + t (both zeroed by the caller when the
// image lies outside the grid), the voxel record (A4 = mean | n, B4 = Sigma^-1 xx xy xz yy, C2 = yz zz).
// acc: Htt(6) Htr(9) Hrr(6) g(6) score n_hit [+ M(9) in Newton mode].
template <int MODE>
__device__ __forceinline__ void accumulate_point3(const Rot3F& T, float x, float y, float z, float px, float py,
                                                  float pz, bool in, const float4& A4, const float4& B4,
                                                  const float4& C2, float d1, float d2, float nhd2, float* acc) {
  const bool hit = in & (A4.w > 0.f);
  const float qx = px - A4.x, qy = py - A4.y, qz = pz - A4.z;
  const float cxx = B4.x, cxy = B4.y, cxz = B4.z, cyy = B4.w, cyz = C2.x, czz = C2.y;
  const float vx = fmaf(cxx, qx, fmaf(cxy, qy, cxz * qz));
  const float vy = fmaf(cxy, qx, fmaf(cyy, qy, cyz * qz));
  const float vz = fmaf(cxz, qx, fmaf(cyz, qy, czz * qz));
  const float m = fmaf(qx, vx, fmaf(qy, vy, qz * vz));
  const float s = hit ? d1 * __builtin_amdgcn_exp2f(nhd2 * m) : 0.f;
  const float w = s * d2;
  float J[3][3], U[3][3];   // J[k] = dR_k p ; U[k] = Sigma^-1 J[k]
  const float* Rd[3] = {T.Ra, T.Rb, T.Rg};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    J[k][0] = fmaf(Rd[k][0], x, fmaf(Rd[k][1], y, Rd[k][2] * z));
    J[k][1] = fmaf(Rd[k][3], x, fmaf(Rd[k][4], y, Rd[k][5] * z));
    J[k][2] = fmaf(Rd[k][6], x, fmaf(Rd[k][7], y, Rd[k][8] * z));
    U[k][0] = fmaf(cxx, J[k][0], fmaf(cxy, J[k][1], cxz * J[k][2]));
    U[k][1] = fmaf(cxy, J[k][0], fmaf(cyy, J[k][1], cyz * J[k][2]));
    U[k][2] = fmaf(cxz, J[k][0], fmaf(cyz, J[k][1], czz * J[k][2]));
  }
  const float v3[3] = {vx, vy, vz};
  float tk[3];                                               // v' J_k
#pragma unroll
  for (int k = 0; k < 3; ++k) tk[k] = fmaf(vx, J[k][0], fmaf(vy, J[k][1], vz * J[k][2]));
  const float wd = MODE == 1 ? -d2 * w : 0.f;                // Newton: - d2 w (J'v)(J'v)' on every entry
  {
    const float cc[6] = {cxx, cxy, cxz, cyy, cyz, czz};
    int q = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) {
        acc[q] = fmaf(w, cc[q], acc[q]);
        if (MODE == 1) acc[q] = fmaf(wd * v3[i], v3[j], acc[q]);
        ++q;
      }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      acc[6 + 3 * r + k] = fmaf(w, U[k][r], acc[6 + 3 * r + k]);
      if (MODE == 1) acc[6 + 3 * r + k] = fmaf(wd * v3[r], tk[k], acc[6 + 3 * r + k]);
    }
  {
    int q = 15;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = k; l < 3; ++l) {
        const float h = fmaf(J[k][0], U[l][0], fmaf(J[k][1], U[l][1], J[k][2] * U[l][2]));
        acc[q] = fmaf(w, h, acc[q]);
        if (MODE == 1) acc[q] = fmaf(wd * tk[k], tk[l], acc[q]);
        ++q;
      }
  }
  acc[21] = fmaf(w, vx, acc[21]); acc[22] = fmaf(w, vy, acc[22]); acc[23] = fmaf(w, vz, acc[23]);
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[24 + k] = fmaf(w, tk[k], acc[24 + k]);
  if (MODE == 1) {                                           // M[a][b] += w v_a p_b (x, y, z are zero for a miss)
    const float p3[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[29 + 3 * a + b] = fmaf(w * v3[a], p3[b], acc[29 + 3 * a + b]);
  }
  acc[27] += s;
  acc[28] += hit ? 1.f : 0.f;
}

// a4-a7 of one launch for one alignment, shared by k_iterate3 and the multi-scan chain (k_multi_body3): this
// workgroup's points (thread -> point assignment i, i + stride, ...; the first point already loaded) under `pose`,
// per-thread sums, the wave's sums through LDS, one partial row entry per sum at partial_col[row * kMaxBlocks].
template <int MODE>
__device__ __forceinline__ void evaluate_block3(const Grid3Dev& G, const SolveParams& prm, const double* pose,
                                                const float* __restrict__ sx, const float* __restrict__ sy,
                                                const float* __restrict__ sz, int n, int i, float x, float y, float z,
                                                float (*s_wave)[kNumAcc3], float* s_t_wave, float* __restrict__ partial_col) {
  constexpr int NA = Acc3<MODE>::kUsed;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stride = kMaxBlocks * kBlock;
  Rot3F T;
  make_rot3(pose, T);
  const float fW = (float)G.W, fH = (float)G.H, fD = (float)G.D;
  const float d1 = prm.d1, d2 = prm.d2;
  const float nhd2 = -0.5f * d2 * 1.44269504088896340736f;

  float acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = 0.f;

  while (i < n) {
    const int inext = i + stride;
    float xn = 0.f, yn = 0.f, zn = 0.f;
    if (inext < n) { xn = sx[inext]; yn = sy[inext]; zn = sz[inext]; }
    float px = fmaf(T.R[0], x, fmaf(T.R[1], y, fmaf(T.R[2], z, T.tx)));
    float py = fmaf(T.R[3], x, fmaf(T.R[4], y, fmaf(T.R[5], z, T.ty)));
    float pz = fmaf(T.R[6], x, fmaf(T.R[7], y, fmaf(T.R[8], z, T.tz)));
    const float fx = (px - G.ox) * G.inv_c, fy = (py - G.oy) * G.inv_c, fz = (pz - G.oz) * G.inv_c;
    const bool in = (fx >= 0.f) & (fx < fW) & (fy >= 0.f) & (fy < fH) & (fz >= 0.f) & (fz < fD);
    const int key = in ? (((int)fz * G.H + (int)fy) * G.W + (int)fx) : 0;
    if (!in) { px = py = pz = 0.f; x = y = z = 0.f; }
    const float4 A4 = G.rec[4 * key];
    const float4 B4 = G.rec[4 * key + 1];
    const float4 C2 = G.rec[4 * key + 2];
    accumulate_point3<MODE>(T, x, y, z, px, py, pz, in, A4, B4, C2, d1, d2, nhd2, acc);
    x = xn; y = yn; z = zn; i = inext;
  }

  // the wave's sums through LDS instead of one DPP tree each (wave_reduce11_lds of the 2D path):
  // park [j][lane] (row stride 66 floats), lane 2j+q adds the 32 values of accumulator j whose
  // lane index is q mod 2 in two chains, one quad DPP step folds the pair; 32 accumulators per round
  {
    float* t = s_t_wave;
#pragma unroll
    for (int j = 0; j < NA; ++j) t[j * kSum3RowStride + lane] = acc[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int base = 0; base < NA; base += 32) {
      const int j = base + (lane >> 1);
      float a = 0.f, b = 0.f;
      if (j < NA) {
        const float* row = t + j * kSum3RowStride + (lane & 1);
#pragma unroll
        for (int k = 0; k < 32; k += 2) { a += row[2 * k]; b += row[2 * k + 2]; }
      }
      float v = a + b;
      v += dpp_mov<0xB1, 0xf>(v);
      if ((lane & 1) == 0 && j < NA) s_wave[wave][j] = v;
    }
  }
  __syncthreads();
  if (tid < Acc3<MODE>::kRows) {
    const float r = tid < NA ? ((s_wave[0][tid] + s_wave[1][tid]) + s_wave[2][tid]) + s_wave[3][tid] : 0.f;
    partial_col[(size_t)tid * kMaxBlocks] = r;
  }
}

// Per-call part of the context (k_begin of the 2D path).
__global__ void k_begin3(AlignCall3* __restrict__ call, AlignDyn3* __restrict__ dyn, const float* sx,
                         const float* sy, const float* sz, int n, double p0, double p1, double p2, double p3,
                         double p4, double p5, int fixed_iterations, IterState3* host_state, int* host_flag, int seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  call->seq = seq;
  call->pad = 0;
  call->sx = sx; call->sy = sy; call->sz = sz;
  call->n = n;
  call->fixed_iterations = fixed_iterations;
  call->host_state = host_state;
  call->host_flag = host_flag;
  IterState3 s = {};
  s.pose[0] = p0; s.pose[1] = p1; s.pose[2] = p2;
  s.pose[3] = wrap_angle(p3); s.pose[4] = wrap_angle(p4); s.pose[5] = wrap_angle(p5);
  dyn->state[1] = s;
  dyn->state[0] = IterState3{};
  dyn->ls[0] = LineSearch3{};
  dyn->ls[1] = LineSearch3{};
}

__device__ __forceinline__ void copy_state3(IterState3* dst, const IterState3* src, int have_partials) {
  const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(src);
  unsigned long long* d8 = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
  for (int j = 0; j < (int)(sizeof(IterState3) / 8); ++j) d8[j] = s8[j];
  if (have_partials >= 0) dst->have_partials = have_partials;
}

// ---------------------------------------------------------------------------- iterate
// MODE 0: Gauss-Newton Hessian.  MODE 1: full Newton Hessian (Magnusson 2009, eq. 6.13): per point
// - d2 (J'v)(J'v)' on all 21 entries, and on the rotation block v' d2p'/dp_k dp_l, which is linear in the
// 3x3 matrix M = sum w v p' (p = the untransformed source point) - nine more sums per thread; the
// contraction with the six second-derivative matrices of R happens once per launch, in the prologue.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_iterate3(const AlignStatic3* __restrict__ st,
                                                      const AlignCall3* __restrict__ call,
                                                      AlignDyn3* __restrict__ dyn, int parity) {
  constexpr int NA = Acc3<MODE>::kUsed, RPW = Acc3<MODE>::kRowsPerWave;
  __shared__ double s_red[kNumAcc3];
  __shared__ float s_wave[kBlock / 64][kNumAcc3];
  __shared__ float s_t[kBlock / 64][NA * kSum3RowStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const IterState3* prev = &dyn->state[parity ^ 1];
  IterState3* cur = &dyn->state[parity];
  const bool writer = (blockIdx.x == 0) && (tid == 0);

  double pose[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) pose[j] = prev->pose[j];
  const int ps_iter = prev->iter, ps_done = prev->done, ps_have = prev->have_partials, ps_launch = prev->pad;
  const SolveParams prm = st->prm;
  const Grid3Dev G = st->grid;
  const int n = call->n;
  const int fixed_iterations = call->fixed_iterations;
  const float* __restrict__ sx = call->sx;
  const float* __restrict__ sy = call->sy;
  const float* __restrict__ sz = call->sz;
  IterState3* const host_state = call->host_state;
  int* const host_flag = call->host_flag;
  float4 pv[RPW];
  {
    const float* part = &dyn->partials[parity ^ 1][0][0];
#pragma unroll
    for (int v = 0; v < RPW; ++v)
      pv[v] = *reinterpret_cast<const float4*>(part + (wave * RPW + v) * kMaxBlocks + lane * 4);
  }
  asm volatile("" ::"s"(G.ox), "s"(G.oy), "s"(G.oz), "s"(G.inv_c), "s"(G.W), "s"(G.H), "s"(G.D), "s"(G.rec),
               "s"(prm.d1), "s"(prm.d2), "s"(prm.min_hits), "s"(prm.max_iterations),
               "s"(prm.eps_trans), "s"(prm.eps_rot), "s"(prm.step_max_trans), "s"(prm.step_max_rot), "s"(ps_iter),
               "s"(ps_done), "s"(ps_have), "s"(fixed_iterations));
  int i = blockIdx.x * kBlock + tid;
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < n) { x = sx[i]; y = sy[i]; z = sz[i]; }

  if (ps_done) {
    if (writer) {
      copy_state3(cur, prev, -1);
      if (host_flag) __hip_atomic_store(host_flag + 2, call->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // as k_iterate
    }
    return;
  }
  int iter = ps_iter;
  if (ps_have) {
    // this wave's rows of 256 block partials -> their totals.  Each lane folds its 4 blocks per row in
    // float64 and parks the values in LDS ([row][lane], stride 66); lane 8v+q then adds the 8
    // values of row v whose lane index is q mod 8 and three DPP steps fold the 8 lanes - instead of
    // float64 DPP trees (each 6 steps x 3 instructions) on the iteration's critical path.  Eight
    // rows per round (Gauss-Newton: one round; Newton: 10 rows, a second round of two).
    {
      double* t = reinterpret_cast<double*>(s_t[wave]);       // RPW x 66 doubles fit in the epilogue's buffer
#pragma unroll
      for (int v = 0; v < RPW; ++v)
        t[v * kSum3RowStride + lane] = (((double)pv[v].x + (double)pv[v].y) + (double)pv[v].z) + (double)pv[v].w;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int base = 0; base < RPW; base += 8) {
        const int v = base + (lane >> 3);
        double a = 0.0;
        if (v < RPW) {
          const double* row = t + v * kSum3RowStride + (lane & 7);
          a = ((row[0] + row[8]) + (row[16] + row[24])) + ((row[32] + row[40]) + (row[48] + row[56]));
        }
        a += dpp_mov<0xB1, 0xf>(a);
        a += dpp_mov<0x4E, 0xf>(a);
        a += dpp_mov<0x124, 0xf>(a);                             // row_ror:4 moves data up: lane 8v+4 gets lane 8v
        if ((lane & 7) == 4 && v < RPW) s_red[wave * RPW + v] = a;
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // 6x6 from the 21 packed sums: Htt(6) Htr(9) Hrr(6)
    double A[36], g[6];
    A[0] = s_red[0]; A[1] = s_red[1]; A[2] = s_red[2]; A[7] = s_red[3]; A[8] = s_red[4]; A[14] = s_red[5];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) A[6 * r + 3 + k] = s_red[6 + 3 * r + k];
    A[21] = s_red[15]; A[22] = s_red[16]; A[23] = s_red[17]; A[28] = s_red[18]; A[29] = s_red[19]; A[35] = s_red[20];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < r; ++c) A[6 * r + c] = A[6 * c + r];
#pragma unroll
    for (int j = 0; j < 6; ++j) g[j] = s_red[21 + j];
    const double score = s_red[27];
    const int n_hit = (int)(s_red[28] + 0.5);
    if (MODE == 1) newton_rot_block3(pose, &s_red[29], A);
    int status = 0;
    const bool done = gn_update3(pose, A, g, n_hit, iter, status, prm, fixed_iterations, score, &dyn->ls[parity ^ 1],
                                 &dyn->ls[parity], writer);
    if (writer) {
      auto store = [&](IterState3* o) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { o->pose[j] = pose[j]; o->g[j] = g[j]; }
#pragma unroll
        for (int j = 0; j < 21; ++j) o->H[j] = s_red[j];
        if (MODE == 1) {                 // the stored rotation block is the full one (s_red holds it without the second derivatives)
          o->H[15] = A[21]; o->H[16] = A[22]; o->H[17] = A[23]; o->H[18] = A[28]; o->H[19] = A[29]; o->H[20] = A[35];
        }
        o->score = score;
        o->n_hit = n_hit; o->iter = iter; o->status = status;
        o->done = done ? 1 : 0; o->have_partials = 1; o->pad = ps_launch + 1;   // index of this launch
      };
      store(cur);
      if (host_flag) {                   // tell the host directly
        if (done) {                      // state and this launch's number first, then the flag
          store(host_state);
          __hip_atomic_store(host_flag + 1, ps_launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          const_cast<AlignCall3*>(call)->n = 0;    // the launches enqueued past the end load no points
          __threadfence_system();
          __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {                         // progress: which launch this is
          __hip_atomic_store(host_flag + 1, ps_launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (done) return;
  } else if (writer) {
    copy_state3(cur, prev, 1);
  }

  evaluate_block3<MODE>(G, prm, pose, sx, sy, sz, n, i, x, y, z, s_wave, s_t[wave], &dyn->partials[parity][0][blockIdx.x]);
}

}  // namespace ndt
