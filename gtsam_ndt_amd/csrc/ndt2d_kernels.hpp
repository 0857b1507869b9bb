// 2D NDT kernels for gfx950 (MI355X).  Hand-written HIP; wave64, DPP reductions, no MFMA
// (point-wise gather + reduction, HBM/latency bound - BASELINE.json north_star).
//
// Stage map (the reference has no source to cite,
// /root/reference/README.md:1 is its only line):
//   k_bounds        a1  bounding box of the target cloud
//   k_accumulate    a1+a2  cell key + exact fixed-point per-cell sums (n, Su, Suu')
//   k_finalise      a3  sums -> Welford form (n, mean, M2) -> Sigma -> clamp -> Sigma^-1 record
//   k_iterate<MODE> a4-a8  one launch per Gauss-Newton iteration:
//                       prologue  = fixed-order reduction of the previous launch's block
//                                   partials + 3x3 solve + pose update (every block, redundantly)
//                       body      = transform, lookup, score, Jacobian, per-thread sums
//                       epilogue  = wave DPP tree -> LDS -> one partial row per block
#pragma once
#include "ndt_device.hpp"

namespace ndt {

constexpr int kBlock = 256;        // threads per workgroup (4 waves)
constexpr int kMaxBlocks = 256;    // one workgroup per CU; also rows in the partials table
constexpr int kNumAcc = 12;        // Hxx Hxy Hyy Hxt Hyt Htt gx gy gt score nhit (pad)
constexpr int kFixShift = 22;      // fixed-point: U = rint(u * 2^22 / c), |U| <= 2^21
constexpr unsigned kMaxCellCount = 1u << 20;  // n*U^2 < 2^63 needs n <= 2^20 per cell

// ---- per-cell exact sums (48 B) -----------------------------------------------------
struct CellAcc {
  long long sx, sy, sxx, sxy, syy;
  unsigned int n;
  unsigned int pad;
};

constexpr int kMaxGrids = 4;       // Biber's four half-cell-shifted grids (params.overlap_grids = 4)

struct GridDev {
  float ox, oy, inv_c, cell32;   // origin of grid 0
  int W, H;
  int ngrid, pad;                // 1 or 4; grid g's cells are [g*W*H, (g+1)*W*H) of rec / acc
  float gx[kMaxGrids], gy[kMaxGrids];   // origin of grid g (gx[0] = ox, gy[0] = oy)
  double cell;       // params.cell_size
  double fix_scale;  // 2^kFixShift / cell
  float4* rec;       // one 32-byte record per cell = two float4: rec[2k] = (mean_x, mean_y, a, b),
                     // rec[2k+1] = (b, c, n as float, 0); both halves share one 64-byte line
  CellAcc* acc;
};

struct SolveParams {
  float d1, d2;
  int hessian_mode;
  int max_iterations;
  int min_hits;
  int line_search;   // > 0: backtracking line search, at most this many halvings per step
  double eps_trans, eps_rot, step_max_trans, step_max_rot;
  double step_scale;   // over-relaxation factor on the solved step (1 = plain)
};

// Backtracking line-search state (oracle/ndt2d.py gn_update): the pose the current step
// started from, its score, the clamped step and how far along it the trial pose sits.
struct LineSearch {
  double base[3];
  double step[3];
  double score;
  double alpha;
  int trials;
  int valid;
};
constexpr double kLineSearchTol = 1e-3;   // oracle/ndt2d.py LS_TOL

// State handed from launch k-1 to launch k through HBM (kernel boundary = the only
// inter-workgroup synchronisation; no in-launch hand-off, no atomics).
struct IterState {
  double pose[3];
  double H[6];   // xx xy yy xt yt tt of the last evaluation
  double g[3];
  double score;
  int n_hit;
  int iter;
  int status;
  int done;
  int have_partials;
  int pad;           // index of the launch that wrote this state (what the host sees as progress)
};

// Device context, split by who writes it so that the read-only parts can be fetched with
// scalar loads in one batch at kernel entry:
//   AlignStatic  grid + solver parameters; uploaded by the host when the target changes
//   AlignCall    per-alignment arguments; written by k_begin
//   AlignDyn     iteration state and block partials; written by k_iterate
struct AlignStatic {
  GridDev grid;
  SolveParams prm;
};
struct AlignCall {
  const float* sx;
  const float* sy;
  int n;
  int fixed_iterations;
  // converged mode: pinned host memory the finishing launch writes the final state and a flag
  // into, so the host sees the end of the loop without a copy or a stream sync (null = off)
  IterState* host_state;
  int* host_flag;
  int seq;                   // this call's number: what the first launch past the end reports in host_flag[2]
  int pad;
};
struct AlignDyn {
  IterState state[2];
  float partials[2][kNumAcc][kMaxBlocks];
  LineSearch ls[2];       // ping-pong like state; touched only when prm.line_search > 0
};

// A scan moved into the map frame with the pose an alignment returned, before it is merged into
// the submap (ndt2d_add_target_points_dev): float32 products and sums in a fixed order (no
// contraction), so that a host restatement reproduces the points bit for bit.
__global__ __launch_bounds__(kBlock) void k_transform_points(const float* __restrict__ x, const float* __restrict__ y,
                                                              size_t n, float cs, float sn, float tx, float ty,
                                                              float* __restrict__ ox, float* __restrict__ oy) {
#pragma clang fp contract(off)      // hipcc's __fmul_rn is a plain `*` and would be fused into an fma
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float px = x[i], py = y[i];
  const float a = cs * px, b = sn * py, c = sn * px, d = cs * py;
  ox[i] = (a - b) + tx;
  oy[i] = (c + d) + ty;
}

// ---------------------------------------------------------------------------- a1 bounds
__global__ __launch_bounds__(kBlock) void k_bounds(const float* __restrict__ x,
                                                    const float* __restrict__ y, size_t n,
                                                    unsigned int* __restrict__ out /*[4]*/) {
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  // 8 points in flight per thread: a one-point loop pays the memory latency per trip
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += 8 * stride) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t ii = i + u * stride;
      a[u] = ii < n ? x[ii] : NAN;
      b[u] = ii < n ? y[ii] : NAN;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (isfinite(a[u]) && isfinite(b[u])) {
        xmin = fminf(xmin, a[u]); xmax = fmaxf(xmax, a[u]);
        ymin = fminf(ymin, b[u]); ymax = fmaxf(ymax, b[u]);
      }
    }
  }
  xmin = wave_min(xmin); xmax = wave_max(xmax);
  ymin = wave_min(ymin); ymax = wave_max(ymax);
  __shared__ float s_b[kBlock / 64][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_b[wave][0] = xmin; s_b[wave][1] = xmax; s_b[wave][2] = ymin; s_b[wave][3] = ymax; }
  __syncthreads();
  if (threadIdx.x == 0) {   // one atomic per block and bound: the four words are contended
    for (int w = 1; w < kBlock / 64; ++w) {
      xmin = fminf(xmin, s_b[w][0]); xmax = fmaxf(xmax, s_b[w][1]);
      ymin = fminf(ymin, s_b[w][2]); ymax = fmaxf(ymax, s_b[w][3]);
    }
    atomicMin(&out[0], float_to_ordered(xmin));
    atomicMax(&out[1], float_to_ordered(xmax));
    atomicMin(&out[2], float_to_ordered(ymin));
    atomicMax(&out[3], float_to_ordered(ymax));
  }
}

// The grid's outermost ring of cells never receives a target point: the source-side lookup clamps
// keys onto the grid instead of testing them (image_key below), so every out-of-range, NaN or
// dead-lane lookup lands on a ring cell, which must therefore stay invalid.  ndt2d_set_target's
// geometry leaves the ring empty by construction (one guard cell either side) except for a
// boundary point that float32 rounding of (x - ox) * inv_c puts into cell 0 with a cell size that
// is not a power of two; ndt2d_reserve_target + ndt2d_add_target_points can put points anywhere.
// Both are handled here: a point whose cell lies on the ring counts as outside the grid
// (oracle/ndt2d.py cell_keys32(interior=True) states the same rule).
__device__ __forceinline__ bool in_interior(float fx, float fy, int W, int H) {
  return (fx >= 1.f) & (fx < (float)(W - 1)) & (fy >= 1.f) & (fy < (float)(H - 1));
}

// ---------------------------------------------------------------- shared cell arithmetic
// Used verbatim by the global-memory path (k_accumulate/k_finalise) and by the LDS-resident
// batch kernel, so both build bit-identical cell records from the same points.
__device__ __forceinline__ double cell_centre(float o, int i, double cell) {
  return fma((double)i + 0.5, cell, (double)o);
}
// |U| <= 2^21 for a point inside its cell, so the conversion is the single-instruction
// v_cvt_i32_f64 (an f64 -> i64 conversion is emulated) and products are 32 x 32 -> 64 bit.
__device__ __forceinline__ int fix_coord(float p, double centre, double fix_scale) {
  return __double2int_rn(((double)p - centre) * fix_scale);
}
__device__ __forceinline__ unsigned long long prod64(int a, int b) {
  return (unsigned long long)((long long)a * (long long)b);
}

// a3: exact sums -> Welford form (n, mean, M2) -> Sigma = M2/(n-1) -> eigenvalue clamp ->
// Sigma^-1 record.  float64 scalar code in the order of oracle/ndt2d.py finalise_cell().
// Returns false (record zeroed) when the cell is not usable.
__device__ __forceinline__ bool finalise_sums(int n, long long sx, long long sy, long long sxx, long long sxy,
                                              long long syy, double cx, double cy, double fix_scale,
                                              int min_points, double eig_ratio, float4& ra, float4& rb) {
  ra = make_float4(0.f, 0.f, 0.f, 0.f);
  rb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < min_points || n < 2) return false;
  // one reciprocal per cell: r = 1 / (n S); mean = centre + Su r; 1 / (n S^2) = r^2 n.  (A reciprocal of
  // fix_scale alone would be loop-invariant in every caller and cost a register pair held across the
  // whole build: the 1024-thread batch kernel spilled it.)
  const double dn = (double)n;
  const double r = 1.0 / (dn * fix_scale);
  const double mx = fma((double)sx, r, cx);
  const double my = fma((double)sy, r, cy);
  // M2 = (n*Suu - Su*Su) / (n * S^2), numerator exact in 128-bit
  const double den = r * r * dn;
  const double m2xx = to_double(sub128(mul_s64(n, sxx), mul_s64(sx, sx))) * den;
  const double m2xy = to_double(sub128(mul_s64(n, sxy), mul_s64(sx, sy))) * den;
  const double m2yy = to_double(sub128(mul_s64(n, syy), mul_s64(sy, sy))) * den;
  const double vxx = m2xx / (dn - 1.0), vxy = m2xy / (dn - 1.0), vyy = m2yy / (dn - 1.0);
  const double half_tr = 0.5 * (vxx + vyy);
  const double half_df = 0.5 * (vxx - vyy);
  const double disc = sqrt(fma(half_df, half_df, vxy * vxy));
  const double l1 = half_tr + disc;
  const double l2 = half_tr - disc;
  if (!(l1 > 0.0)) return false;
  const double l2c = fmax(l2, eig_ratio * l1);
  double ex, ey;
  if (half_df >= 0.0) { ex = half_df + disc; ey = vxy; }
  else                { ex = vxy; ey = disc - half_df; }
  const double nrm = sqrt(fma(ex, ex, ey * ey));
  if (nrm > 0.0) { ex /= nrm; ey /= nrm; } else { ex = 1.0; ey = 0.0; }
  const double i1 = 1.0 / l1, i2 = 1.0 / l2c, d = i1 - i2;
  const float b32 = (float)(d * ex * ey);
  ra = make_float4((float)mx, (float)my, (float)fma(d * ex, ex, i2), b32);
  rb = make_float4(b32, (float)fma(d * ey, ey, i2), (float)n, 0.f);
  return true;
}

// ------------------------------------------------------------------ a1+a2 accumulate
// Exact, order-independent per-cell sufficient statistics: cell-centred coordinates
// quantised to c*2^-22 and summed with 64-bit integer atomics.  Integer addition is
// associative, so the sums (and everything derived from them) are bitwise identical from
// run to run and under any point order; merging two clouds' sums is the exact form of the
// Chan/Welford pairwise update.
__global__ __launch_bounds__(kBlock) void k_accumulate(const float* __restrict__ x,
                                                        const float* __restrict__ y, size_t n,
                                                        GridDev g,
                                                        unsigned long long* __restrict__ n_outside) {
  unsigned int outside = 0;
  const size_t ncell = (size_t)g.W * g.H;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float px = x[i], py = y[i];
    bool any = false;
    for (int q = 0; q < g.ngrid; ++q) {
      const float ox = g.gx[q], oy = g.gy[q];
      const float fx = (px - ox) * g.inv_c;
      const float fy = (py - oy) * g.inv_c;
      const bool in = in_interior(fx, fy, g.W, g.H);
      if (in) {
        any = true;
        const int ix = (int)fx, iy = (int)fy;
        const int ux = fix_coord(px, cell_centre(ox, ix, g.cell), g.fix_scale);
        const int uy = fix_coord(py, cell_centre(oy, iy, g.cell), g.fix_scale);
        CellAcc* c = g.acc + (q * ncell + (size_t)iy * g.W + ix);
        atomicAdd(&c->n, 1u);
        atomicAdd((unsigned long long*)&c->sx, (unsigned long long)(long long)ux);
        atomicAdd((unsigned long long*)&c->sy, (unsigned long long)(long long)uy);
        atomicAdd((unsigned long long*)&c->sxx, prod64(ux, ux));
        atomicAdd((unsigned long long*)&c->sxy, prod64(ux, uy));
        atomicAdd((unsigned long long*)&c->syy, prod64(uy, uy));
      }
    }
    if (!any) outside++;
  }
  if (n_outside && outside) atomicAdd(n_outside, (unsigned long long)outside);
}

// ------------------------------------------------------------------------- a3 finalise
// One thread per cell.
__global__ __launch_bounds__(kBlock) void k_finalise(GridDev g, int min_points, double eig_ratio,
                                                      int* __restrict__ counters /*[kCountShards][2]: valid, overflow*/) {
  counters = count_shard(counters);
  const size_t ncell = (size_t)g.W * g.H;
  const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= ncell * g.ngrid) return;
  const CellAcc c = g.acc[k];
  float4 ra, rb;
  bool ok = false;
  if (c.n > kMaxCellCount) {
    atomicAdd(&counters[1], 1);
    ra = make_float4(0.f, 0.f, 0.f, 0.f);
    rb = make_float4(0.f, 0.f, 0.f, 0.f);
  } else if ((int)c.n >= min_points) {             // empty cells (the vast majority) skip everything
    const unsigned int w32 = (unsigned int)g.W, nc32 = (unsigned int)ncell;   // <= 2^27 cells: 32-bit div
    const unsigned int q = (unsigned int)k / nc32, k32 = (unsigned int)k - q * nc32;
    const int ix = (int)(k32 % w32), iy = (int)(k32 / w32);
    ok = finalise_sums((int)c.n, c.sx, c.sy, c.sxx, c.sxy, c.syy, cell_centre(g.gx[q], ix, g.cell),
                       cell_centre(g.gy[q], iy, g.cell), g.fix_scale, min_points, eig_ratio, ra, rb);
  } else {
    ra = make_float4(0.f, 0.f, 0.f, 0.f);
    rb = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // one atomic per wave, not one per valid cell (18.7k same-address atomics cost ~190 us)
  const unsigned long long valid_mask = __ballot(ok);
  if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(valid_mask | (1ull << 63)) && valid_mask)
    atomicAdd(&counters[0], (int)__popcll(valid_mask));
  g.rec[2 * k] = ra;
  g.rec[2 * k + 1] = rb;
}

// ---------------------------------------------------------------------- a8 solve/update
// Scalar float64 on the critical path of every iteration (each workgroup runs it
// redundantly in its prologue), so it is written for latency: LDL^T with three reciprocals
// (v_rcp_f64 + two Newton steps) instead of divisions and square roots, squared norms
// instead of sqrt, and a branch-free sincos for |t| <= pi.  Same algorithm and pivot tests as
// oracle/ndt2d.py solve3()/gn_update().
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ bool solve3(const double* H /*xx xy yy xt yt tt*/, const double* g,
                                       double* d) {
  const double h00 = H[0], h01 = H[1], h11 = H[2], h02 = H[3], h12 = H[4], h22 = H[5];
  const double d0 = fmax(fabs(h00), 1e-12), d1 = fmax(fabs(h11), 1e-12), d2 = fmax(fabs(h22), 1e-12);
  double lam = 0.0;
  for (int attempt = 0; attempt < 12; ++attempt) {
    const double a00 = h00 + lam * d0, a11 = h11 + lam * d1, a22 = h22 + lam * d2;
    if (a00 > 1e-12 * d0) {
      const double r0 = fast_rcp(a00);
      const double l10 = h01 * r0, l20 = h02 * r0;
      const double p1 = a11 - l10 * h01;
      if (p1 > 1e-12 * d1) {
        const double r1 = fast_rcp(p1);
        const double t = h12 - l20 * h01;
        const double l21 = t * r1;
        const double p2 = a22 - l20 * h02 - l21 * t;
        if (p2 > 1e-12 * d2) {
          const double r2 = fast_rcp(p2);
          const double z0 = -g[0];
          const double z1 = -g[1] - l10 * z0;
          const double z2 = -g[2] - l20 * z0 - l21 * z1;
          const double x2 = z2 * r2;
          const double x1 = z1 * r1 - l21 * x2;
          const double x0 = z0 * r0 - l10 * x1 - l20 * x2;
          if (isfinite(x0) && isfinite(x1) && isfinite(x2)) {
            d[0] = x0; d[1] = x1; d[2] = x2;
            return true;
          }
        }
      }
    }
    lam = (lam == 0.0) ? 1e-6 : lam * 10.0;
  }
  return false;
}

__device__ __forceinline__ double wrap_angle(double t) {
  const double pi = 3.141592653589793;
  if (t > pi || t <= -pi) {
    t = t - 2.0 * pi * floor((t + pi) / (2.0 * pi));
    if (t <= -pi) t += 2.0 * pi;
  }
  return t;
}

// sin/cos for |t| <= ~pi (poses are wrapped): quadrant reduction + the classic fdlibm
// minimax kernels on [-pi/4, pi/4]; error < 1 ulp, no table, no slow path.
__device__ __forceinline__ void sincos_wrapped(double t, double* sn, double* cs) {
  const double k = rint(t * 0.63661977236758134308);            // 2/pi
  double r = fma(-k, 1.57079632679489655800e+00, t);             // pi/2 hi
  r = fma(-k, 6.12323399573676603587e-17, r);                    // pi/2 lo
  const double z = r * r;
  const double ps = fma(fma(fma(fma(fma(1.58969099521155010221e-10, z, -2.50507602534068634195e-08), z,
                                    2.75573137070700676789e-06), z, -1.98412698298579493134e-04), z,
                            8.33333333332248946124e-03), z, -1.66666666666666324348e-01);
  const double s = fma(r * z, ps, r);
  const double pc = fma(fma(fma(fma(fma(-1.13596475577881948265e-11, z, 2.08757232129817482790e-09), z,
                                    -2.75573143513906633035e-07), z, 2.48015872894767294178e-05), z,
                            -1.38888888888741095749e-03), z, 4.16666666666666019037e-02);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = ((int)k) & 3;
  const double s1 = (q & 1) ? c : s;
  const double c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// returns done; updates pose/iter/status in place
// ls_in / ls_out: line-search state before / after this step (the same object, or the two
// ping-pong slots); touched only when p.line_search > 0; `write` selects the one storing thread.
__device__ __forceinline__ bool gn_update(double* pose, const double* H, const double* g, int n_hit,
                                          int& iter, int& status, const SolveParams& p, int fixed_iterations,
                                          double score, const LineSearch* ls_in, LineSearch* ls_out, bool write) {
  if (p.line_search > 0 && ls_in->valid && ls_in->trials < p.line_search &&
      (n_hit < p.min_hits || score < ls_in->score - kLineSearchTol * fabs(ls_in->score))) {
    // the evaluation was a trial of the previous step and it scored worse: halve and retry
    LineSearch ls = *ls_in;
    ls.alpha *= 0.5;
    ls.trials += 1;
    pose[0] = ls.base[0] + ls.alpha * ls.step[0];
    pose[1] = ls.base[1] + ls.alpha * ls.step[1];
    pose[2] = wrap_angle(ls.base[2] + ls.alpha * ls.step[2]);
    if (write) *ls_out = ls;
    iter += 1;
    status = 0;
    if (fixed_iterations > 0) return iter >= fixed_iterations;
    if (iter >= p.max_iterations) { status = 1; return true; }
    return false;
  }
  if (n_hit < p.min_hits) { status = 3; return true; }
  double d[3];
  if (!solve3(H, g, d)) { status = 2; return true; }
  d[0] *= p.step_scale; d[1] *= p.step_scale; d[2] *= p.step_scale;
  const double nt2 = d[0] * d[0] + d[1] * d[1];
  const double nr = fabs(d[2]);
  double alpha = 1.0;
  if (nt2 > p.step_max_trans * p.step_max_trans) alpha = p.step_max_trans / sqrt(nt2);
  if (nr * alpha > p.step_max_rot) alpha = p.step_max_rot / nr;
  if (p.line_search > 0 && write) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { ls_out->base[j] = pose[j]; ls_out->step[j] = d[j] * alpha; }
    ls_out->score = score; ls_out->alpha = 1.0; ls_out->trials = 0; ls_out->valid = 1;
  }
  pose[0] += d[0] * alpha;
  pose[1] += d[1] * alpha;
  pose[2] = wrap_angle(pose[2] + d[2] * alpha);
  iter += 1;
  status = 0;
  if (fixed_iterations > 0) return iter >= fixed_iterations;
  if (nt2 * alpha * alpha < p.eps_trans * p.eps_trans && nr * alpha < p.eps_rot) return true;
  if (iter >= p.max_iterations) { status = 1; return true; }
  return false;
}

// Per-call part of the context, written from kernel arguments (no host buffer lifetime).
__global__ void k_begin(AlignCall* __restrict__ call, AlignDyn* __restrict__ dyn, const float* sx,
                        const float* sy, int n, double p0, double p1, double p2, int fixed_iterations,
                        IterState* host_state, int* host_flag, int seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  call->seq = seq;
  call->pad = 0;
  call->sx = sx;
  call->sy = sy;
  call->n = n;
  call->fixed_iterations = fixed_iterations;
  call->host_state = host_state;
  call->host_flag = host_flag;
  IterState s = {};
  s.pose[0] = p0; s.pose[1] = p1; s.pose[2] = wrap_angle(p2);
  dyn->state[1] = s;            // launch 0 has parity 0 and reads slot 1
  dyn->state[0] = IterState{};
  dyn->ls[0] = LineSearch{};
  dyn->ls[1] = LineSearch{};
}

// ---- per-point pieces of the body (rows a4-a6) ------------------------------------------
// The loop-closure kernel is VALU-issue bound, so this code is written
// for instructions per point: clamped keys instead of an in-grid compare chain, one-instruction
// floor, 24-bit multiply, d1 folded into the exponent and d2 applied once to the finished sums,
// the SE(2) Jacobian read off the image point, hits counted on the scalar unit.  (Writing the
// (x,y) math on register pairs for v_pk_*_f32 was measured: no gain, gfx950 issues packed f32
// at half rate.)
typedef float v2f __attribute__((ext_vector_type(2)));

struct PoseF {
  float cs, sn, tx, ty, ox, oy, inv_c;
  int W, wm1, hm1;
  float lg_d1;   // log2(d1):  d1*exp(-d2/2 m) = exp2(nhd2*m + lg_d1)
  float nhd2;    // -d2/2 * log2(e)
  float d2;
};
__device__ __forceinline__ PoseF make_pose(float cs, float sn, float tx, float ty, float ox, float oy, float inv_c,
                                           int W, int H, float d1, float d2) {
  PoseF P;
  P.cs = cs; P.sn = sn; P.tx = tx; P.ty = ty; P.ox = ox; P.oy = oy; P.inv_c = inv_c;
  P.W = W; P.wm1 = W - 1; P.hm1 = H - 1;
  P.lg_d1 = __builtin_amdgcn_logf(d1);                 // v_log_f32 = log2
  P.nhd2 = -0.5f * d2 * 1.44269504088896340736f;
  P.d2 = d2;
  return P;
}

// floor(f) as int32 in ONE instruction (v_cvt_flr_i32_f32: saturating, NaN -> 0); the
// portable spelling costs v_floor_f32 + v_cvt_i32_f32 per coordinate in the hottest loop.
__device__ __forceinline__ int floor_to_int(float f) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}

// clamp(i, 0, hi) in one instruction (v_med3_i32); min(max()) with a runtime bound is two
__device__ __forceinline__ int clamp_index(int i, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "s"(hi));
  return r;
}

struct PointRec {
  float px, py;   // image R p + t of the source point (non-finite coordinates clamped to +-1e15)
  float4 A;       // mean_x, mean_y, a, b
  float4 B;       // b, c, n, 0
};

// a4 (1/2): transform and cell key.  No in-grid compare chain: the key is clamped onto the grid,
// whose outermost ring of cells is empty by construction (one guard cell below the minimum,
// one above the maximum), so every out-of-range, infinite or NaN point lands on an invalid
// cell.  NaN/inf never reach the sums: coordinates are clamped to +-1e15 first (v_med3_f32
// returns the finite bound for a NaN), which keeps every later product finite.
__device__ __forceinline__ void image_point(const PoseF& P, float x, float y, PointRec& r) {
  x = __builtin_amdgcn_fmed3f(x, -1e15f, 1e15f);
  y = __builtin_amdgcn_fmed3f(y, -1e15f, 1e15f);
  r.px = fmaf(P.cs, x, fmaf(-P.sn, y, P.tx));
  r.py = fmaf(P.sn, x, fmaf(P.cs, y, P.ty));
}
__device__ __forceinline__ int image_key(const PoseF& P, float ox, float oy, const PointRec& r, bool live) {
  int ix = floor_to_int((r.px - ox) * P.inv_c), iy = floor_to_int((r.py - oy) * P.inv_c);
  ix = clamp_index(ix, P.wm1);
  iy = clamp_index(iy, P.hm1);
  return live ? __mul24(iy, P.W) + ix : 0;                        // cell 0 is a guard cell; W, iy < 2^24
}
__device__ __forceinline__ int point_key(const PoseF& P, float x, float y, bool live, PointRec& r) {
  image_point(P, x, y, r);
  return image_key(P, P.ox, P.oy, r, live);
}

// a4 (2/2), global-memory grid: one 32-byte record per cell
__device__ __forceinline__ void lookup_point(const PoseF& P, const float4* __restrict__ rec, float x, float y,
                                             bool live, PointRec& r) {
  const int key = point_key(P, x, y, live, r);
  r.A = rec[2 * key];
  r.B = rec[2 * key + 1];
}

struct Acc2D {            // the running sums of one thread (unscaled: d2 is applied in acc_store)
  float h0, h1, h2, h3, h4, h5;   // Hxx Hxy Hyy Hxt Hyt Htt
  float g0, g1, g2, s;            // gx gy gt score
  int n_wave;                     // hits of the whole wave, kept on the scalar unit
};
__device__ __forceinline__ void acc_zero(Acc2D& a) {
  a.h0 = a.h1 = a.h2 = a.h3 = a.h4 = a.h5 = a.g0 = a.g1 = a.g2 = a.s = 0.f;
  a.n_wave = 0;
}
__device__ __forceinline__ void acc_store(const Acc2D& a, float d2, float* out /*[11]*/) {
  out[0] = a.h0 * d2; out[1] = a.h1 * d2; out[2] = a.h2 * d2; out[3] = a.h3 * d2; out[4] = a.h4 * d2;
  out[5] = a.h5 * d2; out[6] = a.g0 * d2; out[7] = a.g1 * d2; out[8] = a.g2 * d2; out[9] = a.s;
  out[10] = (threadIdx.x & 63) == 0 ? (float)a.n_wave : 0.f;     // the wave's hits, once
}

// a5+a6: Mahalanobis score, SE(2) Jacobian, gradient / Hessian terms into the thread's sums
template <int MODE>
__device__ __forceinline__ void accumulate_point(const PoseF& P, const PointRec& r, Acc2D& acc) {
  const bool hit = r.B.z > 0.f;                                   // n > 0: a finalised cell
  const float a = r.A.z, b = r.A.w, c = r.B.y;
  const float qx = r.px - r.A.x, qy = r.py - r.A.y;
  const float vx = fmaf(a, qx, b * qy), vy = fmaf(b, qx, c * qy);   // Sigma^-1 q
  const float m = fmaf(qx, vx, qy * vy);
#if defined(NDT_BATCH_ABLATE) && (NDT_BATCH_ABLATE & 4)      // tools only: no transcendental
  const float s = hit ? fmaf(P.nhd2, m, 1.f) : 0.f;
#else
  const float s = hit ? __builtin_amdgcn_exp2f(fmaf(P.nhd2, m, P.lg_d1)) : 0.f;
#endif
  const float jx = P.ty - r.py, jy = r.px - P.tx;                  // dR/dtheta p = perp(p' - t)
  const float vt = fmaf(vx, jx, vy * jy);
  const float ux = fmaf(a, jx, b * jy), uy = fmaf(b, jx, c * jy);   // Sigma^-1 j
  float hxx = a, hxy = b, hyy = c, hxt = ux, hyt = uy, htt = fmaf(jx, ux, jy * uy);
  if (MODE == 1) {   // full Newton Hessian (Biber / Magnusson)
    const float dvx = -P.d2 * vx, dvy = -P.d2 * vy, dvt = -P.d2 * vt;
    hxx = fmaf(dvx, vx, hxx);
    hxy = fmaf(dvx, vy, hxy);
    hyy = fmaf(dvy, vy, hyy);
    hxt = fmaf(dvt, vx, hxt);
    hyt = fmaf(dvt, vy, hyt);
    htt = fmaf(dvt, vt, htt) + fmaf(vy, jx, -vx * jy);
  }
  acc.h0 = fmaf(s, hxx, acc.h0); acc.h1 = fmaf(s, hxy, acc.h1); acc.h2 = fmaf(s, hyy, acc.h2);
  acc.h3 = fmaf(s, hxt, acc.h3); acc.h4 = fmaf(s, hyt, acc.h4); acc.h5 = fmaf(s, htt, acc.h5);
  acc.g0 = fmaf(s, vx, acc.g0);  acc.g1 = fmaf(s, vy, acc.g1);  acc.g2 = fmaf(s, vt, acc.g2);
  acc.s += s;
  acc.n_wave += (int)__popcll(__ballot(hit));                     // s_bcnt1 on the compare mask: no VALU
}

// 128-byte state copy without a struct temporary (a by-value IterState lands in scratch)
__device__ __forceinline__ void copy_state(IterState* dst, const IterState* src, int have_partials) {
  static_assert(sizeof(IterState) == 128, "IterState is copied as 8 x 16 bytes");
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int j = 0; j < 8; ++j) d4[j] = s4[j];
  if (have_partials >= 0) dst->have_partials = have_partials;
}

// The 11 sums of a wave through LDS instead of 11 DPP trees (66 dependent steps): every lane
// parks its 11 values ([j][lane], row stride 68 floats: at most 2-way bank conflicts on the way
// back), then lane 4j+q adds the 16 values of accumulator j whose lane index is q mod 4 (two
// independent chains), and the quad folds with two DPP steps.  Same wave throughout: LDS runs a
// wave's operations in order, so no barrier.  Returns accumulator (lane >> 2)'s total, valid in
// lanes < 44; a fixed order, so sums stay reproducible.
constexpr int kSumRowStride = 68;
__device__ __forceinline__ float wave_reduce11_lds(const float* acc, float* s_t, int lane) {
#pragma unroll
  for (int j = 0; j < kNumAcc - 1; ++j) s_t[j * kSumRowStride + lane] = acc[j];
  __builtin_amdgcn_wave_barrier();
  const int j = lane >> 2, q = lane & 3;
  float a = 0.f, b = 0.f;
  if (lane < 4 * (kNumAcc - 1)) {
    const float* row = s_t + j * kSumRowStride + q;
#pragma unroll
    for (int k = 0; k < 16; k += 2) { a += row[4 * k]; b += row[4 * k + 4]; }
  }
  float v = a + b;
  v += dpp_mov<0xB1, 0xf>(v);
  v += dpp_mov<0x4E, 0xf>(v);
  __builtin_amdgcn_wave_barrier();                     // s_t is rewritten by this wave next iteration
  return v;
}

// ---------------------------------------------------------------- a4-a8 iterate kernel
// Launch k (parity = k & 1) consumes state[parity^1] and partials[parity^1] written by
// launch k-1 and produces state[parity], partials[parity].  Always kMaxBlocks workgroups
// (one per CU); a workgroup without points contributes a zero partial row.
//
// The iteration is a serial dependence chain (reduce -> solve -> transform -> gather ->
// reduce), so the kernel is written to keep the number of dependent memory round trips at
// two: everything the prologue needs (previous state, partial rows, first source point) is
// requested up front in one batch, then the cell-record gather.
// EXP is an ablation mask for tools/exp_iter.hip only (1: no reduce/solve, 2: no body,
// 4: no epilogue tree, 8: empty); the library instantiates EXP = 0.
template <int MODE, int EXP = 0, int THREADS = kBlock, int NG = 1>
__global__ __launch_bounds__(THREADS) void k_iterate(const AlignStatic* __restrict__ st,
                                                    const AlignCall* __restrict__ call,
                                                    AlignDyn* __restrict__ dyn, int parity) {
  __shared__ double s_red[kNumAcc];
  __shared__ float s_wave[THREADS / 64][kNumAcc];
  __shared__ float s_t[THREADS / 64][(kNumAcc - 1) * kSumRowStride];
  if (EXP & 8) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const IterState* prev = &dyn->state[parity ^ 1];
  IterState* cur = &dyn->state[parity];
  const bool writer = (blockIdx.x == 0) && (tid == 0);

  // ---- batch 1 of loads: previous state (scalar), partial rows (vector), first point
  const double ps_pose0 = prev->pose[0], ps_pose1 = prev->pose[1], ps_pose2 = prev->pose[2];
  const int ps_iter = prev->iter, ps_done = prev->done, ps_have = prev->have_partials, ps_launch = prev->pad;
  const SolveParams prm = st->prm;       // read-only: scalar loads, all in this first batch
  const GridDev G = st->grid;
  const int n = call->n;
  const int fixed_iterations = call->fixed_iterations;
  const float* __restrict__ sx = call->sx;
  const float* __restrict__ sy = call->sy;
  IterState* const host_state = call->host_state;
  int* const host_flag = call->host_flag;
  float4 pv[3];
  if (!(EXP & 1) && wave < 4) {          // waves 0..3 own the 12 partial rows
    const float* part = &dyn->partials[parity ^ 1][0][0];
#pragma unroll
    for (int v = 0; v < 3; ++v)
      pv[v] = *reinterpret_cast<const float4*>(part + (wave * 3 + v) * kMaxBlocks + lane * 4);
  }
  // Pin the read-only scalars here: without this hipcc sinks their s_loads below the
  // `done` branch and they become a third dependent round trip.
  asm volatile("" ::"s"(G.ox), "s"(G.oy), "s"(G.inv_c), "s"(G.W), "s"(G.H), "s"(G.rec),
               "s"(prm.d1), "s"(prm.d2), "s"(prm.min_hits), "s"(prm.max_iterations), "s"(prm.eps_trans),
               "s"(prm.eps_rot), "s"(prm.step_max_trans), "s"(prm.step_max_rot), "s"(prm.step_scale), "s"(ps_pose0), "s"(ps_pose1),
               "s"(ps_pose2), "s"(ps_iter), "s"(ps_done), "s"(ps_have), "s"(ps_launch), "s"(fixed_iterations), "s"(host_state),
               "s"(host_flag));
  const int stride = kMaxBlocks * THREADS;
  int i = blockIdx.x * THREADS + tid;
  float x = 0.f, y = 0.f, x1 = 0.f, y1 = 0.f;
  if (i < n) { x = sx[i]; y = sy[i]; }
  if (i + stride < n) { x1 = sx[i + stride]; y1 = sy[i + stride]; }

  if (ps_done) {                         // uniform: a finished alignment just carries its state
    if (writer) {
      copy_state(cur, prev, -1);
      // the launch that finished the loop is complete (this one started after it) and left n = 0
      // behind: nothing reads the source arrays any more - tell the host it may hand them back
      if (host_flag) __hip_atomic_store(host_flag + 2, call->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  double pose[3] = {ps_pose0, ps_pose1, ps_pose2};
  int iter = ps_iter;
  if (ps_have) {
    double H[6], g[3], score = 0.0;
    int n_hit = 0, status = 0;
    bool done = false;
    if (!(EXP & 1)) {
      // ---- prologue: fixed-order reduction (wave w owns sums 3w..3w+2), then the solve.
      // Each lane folds its 4 blocks per row in float64 and parks the 3 values in LDS ([row][lane]);
      // lane 16v+q adds the 4 values of row v whose lane index is q mod 16, four DPP steps fold the
      // 16 lanes - instead of three 6-step float64 DPP trees on the critical path.
      if (wave < 4) {
        double* t = reinterpret_cast<double*>(s_t[wave]);     // 3 x 66 doubles fit in the epilogue's buffer
#pragma unroll
        for (int v = 0; v < 3; ++v)
          t[v * 66 + lane] = (((double)pv[v].x + (double)pv[v].y) + (double)pv[v].z) + (double)pv[v].w;
        __builtin_amdgcn_wave_barrier();
        double a = 0.0;
        if (lane < 48) {
          const double* row = t + (lane >> 4) * 66 + (lane & 15);
          a = (row[0] + row[16]) + (row[32] + row[48]);
        }
        a += dpp_mov<0xB1, 0xf>(a);
        a += dpp_mov<0x4E, 0xf>(a);
        a += dpp_mov<0x124, 0xf>(a);
        a += dpp_mov<0x128, 0xf>(a);                           // every lane of the 16-lane row holds its total
        if ((lane & 15) == 0 && lane < 48) s_red[wave * 3 + (lane >> 4)] = a;
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 6; ++j) H[j] = s_red[j];
#pragma unroll
      for (int j = 0; j < 3; ++j) g[j] = s_red[6 + j];
      score = s_red[9];
      n_hit = (int)(s_red[10] + 0.5);
      done = gn_update(pose, H, g, n_hit, iter, status, prm, fixed_iterations, score, &dyn->ls[parity ^ 1],
                       &dyn->ls[parity], writer);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) H[j] = 0.0;
      g[0] = g[1] = g[2] = 0.0;
      iter += 1;
      done = iter >= fixed_iterations;
    }
    if (writer) {
      IterState o;
      o.pose[0] = pose[0]; o.pose[1] = pose[1]; o.pose[2] = pose[2];
#pragma unroll
      for (int j = 0; j < 6; ++j) o.H[j] = H[j];
#pragma unroll
      for (int j = 0; j < 3; ++j) o.g[j] = g[j];
      o.score = score;
      o.n_hit = n_hit;
      o.iter = iter;
      o.status = status;
      o.done = done ? 1 : 0;
      o.have_partials = 1;
      o.pad = ps_launch + 1;
      *cur = o;
      if (host_flag) {                   // tell the host directly
        if (done) {                      // state and this launch's number first, then the flag
          *host_state = o;
          __hip_atomic_store(host_flag + 1, ps_launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          const_cast<AlignCall*>(call)->n = 0;     // the launches enqueued past the end load no points
          __threadfence_system();
          __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {                         // progress: which launch this is
          __hip_atomic_store(host_flag + 1, ps_launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (done) return;                    // uniform
  } else if (writer) {
    copy_state(cur, prev, 1);
  }

  // ---- body: per-point terms at `pose`
  double sn_d, cs_d;
  sincos_wrapped(pose[2], &sn_d, &cs_d);
  const float4* __restrict__ rec = G.rec;
  const PoseF P = make_pose((float)cs_d, (float)sn_d, (float)pose[0], (float)pose[1], G.ox, G.oy, G.inv_c, G.W,
                            G.H, prm.d1, prm.d2);
  Acc2D A;
  acc_zero(A);

  // two points in flight per thread: both gathers are issued before either is consumed
  while (!(EXP & 2) && i < n) {
    const int i2 = i + 2 * stride;
    float xn0 = 0.f, yn0 = 0.f, xn1 = 0.f, yn1 = 0.f;
    if (i2 < n) { xn0 = sx[i2]; yn0 = sy[i2]; }
    if (i2 + stride < n) { xn1 = sx[i2 + stride]; yn1 = sy[i2 + stride]; }
    PointRec r0, r1;
    const bool two = (i + stride) < n;
    if (NG == 1) {
      lookup_point(P, rec, x, y, true, r0);
      lookup_point(P, rec, x1, y1, two, r1);
      accumulate_point<MODE>(P, r0, A);
      accumulate_point<MODE>(P, r1, A);
    } else {
      // overlapping grids (Biber): the same image point scores against every grid
      image_point(P, x, y, r0);
      image_point(P, x1, y1, r1);
      const int ncell = G.W * G.H;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const int k0 = q * ncell + image_key(P, G.gx[q], G.gy[q], r0, true);
        const int k1 = q * ncell + image_key(P, G.gx[q], G.gy[q], r1, two);
        r0.A = rec[2 * k0]; r0.B = rec[2 * k0 + 1];
        r1.A = rec[2 * k1]; r1.B = rec[2 * k1 + 1];
        accumulate_point<MODE>(P, r0, A);
        accumulate_point<MODE>(P, r1, A);
      }
    }
    x = xn0; y = yn0; x1 = xn1; y1 = yn1; i = i2;
  }
  float acc[kNumAcc];
  acc_store(A, prm.d2, acc);
  acc[11] = 0.f;

  // ---- epilogue: wave tree -> LDS -> one partial row per block
  if (EXP & 4) {
    if (tid < kNumAcc) dyn->partials[parity][tid][blockIdx.x] = acc[tid & 1];
    return;
  }
  {
    const float r = wave_reduce11_lds(acc, s_t[wave], lane);
    if ((lane & 3) == 0 && lane < 4 * (kNumAcc - 1)) s_wave[wave][lane >> 2] = r;
  }
  __syncthreads();
  if (tid < kNumAcc) {
    float r = 0.f;
    if (tid < kNumAcc - 1) {
#pragma unroll
      for (int w = 0; w < THREADS / 64; ++w) r += s_wave[w][tid];     // fixed order
    }
    dyn->partials[parity][tid][blockIdx.x] = r;
  }
}

// ------------------------------------------------------------- scan format (section 8f rank 4)
// Range/bearing -> Cartesian SoA.  One sincos per beam from the beam index (no accumulated
// angle error); invalid returns become NaN points, which every kernel above ignores.
__global__ __launch_bounds__(kBlock) void k_polar_to_points(const float* __restrict__ r, size_t n, double angle_min,
                                                             double angle_inc, float range_min, float range_max,
                                                             float* __restrict__ x, float* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float ri = r[i];
    double s, c;
    sincos(fma((double)i, angle_inc, angle_min), &s, &c);
    const bool ok = isfinite(ri) & (ri >= range_min) & (ri <= range_max);
    x[i] = ok ? (float)((double)ri * c) : NAN;
    y[i] = ok ? (float)((double)ri * s) : NAN;
  }
}

}  // namespace ndt
