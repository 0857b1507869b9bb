// Short scans on the single-pair path: one workgroup runs the whole Gauss-Newton loop.
//
// A planar lidar delivers 360 - 4000 returns per scan (BASELINE config 1: 1000).  At that size
// one launch per iteration (k_iterate) is all launch boundary: 4.5 us per iteration of which the
// points are 0.1 us.  With <= 4096 source points the loop fits one workgroup of 256 or 1024
// threads: the points stay in registers, records come from the cached global-memory grid (L1/L2
// hits after the first iteration), the reduction is per-wave LDS sums -> wave 0 in float64 (the
// loop-closure kernel's scheme, ndt2d_batch.hpp), and convergence is decided on the device - no
// launch boundary, no host polling.  The finishing thread writes the state to device memory and
// to pinned host memory, then raises the host flag.  Same per-point code and the same update rule
// (gn_update) as the other two kernels.
#pragma once
#include "ndt2d_kernels.hpp"

namespace ndt {

// One CU has four SIMDs: with more than four waves every VALU-heavy section (the DPP tree, the
// per-point math) is issue-bound, so a scan of up to 2048 points runs on 256 threads (one wave per
// SIMD, 8 points per thread) and only longer ones on 1024.  Beyond 4096 points the workgroup's
// single CU loses to k_iterate's 256 (measured: 5.0 vs 5.5 us per iteration at 4000 points before
// these changes, 7.2 vs 5.5 at 8192).
constexpr int kSmallPts = 8;                                   // source points per thread, in registers
constexpr int kSmallThreadsLo = 256, kSmallThreadsHi = 1024;
constexpr int kSmallLoPoints = kSmallThreadsLo * kSmallPts;    // 2048
constexpr int kSmallMaxPoints = 4096;

template <int MODE, int NG, int kSmallThreads>
__global__ __launch_bounds__(kSmallThreads) void k_align_small(const AlignStatic* __restrict__ st,
                                                               const float* __restrict__ sx,
                                                               const float* __restrict__ sy, int n, double p0, double p1,
                                                               double p2, int fixed_iterations,
                                                               IterState* __restrict__ dev_state,
                                                               IterState* __restrict__ host_state,
                                                               int* __restrict__ host_flag) {
  constexpr int kSmallWaves = kSmallThreads / 64;
  __shared__ float s_red[kSmallWaves][kNumAcc];
  __shared__ float s_t[kSmallWaves][(kNumAcc - 1) * kSumRowStride];
  __shared__ double s_bc[16];       // pose(3) | H(6) g(3) score n_hit of the last evaluation
  __shared__ int s_misc[4];         // done, iter, status
  __shared__ LineSearch s_ls;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SolveParams prm = st->prm;
  const GridDev G = st->grid;
  const float4* __restrict__ rec = G.rec;

  // the scan, once: point u of this thread is tid + u * kSmallThreads
  float px[kSmallPts], py[kSmallPts];
#pragma unroll
  for (int u = 0; u < kSmallPts; ++u) {
    const int i = tid + u * kSmallThreads;
    px[u] = i < n ? sx[i] : 0.f;
    py[u] = i < n ? sy[i] : 0.f;
  }
  const int trips = __builtin_amdgcn_readfirstlane((n + kSmallThreads - 1) / kSmallThreads);
  double pose[3] = {p0, p1, wrap_angle(p2)};
  if (tid == 0) { s_ls.valid = 0; s_ls.trials = 0; s_misc[1] = 0; s_misc[2] = 0; }
  __syncthreads();

#ifdef NDT_SMALL_PROFILE          // tools only: where an iteration's time goes (100 MHz ticks, wave 0)
  unsigned long long t_body = 0, t_tree = 0, t_solve = 0, t_sync = 0, t_mark = wall_clock64();
#define NDT_TICK(acc) do { const unsigned long long t_now = wall_clock64(); acc += t_now - t_mark; t_mark = t_now; } while (0)
#else
#define NDT_TICK(acc) do {} while (0)
#endif
  for (;;) {
    float acc[kNumAcc];
    {
      double sn_d, cs_d;
      sincos_wrapped(pose[2], &sn_d, &cs_d);
      auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
      const PoseF P = make_pose(uni((float)cs_d), uni((float)sn_d), uni((float)pose[0]), uni((float)pose[1]), G.ox, G.oy,
                                G.inv_c, G.W, G.H, prm.d1, prm.d2);
      Acc2D A;
      acc_zero(A);
      const int ncell = G.W * G.H;
      // kFly points in flight: their gathers are issued before any is consumed (the 256-thread
      // variant has one wave per SIMD and registers to spare)
      constexpr int kFly = (kSmallThreads == kSmallThreadsLo && NG == 1) ? 4 : 2;
#pragma unroll
      for (int u = 0; u < kSmallPts; u += kFly) {
        if (u < trips) {                                   // uniform
          PointRec r[kFly];
          bool live[kFly];
#pragma unroll
          for (int f = 0; f < kFly; ++f) live[f] = tid + (u + f) * kSmallThreads < n;
          if (NG == 1) {
#pragma unroll
            for (int f = 0; f < kFly; ++f) lookup_point(P, rec, px[u + f], py[u + f], live[f], r[f]);
#pragma unroll
            for (int f = 0; f < kFly; ++f) accumulate_point<MODE>(P, r[f], A);
          } else {
#pragma unroll
            for (int f = 0; f < kFly; ++f) image_point(P, px[u + f], py[u + f], r[f]);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
#pragma unroll
              for (int f = 0; f < kFly; ++f) {
                const int k = q * ncell + image_key(P, G.gx[q], G.gy[q], r[f], live[f]);
                r[f].A = rec[2 * k]; r[f].B = rec[2 * k + 1];
              }
#pragma unroll
              for (int f = 0; f < kFly; ++f) accumulate_point<MODE>(P, r[f], A);
            }
          }
        }
      }
      acc_store(A, prm.d2, acc);
      acc[11] = 0.f;
    }
    NDT_TICK(t_body);
    {
      const float r = wave_reduce11_lds(acc, s_t[wave], lane);
      if ((lane & 3) == 0 && lane < 4 * (kNumAcc - 1)) s_red[wave][lane >> 2] = r;
    }
    __syncthreads();
    NDT_TICK(t_tree);
    if (wave == 0) {
      // lane j < 11 sums column j over the waves in a fixed order, in float64, and parks it in
      // LDS; every lane then reads the 11 totals back (broadcast reads: one wait instead of a
      // chain of cross-lane shuffles)
      if (lane < kNumAcc - 1) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kSmallWaves; ++w) tot += (double)s_red[w][lane];
        s_bc[3 + lane] = tot;
      }
      __builtin_amdgcn_wave_barrier();                 // same wave: LDS executes its operations in order
      double H[6], g[3];
#pragma unroll
      for (int j = 0; j < 6; ++j) H[j] = s_bc[3 + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) g[j] = s_bc[9 + j];
      const double score = s_bc[12];
      const int n_hit = (int)(s_bc[13] + 0.5);
      int iter = s_misc[1], status = 0;
      const bool done = gn_update(pose, H, g, n_hit, iter, status, prm, fixed_iterations, score, &s_ls, &s_ls, lane == 0);
      if (lane == 0) {
        s_bc[0] = pose[0]; s_bc[1] = pose[1]; s_bc[2] = pose[2];
        s_misc[0] = done ? 1 : 0;
        s_misc[1] = iter;
        s_misc[2] = status;
      }
    }
    NDT_TICK(t_solve);
    __syncthreads();
    pose[0] = s_bc[0]; pose[1] = s_bc[1]; pose[2] = s_bc[2];
    NDT_TICK(t_sync);
    if (__builtin_amdgcn_readfirstlane(s_misc[0])) break;
  }
#ifdef NDT_SMALL_PROFILE
  if (tid == 0)
    printf("k_align_small n=%d iters=%d: body %.2f tree+barrier %.2f solve %.2f barrier+pose %.2f us/iter\n", n, s_misc[1],
           0.01 * t_body / s_misc[1], 0.01 * t_tree / s_misc[1], 0.01 * t_solve / s_misc[1], 0.01 * t_sync / s_misc[1]);
#endif
#undef NDT_TICK

  if (tid == 0) {
    IterState o;
    o.pose[0] = pose[0]; o.pose[1] = pose[1]; o.pose[2] = pose[2];
#pragma unroll
    for (int j = 0; j < 6; ++j) o.H[j] = s_bc[3 + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) o.g[j] = s_bc[9 + j];
    o.score = s_bc[12];
    o.n_hit = (int)(s_bc[13] + 0.5);
    o.iter = s_misc[1];
    o.status = s_misc[2];
    o.done = 1;
    o.have_partials = 0;
    o.pad = 0;
    *dev_state = o;
    if (host_flag) {
      *host_state = o;
      __threadfence_system();
      __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace ndt
