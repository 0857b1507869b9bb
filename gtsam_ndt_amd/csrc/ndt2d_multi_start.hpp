// Multi-start alignment: up to kMaxStarts independent Gauss-Newton loops of the SAME scan against the
// SAME cached grid, from different initial poses, carried by one launch chain.
//
// Why: a single 100k-point alignment is bound by the launch boundary (1.7 us) and the reduce + solve
// prologue (1.1 us) of k_iterate, not by its 3.2 MB of algorithmic traffic per iteration
// (DESIGN.md section 5.1): the chip idles through most of a launch.  Here one launch carries the
// iteration of every start: the grid is (256 workgroups) x (subsets of NH starts), workgroup (b, s)
// does for its NH starts exactly what workgroup b of k_iterate does for one - so the boundary is
// paid once per launch for all starts and their prologues, gathers and reductions overlap on the
// CUs (4 or more workgroups resident per CU instead of one).  It is also what a loop-closure front
// end wants from a poor initial guess: several starts around it, best score wins - without the
// pyramid's extra grids.
//
// Contract: start k's result is what ndt2d_align_dev returns for init_poses[k] on the
// launch-per-iteration path, bit for bit: same thread -> point assignment, same per-thread
// accumulation order, same reduction trees, same update (the shared device functions of
// ndt2d_kernels.hpp); a start that has finished is frozen while the others go on.
#pragma once
#include "ndt2d_kernels.hpp"

namespace ndt {

constexpr int kMaxStarts = 64;

struct AlignDynMulti {
  IterState state[2][kMaxStarts];
  float partials[2][kMaxStarts][kNumAcc][kMaxBlocks];
  LineSearch ls[2][kMaxStarts];
  int launch[2];          // ping-pong launch counter (what the host sees as progress)
  int subsets_done;       // subsets (blockIdx.y) all of whose starts have finished
  int pad;
  // multi-scan calls (ndt2d_align_multi_scan_dev): every start has its own source scan
  const float* sx[kMaxStarts];
  const float* sy[kMaxStarts];
  int n[kMaxStarts];
  // split chains (k_multi_solve + k_multi_body): the pose each start's body uses, and how many starts are through
  struct BodyPose { float cs, sn, tx, ty; int done; int pad[3]; } posef[2][kMaxStarts];
  int starts_done;
  int pad2[3];
};

struct StartPoses {
  double p[kMaxStarts][3];
};
struct StartScans {
  const float* sx[kMaxStarts];
  const float* sy[kMaxStarts];
  int n[kMaxStarts];
};

// Per-call part of the context, written from kernel arguments; slots >= m never run.
__global__ void k_begin_multi(AlignCall* __restrict__ call, AlignDynMulti* __restrict__ dyn, const float* sx,
                              const float* sy, int n, StartPoses poses, StartScans scans, int m, int fixed_iterations,
                              IterState* host_state, int* host_flag, int seq) {
  const int h = threadIdx.x;
  if (blockIdx.x != 0 || h >= kMaxStarts) return;
  dyn->sx[h] = scans.sx[h]; dyn->sy[h] = scans.sy[h]; dyn->n[h] = h < m ? scans.n[h] : 0;
  dyn->posef[0][h].done = dyn->posef[1][h].done = h < m ? 0 : 1;     // split chains: slots >= m are never evaluated
  if (h == 0) {
    call->seq = seq;
    call->pad = m;
    call->sx = sx;
    call->sy = sy;
    call->n = n;
    call->fixed_iterations = fixed_iterations;
    call->host_state = host_state;
    call->host_flag = host_flag;
    dyn->launch[0] = 0; dyn->launch[1] = 0;
    dyn->subsets_done = 0;
    dyn->starts_done = 0;
  }
  IterState s = {};
  if (h < m) {
    s.pose[0] = poses.p[h][0]; s.pose[1] = poses.p[h][1]; s.pose[2] = wrap_angle(poses.p[h][2]);
  } else {
    s.done = 1;                 // unused slot: never evaluated
  }
  dyn->state[1][h] = s;         // launch 0 has parity 0 and reads slot 1
  dyn->state[0][h] = IterState{};
  dyn->ls[0][h] = LineSearch{};
  dyn->ls[1][h] = LineSearch{};
}

// what the body needs of one start, parked in LDS by the wave that solved it
struct StartPose {
  float cs, sn, tx, ty;
};

// Launch k (parity = k & 1) consumes state[parity ^ 1][*] and partials[parity ^ 1][*] and produces
// state[parity][*], partials[parity][*] - the scheme of k_iterate, per start.  Workgroup
// (blockIdx.x, blockIdx.y) owns starts blockIdx.y * NH .. + NH - 1 on block blockIdx.x's points.
// SHARED: every start aligns the same scan (multi-start: the points are loaded once per workgroup and
// looked up under NH poses); otherwise every start has its own scan (multi-scan: one point loop per start).
// Two points of one thread against Biber's four overlapping grids, in k_iterate's order (both image points, then per grid
// both lookups and both sums): what keeps a start of these chains bit-identical to its single alignment with the option.
template <int MODE>
__device__ __forceinline__ void score_two_overlapped(const PoseF& P, const GridDev& G, const float4* __restrict__ rec, float x, float y,
                                                     float x1, float y1, bool two, Acc2D& A) {
  PointRec r0, r1;
  image_point(P, x, y, r0);
  image_point(P, x1, y1, r1);
  const int ncell = G.W * G.H;
#pragma unroll
  for (int q = 0; q < kMaxGrids; ++q) {
    const int k0 = q * ncell + image_key(P, G.gx[q], G.gy[q], r0, true);
    const int k1 = q * ncell + image_key(P, G.gx[q], G.gy[q], r1, two);
    r0.A = rec[2 * k0]; r0.B = rec[2 * k0 + 1];
    r1.A = rec[2 * k1]; r1.B = rec[2 * k1 + 1];
    accumulate_point<MODE>(P, r0, A);
    accumulate_point<MODE>(P, r1, A);
  }
}

template <int MODE, int NH, int THREADS, bool SHARED = true, int NG = 1>
__global__ __launch_bounds__(THREADS) void k_iterate_multi(const AlignStatic* __restrict__ st,
                                                           const AlignCall* __restrict__ call,
                                                           AlignDynMulti* __restrict__ dyn, int parity) {
  constexpr int kWaves = THREADS / 64;
  static_assert(NH <= kWaves, "one wave per start solves");
  __shared__ double s_red[NH][kNumAcc];
  __shared__ float s_wave[kWaves][NH][kNumAcc];
  __shared__ float s_t[kWaves][(kNumAcc - 1) * kSumRowStride];
  __shared__ StartPose s_pose[NH];
  __shared__ int s_done[NH];          // after this launch's update
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hb = blockIdx.y * NH;     // first start of this workgroup's subset
  const IterState* prev = dyn->state[parity ^ 1] + hb;
  IterState* cur = dyn->state[parity] + hb;
  const bool block0 = blockIdx.x == 0;
  const bool herald = block0 && blockIdx.y == 0 && tid == 0;     // the one thread that talks to the host

  // ---- batch 1 of loads: read-only scalars, the previous `done` flags, partial rows, first points
  const SolveParams prm = st->prm;
  const GridDev G = st->grid;
  const int armed = call->n;                   // 0: the call is over, no launch reads a source array any more
  const int fixed_iterations = call->fixed_iterations;
  // this subset's scans (SHARED: the one scan of the call)
  const float* __restrict__ sxs[NH];
  const float* __restrict__ sys[NH];
  int ns[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    sxs[h] = SHARED ? call->sx : dyn->sx[hb + h];
    sys[h] = SHARED ? call->sy : dyn->sy[hb + h];
    ns[h] = armed ? (SHARED ? armed : dyn->n[hb + h]) : 0;
  }
  const int n = ns[0];
  const float* __restrict__ sx = sxs[0];
  const float* __restrict__ sy = sys[0];
  IterState* const host_state = call->host_state;
  int* const host_flag = call->host_flag;
  const int launch = dyn->launch[parity ^ 1] + 1;
  const int subsets_done = dyn->subsets_done;       // as of the previous launch (this launch's increments may race in)
  int pdone[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) pdone[h] = prev[h].done;
  // group g = 4 h + q holds rows 3q .. 3q + 2 of start h; wave w owns groups w, w + kWaves, ...
  constexpr int kGroups = 4 * NH;
  constexpr int kPerWave = (kGroups + kWaves - 1) / kWaves;
  float4 pv[kPerWave][3];
#pragma unroll
  for (int j = 0; j < kPerWave; ++j) {
    const int g = wave + j * kWaves;
    if (g < kGroups) {
      const float* part = &dyn->partials[parity ^ 1][hb + (g >> 2)][0][0];
#pragma unroll
      for (int v = 0; v < 3; ++v)
        pv[j][v] = *reinterpret_cast<const float4*>(part + ((g & 3) * 3 + v) * kMaxBlocks + lane * 4);
    }
  }
  // the state wave h updates: requested now, used after the reduction - not a dependent round trip
  // behind the barrier
  double q_pose[3] = {0.0, 0.0, 0.0};
  int q_iter = 0, q_done = 1, q_have = 0;
  if (wave < NH) {
    q_pose[0] = prev[wave].pose[0]; q_pose[1] = prev[wave].pose[1]; q_pose[2] = prev[wave].pose[2];
    q_iter = prev[wave].iter; q_done = prev[wave].done; q_have = prev[wave].have_partials;
  }
  asm volatile("" ::"s"(G.ox), "s"(G.oy), "s"(G.inv_c), "s"(G.W), "s"(G.H), "s"(G.rec), "s"(prm.d1), "s"(prm.d2),
               "s"(prm.min_hits), "s"(prm.max_iterations), "s"(prm.eps_trans), "s"(prm.eps_rot), "s"(prm.step_max_trans),
               "s"(prm.step_max_rot), "s"(prm.step_scale), "s"(launch), "s"(subsets_done), "s"(fixed_iterations),
               "s"(host_state), "s"(host_flag));
  const int stride = kMaxBlocks * THREADS;
  int i = blockIdx.x * THREADS + tid;
  float x = 0.f, y = 0.f, x1 = 0.f, y1 = 0.f;
  if (i < n) { x = sx[i]; y = sy[i]; }
  if (i + stride < n) { x1 = sx[i + stride]; y1 = sy[i + stride]; }

  if (herald) {
    dyn->launch[parity] = launch;
    // progress, while anything is still running.  Not from the launches past the end: they may execute
    // after the host has reset the flags for its NEXT call, and a stale progress number there makes that
    // call's feeding loop run ahead of its own chain (found by tools/soak_round2.py: one call in a few
    // thousand hit the loop's launch cap).
    if (host_flag && subsets_done != (int)gridDim.y)
      __hip_atomic_store(host_flag + 1, launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  bool sub_prev_done = true;
#pragma unroll
  for (int h = 0; h < NH; ++h) sub_prev_done = sub_prev_done && pdone[h];
  if (sub_prev_done) {                   // uniform: this subset has finished - carry its states
    if (block0 && tid < NH) copy_state(&cur[tid], &prev[tid], -1);
    if (herald && host_flag && subsets_done == (int)gridDim.y) {
      // Every start had finished (and written its final state to the host) before this launch began.
      // First such launch: stop the launches behind it from loading points, then raise the flag.
      // Second: the first one is complete, nothing reads the source arrays any more.
      if (armed != 0) {
        const_cast<AlignCall*>(call)->n = 0;
        __threadfence_system();
        __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        __hip_atomic_store(host_flag + 2, call->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }

  // ---- prologue (1/2): fixed-order float64 reduction of the partial rows, the tree of k_iterate
  // row for row
#pragma unroll
  for (int j = 0; j < kPerWave; ++j) {
    const int g = wave + j * kWaves;
    if (g < kGroups) {                                        // uniform per wave
      double* t = reinterpret_cast<double*>(s_t[wave]);
#pragma unroll
      for (int v = 0; v < 3; ++v)
        t[v * 66 + lane] = (((double)pv[j][v].x + (double)pv[j][v].y) + (double)pv[j][v].z) + (double)pv[j][v].w;
      __builtin_amdgcn_wave_barrier();
      double a = 0.0;
      if (lane < 48) {
        const double* row = t + (lane >> 4) * 66 + (lane & 15);
        a = (row[0] + row[16]) + (row[32] + row[48]);
      }
      a += dpp_mov<0xB1, 0xf>(a);
      a += dpp_mov<0x4E, 0xf>(a);
      a += dpp_mov<0x124, 0xf>(a);
      a += dpp_mov<0x128, 0xf>(a);
      if ((lane & 15) == 0 && lane < 48) s_red[g >> 2][(g & 3) * 3 + (lane >> 4)] = a;
      __builtin_amdgcn_wave_barrier();                        // t is rewritten by the next group
    }
  }
  __syncthreads();

  // ---- prologue (2/2): wave h solves and updates start h and parks its pose for the body
  if (wave < NH) {                                            // uniform per wave
    const int h = wave;
    const bool writer = block0 && lane == 0;
    double pose[3] = {q_pose[0], q_pose[1], q_pose[2]};
    int done = q_done;
    if (done) {
      if (writer) copy_state(&cur[h], &prev[h], -1);
    } else if (q_have) {
      double H[6], g[3];
#pragma unroll
      for (int k = 0; k < 6; ++k) H[k] = s_red[h][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] = s_red[h][6 + k];
      const double score = s_red[h][9];
      const int n_hit = (int)(s_red[h][10] + 0.5);
      int iter = q_iter, status = 0;
      done = gn_update(pose, H, g, n_hit, iter, status, prm, fixed_iterations, score, &dyn->ls[parity ^ 1][hb + h],
                       &dyn->ls[parity][hb + h], writer) ? 1 : 0;
      if (writer) {
        IterState o;
        o.pose[0] = pose[0]; o.pose[1] = pose[1]; o.pose[2] = pose[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) o.H[k] = H[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) o.g[k] = g[k];
        o.score = score;
        o.n_hit = n_hit;
        o.iter = iter;
        o.status = status;
        o.done = done;
        o.have_partials = 1;
        o.pad = launch;
        cur[h] = o;
        if (host_flag && done) host_state[hb + h] = o;        // final states go to the host as they come
      }
    } else if (writer) {
      copy_state(&cur[h], &prev[h], 1);
    }
    if (lane == 0) {
      if (!done) {
        double sn_d, cs_d;
        sincos_wrapped(pose[2], &sn_d, &cs_d);
        s_pose[h] = StartPose{(float)cs_d, (float)sn_d, (float)pose[0], (float)pose[1]};
      }
      s_done[h] = done;
    }
  }
  __syncthreads();
  int done[NH];
  bool all_done = true;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    done[h] = __builtin_amdgcn_readfirstlane(s_done[h]);
    all_done = all_done && done[h];
  }
  if (all_done) {                                             // uniform: the subset's last start finished now
    if (block0 && tid == 0) {
      __threadfence_system();                                 // its final states first (written above, before the barrier)
      atomicAdd(&dyn->subsets_done, 1);
    }
    return;
  }

  // ---- body: every point against every live start's pose
  const float4* __restrict__ rec = G.rec;
  PoseF P[NH];
  Acc2D A[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    P[h] = make_pose(uni(s_pose[h].cs), uni(s_pose[h].sn), uni(s_pose[h].tx), uni(s_pose[h].ty), G.ox, G.oy, G.inv_c, G.W,
                     G.H, prm.d1, prm.d2);
    acc_zero(A[h]);
  }
  if constexpr (SHARED) {
    while (i < n) {
      const int i2 = i + 2 * stride;
      float xn0 = 0.f, yn0 = 0.f, xn1 = 0.f, yn1 = 0.f;
      if (i2 < n) { xn0 = sx[i2]; yn0 = sy[i2]; }
      if (i2 + stride < n) { xn1 = sx[i2 + stride]; yn1 = sy[i2 + stride]; }
      const bool two = (i + stride) < n;
      if constexpr (NG == 1) {
        PointRec r0[NH], r1[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          if (!done[h]) {                                     // uniform
            lookup_point(P[h], rec, x, y, true, r0[h]);
            lookup_point(P[h], rec, x1, y1, two, r1[h]);
          }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          if (!done[h]) {
            accumulate_point<MODE>(P[h], r0[h], A[h]);
            accumulate_point<MODE>(P[h], r1[h], A[h]);
          }
        }
      } else {
#pragma unroll
        for (int h = 0; h < NH; ++h)
          if (!done[h]) score_two_overlapped<MODE>(P[h], G, rec, x, y, x1, y1, two, A[h]);       // uniform
      }
      x = xn0; y = yn0; x1 = xn1; y1 = yn1; i = i2;
    }
  } else {
    // one point loop per live start, each over its own scan, in k_iterate's order
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (!done[h]) {                                         // uniform
        const float* __restrict__ px = sxs[h];
        const float* __restrict__ py = sys[h];
        const int nn = ns[h];
        int ii = blockIdx.x * THREADS + tid;
        float u = x, v = y, u1 = x1, v1 = y1;                 // start 0's first points were requested at the top
        if (h > 0) {
          u = v = u1 = v1 = 0.f;
          if (ii < nn) { u = px[ii]; v = py[ii]; }
          if (ii + stride < nn) { u1 = px[ii + stride]; v1 = py[ii + stride]; }
        }
        while (ii < nn) {
          const int i2 = ii + 2 * stride;
          float xn0 = 0.f, yn0 = 0.f, xn1 = 0.f, yn1 = 0.f;
          if (i2 < nn) { xn0 = px[i2]; yn0 = py[i2]; }
          if (i2 + stride < nn) { xn1 = px[i2 + stride]; yn1 = py[i2 + stride]; }
          const bool two = (ii + stride) < nn;
          if constexpr (NG == 1) {
            PointRec r0, r1;
            lookup_point(P[h], rec, u, v, true, r0);
            lookup_point(P[h], rec, u1, v1, two, r1);
            accumulate_point<MODE>(P[h], r0, A[h]);
            accumulate_point<MODE>(P[h], r1, A[h]);
          } else {
            score_two_overlapped<MODE>(P[h], G, rec, u, v, u1, v1, two, A[h]);
          }
          u = xn0; v = yn0; u1 = xn1; v1 = yn1; ii = i2;
        }
      }
    }
  }

  // ---- epilogue: per start, the 11 sums of the wave through LDS, then one partial row per block
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    if (!done[h]) {
      float acc[kNumAcc];
      acc_store(A[h], prm.d2, acc);
      acc[11] = 0.f;
      const float r = wave_reduce11_lds(acc, s_t[wave], lane);
      if ((lane & 3) == 0 && lane < 4 * (kNumAcc - 1)) s_wave[wave][h][lane >> 2] = r;
    }
  }
  __syncthreads();
  for (int k = tid; k < NH * kNumAcc; k += THREADS) {
    const int h = k / kNumAcc, j = k - h * kNumAcc;
    if (!s_done[h]) {
      float r = 0.f;
      if (j < kNumAcc - 1) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) r += s_wave[w][h][j];        // fixed order
      }
      dyn->partials[parity][hb + h][j][blockIdx.x] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Split chain for many starts: per iteration one launch of k_multi_solve (one workgroup per start: the
// reduction of that start's partial rows and its update - k_iterate's prologue, once instead of in every
// one of the 256 workgroups that evaluate the start) and one of k_multi_body (the evaluation).  In the
// fused kernel above the redundant prologues cost as much as the evaluation from a few dozen starts on
// (64 starts: 16384 solves per launch); here a second kernel boundary (1.7 us) buys them back.  Same
// arithmetic, same order: results are bit-identical to the fused chain and to the single-start path.
// Launch pair k: solve reads state[p^1], partials[p^1], writes state[p], posef[p]; body reads posef[p],
// writes partials[p]  (p = k & 1).
__global__ __launch_bounds__(kBlock) void k_multi_solve(const AlignStatic* __restrict__ st, const AlignCall* __restrict__ call,
                                                         AlignDynMulti* __restrict__ dyn, int parity) {
  __shared__ double s_red[kNumAcc];
  __shared__ double s_t[4][3 * 66];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, m = call->pad;
  const IterState* prev = &dyn->state[parity ^ 1][h];
  IterState* cur = &dyn->state[parity][h];
  const bool writer = tid == 0;
  const bool herald = writer && h == 0;
  const double ps_pose0 = prev->pose[0], ps_pose1 = prev->pose[1], ps_pose2 = prev->pose[2];
  const int ps_iter = prev->iter, ps_done = prev->done, ps_have = prev->have_partials;
  const SolveParams prm = st->prm;
  const int armed = call->n;
  const int fixed_iterations = call->fixed_iterations;
  IterState* const host_state = call->host_state;
  int* const host_flag = call->host_flag;
  const int launch = dyn->launch[parity ^ 1] + 1;
  const int starts_done = dyn->starts_done;           // as of the previous launches
  float4 pv[3];
  {
    const float* part = &dyn->partials[parity ^ 1][h][0][0];
#pragma unroll
    for (int v = 0; v < 3; ++v)
      pv[v] = *reinterpret_cast<const float4*>(part + (wave * 3 + v) * kMaxBlocks + lane * 4);
  }
  if (herald) {
    dyn->launch[parity] = launch;
    if (host_flag) {
      if (starts_done != m) {                           // progress, while anything is still running
        __hip_atomic_store(host_flag + 1, launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else if (armed != 0) {                          // every start had finished before this launch: end of the call
        const_cast<AlignCall*>(call)->n = 0;            // the launches behind load no points
        __threadfence_system();
        __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {                                          // ... and that launch is complete: the sources are free
        __hip_atomic_store(host_flag + 2, call->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  AlignDynMulti::BodyPose* bp = &dyn->posef[parity][h];
  if (ps_done) {                                        // uniform: a finished start carries its state
    if (writer) { copy_state(cur, prev, -1); bp->done = 1; }
    return;
  }
  double pose[3] = {ps_pose0, ps_pose1, ps_pose2};
  int done = 0;
  if (ps_have) {
    // k_iterate's prologue: wave w owns rows 3w .. 3w+2
    double* t = s_t[wave];
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
    a += dpp_mov<0x128, 0xf>(a);
    if ((lane & 15) == 0 && lane < 48) s_red[wave * 3 + (lane >> 4)] = a;
    __syncthreads();
    if (wave == 0) {
      double H[6], g[3];
#pragma unroll
      for (int j = 0; j < 6; ++j) H[j] = s_red[j];
#pragma unroll
      for (int j = 0; j < 3; ++j) g[j] = s_red[6 + j];
      const double score = s_red[9];
      const int n_hit = (int)(s_red[10] + 0.5);
      int iter = ps_iter, status = 0;
      done = gn_update(pose, H, g, n_hit, iter, status, prm, fixed_iterations, score, &dyn->ls[parity ^ 1][h], &dyn->ls[parity][h],
                       writer) ? 1 : 0;
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
        o.done = done;
        o.have_partials = 1;
        o.pad = launch;
        *cur = o;
        if (done) {
          if (host_flag) { host_state[h] = o; __threadfence_system(); }
          atomicAdd(&dyn->starts_done, 1);
        }
      }
    }
  } else if (writer) {
    copy_state(cur, prev, 1);
  }
  if (writer) {
    double sn_d, cs_d;
    sincos_wrapped(pose[2], &sn_d, &cs_d);
    bp->cs = (float)cs_d; bp->sn = (float)sn_d; bp->tx = (float)pose[0]; bp->ty = (float)pose[1];
    bp->done = done;
  }
}

template <int MODE, int NH, int THREADS, bool SHARED, int NG = 1>
__global__ __launch_bounds__(THREADS) void k_multi_body(const AlignStatic* __restrict__ st, const AlignCall* __restrict__ call,
                                                        AlignDynMulti* __restrict__ dyn, int parity) {
  constexpr int kWaves = THREADS / 64;
  __shared__ float s_wave[kWaves][NH][kNumAcc];
  __shared__ float s_t[kWaves][(kNumAcc - 1) * kSumRowStride];
  __shared__ int s_done[NH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hb = blockIdx.y * NH;
  const SolveParams prm = st->prm;
  const GridDev G = st->grid;
  const int armed = call->n;
  const float* __restrict__ sxs[NH];
  const float* __restrict__ sys[NH];
  int ns[NH], done[NH];
  float pcs[NH], psn[NH], ptx[NH], pty[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    sxs[h] = SHARED ? call->sx : dyn->sx[hb + h];
    sys[h] = SHARED ? call->sy : dyn->sy[hb + h];
    ns[h] = armed ? (SHARED ? armed : dyn->n[hb + h]) : 0;
    const AlignDynMulti::BodyPose* bp = &dyn->posef[parity][hb + h];
    pcs[h] = bp->cs; psn[h] = bp->sn; ptx[h] = bp->tx; pty[h] = bp->ty; done[h] = bp->done;
  }
  const int n = ns[0];
  const float* __restrict__ sx = sxs[0];
  const float* __restrict__ sy = sys[0];
  const int stride = kMaxBlocks * THREADS;
  int i = blockIdx.x * THREADS + tid;
  float x = 0.f, y = 0.f, x1 = 0.f, y1 = 0.f;
  if (i < n) { x = sx[i]; y = sy[i]; }
  if (i + stride < n) { x1 = sx[i + stride]; y1 = sy[i + stride]; }
  bool all_done = true;
#pragma unroll
  for (int h = 0; h < NH; ++h) all_done = all_done && done[h];
  if (all_done) return;                                       // uniform
  if (tid < NH) s_done[tid] = dyn->posef[parity][hb + tid].done;   // for the runtime-indexed tail (read after a barrier)

  const float4* __restrict__ rec = G.rec;
  PoseF P[NH];
  Acc2D A[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    P[h] = make_pose(pcs[h], psn[h], ptx[h], pty[h], G.ox, G.oy, G.inv_c, G.W, G.H, prm.d1, prm.d2);
    acc_zero(A[h]);
  }
  if constexpr (SHARED) {
    while (i < n) {
      const int i2 = i + 2 * stride;
      float xn0 = 0.f, yn0 = 0.f, xn1 = 0.f, yn1 = 0.f;
      if (i2 < n) { xn0 = sx[i2]; yn0 = sy[i2]; }
      if (i2 + stride < n) { xn1 = sx[i2 + stride]; yn1 = sy[i2 + stride]; }
      const bool two = (i + stride) < n;
      if constexpr (NG == 1) {
        PointRec r0[NH], r1[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          if (!done[h]) {                                     // uniform
            lookup_point(P[h], rec, x, y, true, r0[h]);
            lookup_point(P[h], rec, x1, y1, two, r1[h]);
          }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          if (!done[h]) {
            accumulate_point<MODE>(P[h], r0[h], A[h]);
            accumulate_point<MODE>(P[h], r1[h], A[h]);
          }
        }
      } else {
#pragma unroll
        for (int h = 0; h < NH; ++h)
          if (!done[h]) score_two_overlapped<MODE>(P[h], G, rec, x, y, x1, y1, two, A[h]);       // uniform
      }
      x = xn0; y = yn0; x1 = xn1; y1 = yn1; i = i2;
    }
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (!done[h]) {                                         // uniform
        const float* __restrict__ px = sxs[h];
        const float* __restrict__ py = sys[h];
        const int nn = ns[h];
        int ii = blockIdx.x * THREADS + tid;
        float u = x, v = y, u1 = x1, v1 = y1;
        if (h > 0) {
          u = v = u1 = v1 = 0.f;
          if (ii < nn) { u = px[ii]; v = py[ii]; }
          if (ii + stride < nn) { u1 = px[ii + stride]; v1 = py[ii + stride]; }
        }
        while (ii < nn) {
          const int i2 = ii + 2 * stride;
          float xn0 = 0.f, yn0 = 0.f, xn1 = 0.f, yn1 = 0.f;
          if (i2 < nn) { xn0 = px[i2]; yn0 = py[i2]; }
          if (i2 + stride < nn) { xn1 = px[i2 + stride]; yn1 = py[i2 + stride]; }
          const bool two = (ii + stride) < nn;
          if constexpr (NG == 1) {
            PointRec r0, r1;
            lookup_point(P[h], rec, u, v, true, r0);
            lookup_point(P[h], rec, u1, v1, two, r1);
            accumulate_point<MODE>(P[h], r0, A[h]);
            accumulate_point<MODE>(P[h], r1, A[h]);
          } else {
            score_two_overlapped<MODE>(P[h], G, rec, u, v, u1, v1, two, A[h]);
          }
          u = xn0; v = yn0; u1 = xn1; v1 = yn1; ii = i2;
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    if (!done[h]) {
      float acc[kNumAcc];
      acc_store(A[h], prm.d2, acc);
      acc[11] = 0.f;
      const float r = wave_reduce11_lds(acc, s_t[wave], lane);
      if ((lane & 3) == 0 && lane < 4 * (kNumAcc - 1)) s_wave[wave][h][lane >> 2] = r;
    }
  }
  __syncthreads();
  for (int k = tid; k < NH * kNumAcc; k += THREADS) {
    const int h = k / kNumAcc, j = k - h * kNumAcc;
    if (!s_done[h]) {
      float r = 0.f;
      if (j < kNumAcc - 1) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) r += s_wave[w][h][j];        // fixed order
      }
      dyn->partials[parity][hb + h][j][blockIdx.x] = r;
    }
  }
}

}  // namespace ndt
