// Many 3D alignments against ONE cached voxel grid per launch chain (the 3D twin of the split chain of
// ndt2d_multi_start.hpp): m <= 64 scans - different ones (several robots', or several recent, scans
// relocalised in one map) or the same one from m initial poses - each with its own Gauss-Newton / Newton loop.
//
// A single 3D alignment is bound by the launch boundary and the reduce + 6x6-solve prologue of k_iterate3
// (6 us per iteration for 6.8 MB of algorithmic traffic: 14 % of the HBM roofline).  Here an iteration of all
// starts is two launches: k_multi_solve3 (one workgroup per start: k_iterate3's prologue, once) and
// k_multi_body3 (grid 256 x m: the evaluation of start blockIdx.y on block blockIdx.x's points,
// evaluate_block3 - the function k_iterate3 runs).
//
// Contract: start k's result is what ndt3d_align_dev returns for scan k and init_poses[k], bit for bit: same
// thread -> point assignment, same per-thread accumulation order, same reduction trees, same update; a start
// that has finished is frozen while the others go on.
// Launch pair k (p = k & 1): solve reads state[p^1], partials[p^1], writes state[p], body[p]; body reads
// body[p], writes partials[p].
#pragma once
#include "ndt3d_kernels.hpp"

namespace ndt {

constexpr int kMaxStarts3 = 64;

struct AlignDynMulti3 {
  IterState3 state[2][kMaxStarts3];
  float partials[2][kMaxStarts3][kNumAcc3][kMaxBlocks];
  LineSearch3 ls[2][kMaxStarts3];
  int launch[2];          // ping-pong launch counter (what the host sees as progress)
  int starts_done;
  int pad;
  const float* sx[kMaxStarts3];
  const float* sy[kMaxStarts3];
  const float* sz[kMaxStarts3];
  int n[kMaxStarts3];
  struct Body { double pose[6]; int done; int pad; } body[2][kMaxStarts3];   // what each start's evaluation uses
};

struct StartPoses3 { double p[kMaxStarts3][6]; };                 // 3 KB of kernel arguments
struct StartScans3 {                                               // 1.8 KB
  const float* sx[kMaxStarts3];
  const float* sy[kMaxStarts3];
  const float* sz[kMaxStarts3];
  int n[kMaxStarts3];
};

// Per-call part of the context, written from kernel arguments in two launches (each argument block stays
// below the 4 KB limit); slots >= m never run.
__global__ void k_begin_multi3_scans(AlignDynMulti3* __restrict__ dyn, StartScans3 scans, int m) {
  const int h = threadIdx.x;
  if (blockIdx.x != 0 || h >= kMaxStarts3) return;
  dyn->sx[h] = scans.sx[h]; dyn->sy[h] = scans.sy[h]; dyn->sz[h] = scans.sz[h];
  dyn->n[h] = h < m ? scans.n[h] : 0;
}
__global__ void k_begin_multi3(AlignCall3* __restrict__ call, AlignDynMulti3* __restrict__ dyn, StartPoses3 poses, int m,
                               int fixed_iterations, IterState3* host_state, int* host_flag, int seq) {
  const int h = threadIdx.x;
  if (blockIdx.x != 0 || h >= kMaxStarts3) return;
  dyn->body[0][h].done = dyn->body[1][h].done = h < m ? 0 : 1;
  if (h == 0) {
    call->seq = seq;
    call->pad = m;
    call->sx = nullptr; call->sy = nullptr; call->sz = nullptr;
    call->n = 1;                                  // the "armed" word of the chain: 0 once the call is over
    call->fixed_iterations = fixed_iterations;
    call->host_state = host_state;
    call->host_flag = host_flag;
    dyn->launch[0] = 0; dyn->launch[1] = 0;
    dyn->starts_done = 0;
  }
  IterState3 s = {};
  if (h < m) {
#pragma unroll
    for (int j = 0; j < 3; ++j) s.pose[j] = poses.p[h][j];
#pragma unroll
    for (int j = 3; j < 6; ++j) s.pose[j] = wrap_angle(poses.p[h][j]);
  } else {
    s.done = 1;                                   // unused slot: never evaluated
  }
  dyn->state[1][h] = s;                           // launch 0 has parity 0 and reads slot 1
  dyn->state[0][h] = IterState3{};
  dyn->ls[0][h] = LineSearch3{};
  dyn->ls[1][h] = LineSearch3{};
}

// One workgroup per start: k_iterate3's prologue (same loads, same reduction order, same update).
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_multi_solve3(const AlignStatic3* __restrict__ st, const AlignCall3* __restrict__ call,
                                                          AlignDynMulti3* __restrict__ dyn, int parity) {
  constexpr int RPW = Acc3<MODE>::kRowsPerWave;
  __shared__ double s_red[kNumAcc3];
  __shared__ double s_t[kBlock / 64][RPW * kSum3RowStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, m = call->pad;
  const IterState3* prev = &dyn->state[parity ^ 1][h];
  IterState3* cur = &dyn->state[parity][h];
  const bool writer = tid == 0;
  const bool herald = writer && h == 0;
  double pose[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) pose[j] = prev->pose[j];
  const int ps_iter = prev->iter, ps_done = prev->done, ps_have = prev->have_partials;
  const SolveParams prm = st->prm;
  const int armed = call->n;
  const int fixed_iterations = call->fixed_iterations;
  IterState3* const host_state = call->host_state;
  int* const host_flag = call->host_flag;
  const int launch = dyn->launch[parity ^ 1] + 1;
  const int starts_done = dyn->starts_done;             // as of the previous launches
  float4 pv[RPW];
  {
    const float* part = &dyn->partials[parity ^ 1][h][0][0];
#pragma unroll
    for (int v = 0; v < RPW; ++v)
      pv[v] = *reinterpret_cast<const float4*>(part + (wave * RPW + v) * kMaxBlocks + lane * 4);
  }
  if (herald) {
    dyn->launch[parity] = launch;
    if (host_flag) {
      if (starts_done != m) {                           // progress, while anything is still running
        __hip_atomic_store(host_flag + 1, launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else if (armed != 0) {                          // every start had finished before this launch: end of the call
        const_cast<AlignCall3*>(call)->n = 0;           // the launches behind load no points
        __threadfence_system();
        __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {                                          // ... and that launch is complete: the sources are free
        __hip_atomic_store(host_flag + 2, call->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  AlignDynMulti3::Body* bp = &dyn->body[parity][h];
  if (ps_done) {                                        // uniform: a finished start carries its state
    if (writer) { copy_state3(cur, prev, -1); bp->done = 1; }
    return;
  }
  int done = 0;
  if (ps_have) {
    {
      double* t = s_t[wave];
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
        a += dpp_mov<0x124, 0xf>(a);
        if ((lane & 7) == 4 && v < RPW) s_red[wave * RPW + v] = a;
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (wave == 0) {
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
      int iter = ps_iter, status = 0;
      done = gn_update3(pose, A, g, n_hit, iter, status, prm, fixed_iterations, score, &dyn->ls[parity ^ 1][h], &dyn->ls[parity][h],
                        writer) ? 1 : 0;
      if (writer) {
        auto store = [&](IterState3* o) {
#pragma unroll
          for (int j = 0; j < 6; ++j) { o->pose[j] = pose[j]; o->g[j] = g[j]; }
#pragma unroll
          for (int j = 0; j < 21; ++j) o->H[j] = s_red[j];
          if (MODE == 1) {
            o->H[15] = A[21]; o->H[16] = A[22]; o->H[17] = A[23]; o->H[18] = A[28]; o->H[19] = A[29]; o->H[20] = A[35];
          }
          o->score = score;
          o->n_hit = n_hit; o->iter = iter; o->status = status;
          o->done = done; o->have_partials = 1; o->pad = launch;
        };
        store(cur);
        if (done) {
          if (host_flag) { store(&host_state[h]); __threadfence_system(); }
          atomicAdd(&dyn->starts_done, 1);
        }
      }
    }
  } else if (writer) {
    copy_state3(cur, prev, 1);
  }
  if (writer) {
#pragma unroll
    for (int j = 0; j < 6; ++j) bp->pose[j] = pose[j];
    bp->done = done;
  }
}

// grid (256, m): the evaluation of start blockIdx.y on block blockIdx.x's share of its scan
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_multi_body3(const AlignStatic3* __restrict__ st, const AlignCall3* __restrict__ call,
                                                         AlignDynMulti3* __restrict__ dyn, int parity) {
  constexpr int NA = Acc3<MODE>::kUsed;
  __shared__ float s_wave[kBlock / 64][kNumAcc3];
  __shared__ float s_t[kBlock / 64][NA * kSum3RowStride];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int h = blockIdx.y;
  const SolveParams prm = st->prm;
  const Grid3Dev G = st->grid;
  const int armed = call->n;
  const AlignDynMulti3::Body* bp = &dyn->body[parity][h];
  const int done = bp->done;
  double pose[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) pose[j] = bp->pose[j];
  const float* __restrict__ sx = dyn->sx[h];
  const float* __restrict__ sy = dyn->sy[h];
  const float* __restrict__ sz = dyn->sz[h];
  const int n = armed ? dyn->n[h] : 0;
  const int i = blockIdx.x * kBlock + tid;
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < n) { x = sx[i]; y = sy[i]; z = sz[i]; }
  if (done) return;                                     // uniform
  evaluate_block3<MODE>(G, prm, pose, sx, sy, sz, n, i, x, y, z, s_wave, s_t[wave], &dyn->partials[parity][h][0][blockIdx.x]);
}

}  // namespace ndt
