// Mid-size scans on the single-pair path: the whole Gauss-Newton loop in ONE launch, carried by a team
// of 32 workgroups that synchronise through L2 - no kernel boundary per iteration.
//
// Why: with one launch per iteration (k_iterate) a 100k-point alignment spends 1.7 us of every 4.6 us
// on the kernel boundary, and - because the per-XCD L2s are written back and invalidated there - fetches
// the scan, the cell records and the block partials from the memory side again in every launch (two
// dependent ~0.8 us round trips).  An MI355X is eight XCDs of 32 CUs with one L2 each.  Workgroups are
// dispatched round-robin over the XCDs, so the 32 workgroups with blockIdx % 8 == t normally share one L2:
// a barrier among them costs 0.86 us (tools/exp_xcd.hip) instead of 1.7 us for a kernel boundary, and
// scan, records and partial sums stay L2-resident across iterations.  Team t runs start t of a
// multi-start call (ndt2d_align_multi_start_dev); a single alignment is the one-start case.
//
// Correctness does not depend on the placement: team membership is by blockIdx alone, every value that
// crosses workgroups travels through agent-scope (sc1) stores, loads and atomics, which are coherent
// across XCDs, and the reduction order is fixed, so results are reproducible wherever the workgroups
// land - a team spread over several XCDs is merely slower.  Every spin is bounded: a team whose members
// do not all arrive (the GPU is busy with other work and cannot host 32 workgroups at once) raises its
// abort flag, every member leaves, and the host runs the alignment through k_iterate instead.
//
// Same per-point code and the same update rule as the other kernels (ndt2d_kernels.hpp); the sums are
// taken in this kernel's own order, so a result agrees with k_iterate's up to float32 summation order.
#pragma once
#include "ndt2d_kernels.hpp"
#include "ndt2d_multi_start.hpp"

namespace ndt {

constexpr int kXcdTeams = 8;              // = XCDs of an MI355X; team t = workgroups with blockIdx % 8 == t
constexpr int kXcdMembers = 32;           // = CUs of an XCD
constexpr int kXcdThreads = 1024;
constexpr int kXcdSpinLimit = 400000;     // bounded wait at the team barrier (a few tens of milliseconds)
constexpr int kXcdFirstSpinLimit = 3000;  // the FIRST barrier of a launch waits only this long (tens of microseconds):
                                          // members that are not there by then are not resident (the CUs are busy with
                                          // other work), and spinning for them could keep them from ever becoming so

struct XcdTeam {
  unsigned int count;      // arrivals at the team barrier, monotonically increasing within a launch
  unsigned int abort;      // raised by a member that waited too long: everybody leaves
  unsigned int pad[14];
};
struct XcdShared {
  XcdTeam team[kXcdTeams];
  float partials[kXcdTeams][2][kXcdMembers][16];    // [team][parity][member][11 sums + pad]
};

// One row of the team's table (12 floats, 64-byte aligned) with agent-scope loads, all three requested
// before the one wait: as __hip_atomic_load calls the compiler waits after each of the eleven.
__device__ __forceinline__ void load_row_sc1(const float* p, float4& a, float4& b, float4& c) {
  asm volatile(
      "global_load_dwordx4 %0, %3, off sc1\n\t"
      "global_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %3, off offset:32 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c)
      : "v"(p)
      : "memory");
}

__device__ __forceinline__ bool team_barrier(XcdTeam* t, unsigned int target, int* s_ok, int spin_limit) {
  __syncthreads();
  if (threadIdx.x == 0) {
    bool ok = true;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this workgroup's sc1 stores have left
    __hip_atomic_fetch_add(&t->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(&t->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      // the abort flag is looked at every 32nd poll: it doubles the poll's latency otherwise
      if (++spins > spin_limit ||
          ((spins & 31) == 0 && __hip_atomic_load(&t->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(&t->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
    *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*s_ok) != 0;
}

// host_flag[h]: written by start h's team when it is through: +seq = result in host_state[h],
// -seq = the team gave up (abort): run the call through the launch-per-iteration path.
template <int MODE>
__global__ __launch_bounds__(kXcdThreads) void k_align_xcd(const AlignStatic* __restrict__ st, const float* __restrict__ sx,
                                                           const float* __restrict__ sy, int n, StartPoses poses, int m,
                                                           int fixed_iterations, XcdShared* __restrict__ sh,
                                                           IterState* __restrict__ dev_state,
                                                           IterState* __restrict__ host_state, int* __restrict__ host_flag,
                                                           int seq, int home) {
  constexpr int kWaves = kXcdThreads / 64;
  __shared__ float s_wave[kWaves][kNumAcc];
  __shared__ float s_t[kWaves][(kNumAcc - 1) * kSumRowStride];
  __shared__ double s_bc[16];       // pose(3) | H(6) g(3) score n_hit of the last evaluation
  __shared__ double s_col[kNumAcc][kXcdMembers + 1];
  __shared__ int s_misc[4];         // done, iter, status
  __shared__ int s_ok;
  __shared__ LineSearch s_ls;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // team 0 sits on the handle's home XCD (handles are spread over the XCDs so that concurrent
  // single alignments of different handles do not all ask for the same 32 CUs)
  const int team = (blockIdx.x + kXcdTeams - home) % kXcdTeams, me = blockIdx.x / kXcdTeams;
  if (team >= m) return;                                   // uniform: this team has no start
  const SolveParams prm = st->prm;
  const GridDev G = st->grid;
  const float4* __restrict__ rec = G.rec;
  XcdTeam* T = &sh->team[team];
  unsigned int arrivals = 0;

  for (int h = team; h < m; h += kXcdTeams) {              // uniform: starts team, team + 8
    double pose[3] = {poses.p[h][0], poses.p[h][1], wrap_angle(poses.p[h][2])};
    if (tid == 0) { s_ls.valid = 0; s_ls.trials = 0; s_misc[1] = 0; s_misc[2] = 0; }
    __syncthreads();
    bool aborted = false;
#ifdef NDT_XCD_PROFILE            // tools only: where an iteration's time goes (100 MHz ticks, thread 0 of member 0)
    unsigned long long t_body = 0, t_row = 0, t_bar = 0, t_solve = 0, t_bc = 0, t_mark = wall_clock64();
#define NDT_XTICK(a) do { const unsigned long long t_now = wall_clock64(); a += t_now - t_mark; t_mark = t_now; } while (0)
#else
#define NDT_XTICK(a) do {} while (0)
#endif
    for (int k = 0;; ++k) {
      // ---- body: this workgroup's share of the scan at `pose`
      float acc[kNumAcc];
      {
        double sn_d, cs_d;
        sincos_wrapped(pose[2], &sn_d, &cs_d);
        auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
        const PoseF P = make_pose(uni((float)cs_d), uni((float)sn_d), uni((float)pose[0]), uni((float)pose[1]), G.ox, G.oy,
                                  G.inv_c, G.W, G.H, prm.d1, prm.d2);
        Acc2D A;
        acc_zero(A);
        const int stride = kXcdMembers * kXcdThreads;
        int i = me * kXcdThreads + tid;
        float x = 0.f, y = 0.f, x1 = 0.f, y1 = 0.f;
        if (i < n) { x = sx[i]; y = sy[i]; }
        if (i + stride < n) { x1 = sx[i + stride]; y1 = sy[i + stride]; }
        while (i < n) {
          const int i2 = i + 2 * stride;
          float xn0 = 0.f, yn0 = 0.f, xn1 = 0.f, yn1 = 0.f;
          if (i2 < n) { xn0 = sx[i2]; yn0 = sy[i2]; }
          if (i2 + stride < n) { xn1 = sx[i2 + stride]; yn1 = sy[i2 + stride]; }
          PointRec r0, r1;
          const bool two = (i + stride) < n;
          lookup_point(P, rec, x, y, true, r0);
          lookup_point(P, rec, x1, y1, two, r1);
          accumulate_point<MODE>(P, r0, A);
          accumulate_point<MODE>(P, r1, A);
          x = xn0; y = yn0; x1 = xn1; y1 = yn1; i = i2;
        }
        acc_store(A, prm.d2, acc);
        acc[11] = 0.f;
      }
      NDT_XTICK(t_body);
      // ---- the workgroup's 11 sums -> one row of the team's table (sc1 stores: visible across XCDs)
      {
        const float r = wave_reduce11_lds(acc, s_t[wave], lane);
        if ((lane & 3) == 0 && lane < 4 * (kNumAcc - 1)) s_wave[wave][lane >> 2] = r;
      }
      __syncthreads();
      float* row = &sh->partials[team][k & 1][me][0];
      if (tid < kNumAcc - 1) {
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) r += s_wave[w][tid];           // fixed order
        __hip_atomic_store(&row[tid], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      arrivals += kXcdMembers;
      NDT_XTICK(t_row);
      if (!team_barrier(T, arrivals, &s_ok, arrivals == kXcdMembers ? kXcdFirstSpinLimit : kXcdSpinLimit)) { aborted = true; break; }

      NDT_XTICK(t_bar);
      // ---- every workgroup: all 32 rows -> totals in a fixed order (float64), then the update
      if (wave == 0) {
        const float* tab = &sh->partials[team][k & 1][0][0];
        if (lane < kXcdMembers) {
          float4 a, b, c;
          load_row_sc1(tab + lane * 16, a, b, c);
          s_col[0][lane] = (double)a.x; s_col[1][lane] = (double)a.y; s_col[2][lane] = (double)a.z; s_col[3][lane] = (double)a.w;
          s_col[4][lane] = (double)b.x; s_col[5][lane] = (double)b.y; s_col[6][lane] = (double)b.z; s_col[7][lane] = (double)b.w;
          s_col[8][lane] = (double)c.x; s_col[9][lane] = (double)c.y; s_col[10][lane] = (double)c.z;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < kNumAcc - 1) {
          double tot = 0.0;
#pragma unroll
          for (int w = 0; w < kXcdMembers; ++w) tot += s_col[lane][w];
          s_bc[3 + lane] = tot;
        }
        __builtin_amdgcn_wave_barrier();
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
      NDT_XTICK(t_solve);
      __syncthreads();
      pose[0] = s_bc[0]; pose[1] = s_bc[1]; pose[2] = s_bc[2];
      NDT_XTICK(t_bc);
      if (__builtin_amdgcn_readfirstlane(s_misc[0])) break;
      // the next iteration's rows go to the other parity; a member can be at most one barrier ahead
    }
#ifdef NDT_XCD_PROFILE
    if (me == 0 && tid == 0 && s_misc[1] > 0)
      printf("k_align_xcd n=%d iters=%d: body %.2f row %.2f barrier %.2f exchange+solve %.2f broadcast %.2f us/iter\n", n, s_misc[1],
             0.01 * t_body / s_misc[1], 0.01 * t_row / s_misc[1], 0.01 * t_bar / s_misc[1], 0.01 * t_solve / s_misc[1],
             0.01 * t_bc / s_misc[1]);
#endif
#undef NDT_XTICK
    if (me == 0 && tid == 0) {
      if (aborted) {                                       // this start and the team's later ones: not done here
        for (int hh = h; hh < m; hh += kXcdTeams)
          __hip_atomic_store(host_flag + hh, -seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
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
        dev_state[h] = o;
        host_state[h] = o;
        __threadfence_system();
        __hip_atomic_store(host_flag + h, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (aborted) return;                                   // uniform within the team
    __syncthreads();                                       // s_bc / s_misc are rewritten by the next start
  }
}

}  // namespace ndt
