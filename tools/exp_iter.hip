// Ablation timing of k_iterate (scratch tool, not part of the library or the bench):
// builds a grid from a synthetic wall scene with the product kernels, then times graph
// chains of 31 launches for each EXP mask.  hipcc --offload-arch=gfx950 -O3 -o exp_iter exp_iter.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../gtsam_ndt_amd/csrc/ndt2d_kernels.hpp"
using namespace ndt;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#ifndef THR
#define THR 256
#endif
template <int EXP>
float time_chain(hipStream_t st, AlignStatic* d_st, AlignCall* d_call, AlignDyn* d_dyn, const float* sx, const float* sy, int n, int K, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  for (int k = 0; k <= K; ++k) hipLaunchKernelGGL((k_iterate<0, EXP, THR>), dim3(kMaxBlocks), dim3(THR), 0, st, d_st, d_call, d_dyn, k & 1);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) {
    hipLaunchKernelGGL(k_begin, dim3(1), dim3(64), 0, st, d_call, d_dyn, sx, sy, n, 0.1, -0.08, 0.01, K, (IterState*)nullptr, (int*)nullptr, 0);
    CK(hipGraphLaunch(ge, st));
  }
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(k_begin, dim3(1), dim3(64), 0, st, d_call, d_dyn, sx, sy, n, 0.1, -0.08, 0.01, K, (IterState*)nullptr, (int*)nullptr, 0);
    CK(hipGraphLaunch(ge, st));
  }
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return 1e3f * ms / (reps * (K + 1));
}

int main(int argc, char** argv) {
  const int n_t = 1000000, n_s = argc > 1 ? atoi(argv[1]) : 100000;
  std::vector<float> tx(n_t), ty(n_t), sx(n_s), sy(n_s);
  unsigned long long z = 88172645463325252ull;
  auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (double)(z >> 11) / 9007199254740992.0; };
  // walls of a 4x4 block of 50 m rooms + noise
  for (int i = 0; i < n_t; ++i) {
    const int w = (int)(rnd() * 10); const double t = rnd() * 200 - 100, e = (rnd() + rnd() + rnd() + rnd() - 2) * 0.05;
    if (w < 5) { tx[i] = (float)t; ty[i] = (float)(-100 + 50 * w + e); } else { ty[i] = (float)t; tx[i] = (float)(-100 + 50 * (w - 5) + e); }
  }
  for (int i = 0; i < n_s; ++i) {
    const int w = (int)(rnd() * 4); const double t = rnd() * 50, e = (rnd() + rnd() + rnd() + rnd() - 2) * 0.05;
    double X, Y;
    if (w == 0) { X = t; Y = -50 + e; } else if (w == 1) { X = t; Y = 0 + e; } else if (w == 2) { X = 0 + e; Y = t - 50; } else { X = 50 + e; Y = t - 50; }
    sx[i] = (float)(X - 0.1); sy[i] = (float)(Y + 0.08);
  }
  float *dtx, *dty, *dsx, *dsy;
  CK(hipMalloc(&dtx, n_t * 4)); CK(hipMalloc(&dty, n_t * 4)); CK(hipMalloc(&dsx, n_s * 4)); CK(hipMalloc(&dsy, n_s * 4));
  CK(hipMemcpy(dtx, tx.data(), n_t * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dty, ty.data(), n_t * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsx, sx.data(), n_s * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsy, sy.data(), n_s * 4, hipMemcpyHostToDevice));
  GridDev g{};
  g.cell = 0.5; g.cell32 = 0.5f; g.inv_c = 2.f; g.ox = -101.f; g.oy = -101.f; g.W = 404; g.H = 404;
  g.ngrid = 1; g.gx[0] = g.ox; g.gy[0] = g.oy;
  g.fix_scale = std::ldexp(1.0, kFixShift) / 0.5;
  const size_t nc = (size_t)g.W * g.H;
  CK(hipMalloc(&g.rec, 2 * nc * sizeof(float4))); CK(hipMalloc(&g.acc, nc * sizeof(CellAcc)));
  CK(hipMemset(g.acc, 0, nc * sizeof(CellAcc)));
  int* d_cnt; CK(hipMalloc(&d_cnt, 8)); CK(hipMemset(d_cnt, 0, 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipLaunchKernelGGL(k_accumulate, dim3(2048), dim3(kBlock), 0, st, dtx, dty, (size_t)n_t, g, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(k_finalise, dim3((unsigned)((nc + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, g, 3, 1e-3, d_cnt);
  int cnt[2]; CK(hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
  printf("valid cells %d\n", cnt[0]);
  AlignStatic* d_st; AlignCall* d_call; AlignDyn* d_dyn;
  CK(hipMalloc(&d_st, sizeof(AlignStatic))); CK(hipMalloc(&d_call, sizeof(AlignCall))); CK(hipMalloc(&d_dyn, sizeof(AlignDyn)));
  CK(hipMemset(d_dyn, 0, sizeof(AlignDyn)));
  AlignStatic hs{};
  hs.grid = g;
  hs.prm.d1 = 1.f; hs.prm.d2 = 1.f; hs.prm.max_iterations = 100; hs.prm.min_hits = 3;
  hs.prm.eps_trans = 1e-5; hs.prm.eps_rot = 1e-5; hs.prm.step_max_trans = 0.5; hs.prm.step_max_rot = 0.2;
  CK(hipMemcpy(d_st, &hs, sizeof(hs), hipMemcpyHostToDevice));
  const int K = 30, reps = 200;
  for (int round = 0; round < 2; ++round) {
    printf("thr %d n_src %d  us/launch: full %.3f | no-solve %.3f | no-body %.3f | no-epilogue %.3f | prologue only %.3f | body only %.3f | empty %.3f\n", THR, n_s,
           time_chain<0>(st, d_st, d_call, d_dyn, dsx, dsy, n_s, K, reps), time_chain<1>(st, d_st, d_call, d_dyn, dsx, dsy, n_s, K, reps),
           time_chain<2>(st, d_st, d_call, d_dyn, dsx, dsy, n_s, K, reps), time_chain<4>(st, d_st, d_call, d_dyn, dsx, dsy, n_s, K, reps),
           time_chain<6>(st, d_st, d_call, d_dyn, dsx, dsy, n_s, K, reps), time_chain<5>(st, d_st, d_call, d_dyn, dsx, dsy, n_s, K, reps),
           time_chain<8>(st, d_st, d_call, d_dyn, dsx, dsy, n_s, K, reps));
  }
  IterState s; CK(hipMemcpy(&s, &d_dyn->state[K & 1], sizeof(s), hipMemcpyDeviceToHost));
  printf("state: pose %.6f %.6f %.6f iter %d nhit %d\n", s.pose[0], s.pose[1], s.pose[2], s.iter, s.n_hit);
  return 0;
}
