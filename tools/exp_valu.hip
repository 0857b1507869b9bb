// VALU issue-rate microbenchmark (scratch): cycles per wave64 instruction per SIMD for an
// independent f32 fma stream at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int MIX>
__global__ void k_fma(float* out, int iters, float a, float b) {
  float r[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) r[j] = threadIdx.x * 0.001f + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MIX == 0) r[j] = fmaf(r[j], a, b);
      if (MIX == 1) r[j] = (r[j] > b) ? r[j] * a : r[j] + b;          // cmp + cndmask + mul + add
      if (MIX == 2) r[j] = fmaf(r[j], a, r[(j + 1) & 15]);            // cross-register dependence
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += r[j];
  if (s == 12345.678f) out[0] = s;
}
template <int MIX> void run(const char* name, int threads, float* out) {
  const int iters = 4096;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_fma<MIX>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_fma<MIX>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double winstr_per_simd = (double)iters * 16 * (threads / 64) / 4.0;   // wave-instructions per SIMD (source-level ops)
  printf("%-10s %4d threads/CU (%d waves/SIMD): %.3f ms, %.2f ns per source op per SIMD\n", name, threads, threads / 256, ms, ms * 1e6 / winstr_per_simd);
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  for (int t : {256, 512, 1024}) { run<0>("fma", t, out); run<1>("cmp/sel", t, out); run<2>("fma-dep", t, out); }
  return 0;
}
