// Launch-boundary floor of dependent kernel chains for several grid shapes (scratch tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_empty(float* p, int k) { if (p == nullptr && k == 12345) p[0] = 1.f; }
__global__ void k_touch(float* p, int k) {   // one dependent read+write per block through HBM
  if (threadIdx.x == 0) p[blockIdx.x * 32 + ((k & 1) * 16)] = p[((blockIdx.x + 1) % gridDim.x) * 32 + (((k + 1) & 1) * 16)] + 1.f;
}
template <typename F> float chain(hipStream_t st, int K, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  for (int k = 0; k < K; ++k) launch(k);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return 1e3f * ms / (reps * K);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
  const int shapes[][2] = {{1, 64}, {8, 256}, {32, 256}, {64, 1024}, {128, 512}, {256, 256}, {256, 512}, {256, 1024}, {512, 256}, {1024, 256}};
  for (auto& s : shapes) {
    const int b = s[0], t = s[1];
    const float a = chain(st, 62, 100, [&](int k) { hipLaunchKernelGGL(k_empty, dim3(b), dim3(t), 0, st, p, k); });
    const float c = chain(st, 62, 100, [&](int k) { hipLaunchKernelGGL(k_touch, dim3(b), dim3(t), 0, st, p, k); });
    printf("grid %4d x %4d : empty %.3f us/launch   touch %.3f us/launch\n", b, t, a, c);
  }
  return 0;
}
