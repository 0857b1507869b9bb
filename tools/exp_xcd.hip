// Scratch: can one XCD (32 CUs sharing an L2) run the Gauss-Newton loop as a persistent kernel?
//   1. which XCC a workgroup lands on (round-robin by blockIdx?)
//   2. cost of a barrier among the 32 workgroups of one XCD through L2 atomics
//   3. barrier + exchange of 12 partial sums per workgroup (what one iteration needs)
// Every spin is bounded: a workgroup that waits too long raises `abort` and everybody leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ void k_where(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

struct Sync {
  unsigned int count;      // arrivals, monotonically increasing
  unsigned int abort;
  unsigned int pad[14];
};

// returns false if the wait timed out (abort raised)
__device__ __forceinline__ bool xcd_barrier(Sync* s, unsigned int target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this workgroup's stores have reached L2
    __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > 2000000 || __hip_atomic_load(&s->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(&s->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
  }
  __shared__ int s_ok;
  if (threadIdx.x == 0) s_ok = ok ? 1 : 0;
  __syncthreads();
  return s_ok != 0;
}

// participants: blocks with blockIdx % stride == 0
__global__ void k_barrier_loop(Sync* s, int stride, int members, int K, float* partials, int exchange, float* sink) {
  if (blockIdx.x % stride) return;
  const int me = blockIdx.x / stride;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    if (exchange) {
      if (threadIdx.x < 12)
        __hip_atomic_store(&partials[((k & 1) * 64 + me) * 16 + threadIdx.x], (float)(k + me + threadIdx.x), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!xcd_barrier(s, (unsigned)members * (k + 1))) return;
    if (exchange) {
      // every workgroup reads all members' partial rows (agent-scope loads: L2, not this CU's L1)
      if (threadIdx.x < 64) {
        float v = 0.f;
        for (int j = 0; j < 12; ++j)
          if (threadIdx.x < members)
            v += __hip_atomic_load(&partials[((k & 1) * 64 + threadIdx.x) * 16 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
        acc += v;
      }
    }
  }
  if (threadIdx.x == 0) sink[me] = acc;
}

int main() {
  int* d_where; CK(hipMalloc(&d_where, 4096 * 4));
  std::vector<int> h(4096);
  for (int grid : {256, 512}) {
    hipLaunchKernelGGL(k_where, dim3(grid), dim3(1024), 0, 0, d_where);
    CK(hipMemcpy(h.data(), d_where, grid * 4, hipMemcpyDeviceToHost));
    printf("grid %d: xcc of blocks 0..15:", grid);
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    int bad = 0;
    for (int i = 0; i < grid; ++i) bad += (h[i] != h[i % 8]);
    printf("   blocks not following blockIdx%%8: %d\n", bad);
  }
  Sync* s; CK(hipMalloc(&s, sizeof(Sync)));
  float *partials, *sink; CK(hipMalloc(&partials, 2 * 64 * 16 * 4)); CK(hipMalloc(&sink, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int K = 2000;
  struct Case { int grid, stride, members, threads; const char* name; };
  const Case cases[] = {{256, 8, 32, 1024, "32 WGs on one XCD      "}, {256, 8, 32, 256, "32 WGs on one XCD, 256t"},
                        {32, 1, 32, 1024, "32 WGs over 8 XCDs     "}, {256, 1, 256, 1024, "256 WGs over 8 XCDs    "}};
  for (const Case& c : cases)
    for (int exchange = 0; exchange < 2; ++exchange) {
      CK(hipMemset(s, 0, sizeof(Sync)));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_barrier_loop, dim3(c.grid), dim3(c.threads), 0, 0, s, c.stride, c.members, K, partials, exchange, sink);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      Sync hs; CK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost));
      printf("%s exchange=%d: %.3f us per barrier%s\n", c.name, exchange, 1e3f * ms / K, hs.abort ? "  (ABORTED: timed out)" : "");
    }
  return 0;
}
