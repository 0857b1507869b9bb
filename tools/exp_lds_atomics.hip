// Scratch microbenchmark: what an LDS atomic costs on gfx950 by width and address pattern.
//   hipcc --offload-arch=gfx950 -O3 -o exp_lds_atomics tools/exp_lds_atomics.hip && ./exp_lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kSlots = 1056;
template <int WIDTH, int PATTERN, int NATOM>
__global__ __launch_bounds__(1024) void k(const unsigned* __restrict__ idx, int iters, unsigned long long* out) {
  __shared__ unsigned long long s64[NATOM > 5 ? 5 : NATOM][kSlots];
  __shared__ unsigned s32[NATOM][kSlots];
  for (int i = threadIdx.x; i < kSlots; i += blockDim.x) {
    for (int j = 0; j < NATOM; ++j) { s32[j][i] = 0; if (j < 5) s64[j][i] = 0; }
  }
  __syncthreads();
  unsigned c = idx[blockIdx.x * blockDim.x + threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    unsigned slot;
    if (PATTERN == 0) slot = c % kSlots;                             // random slots
    else if (PATTERN == 1) slot = (threadIdx.x + it) % kSlots;        // consecutive lanes, consecutive slots (conflict-free)
    else slot = ((c >> 3) * 33) % kSlots;                             // groups of lanes on the same slot (runs)
#pragma unroll
    for (int j = 0; j < NATOM; ++j) {
      if (WIDTH == 64) atomicAdd(&s64[j < 5 ? j : 0][slot], (unsigned long long)c + j);
      else atomicAdd(&s32[j][slot], c + j);
    }
    c = c * 1664525u + 1013904223u;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s64[0][1] + s32[0][1];
}

template <int WIDTH, int PATTERN, int NATOM>
void run(const char* name, int threads) {
  const int blocks = 256, iters = 2000;
  unsigned* d_idx; unsigned long long* d_out;
  std::vector<unsigned> h(blocks * 1024);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
  hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_out, blocks * 8);
  hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WIDTH, PATTERN, NATOM>), dim3(blocks), dim3(threads), 0, 0, d_idx, 10, d_out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<WIDTH, PATTERN, NATOM>), dim3(blocks), dim3(threads), 0, 0, d_idx, iters, d_out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = (double)iters * NATOM * (threads / 64);       // per CU
  printf("%-44s threads %4d: %7.1f ns per wave-atomic per CU (%.1f cycles at 2.1 GHz)\n", name, threads, ms * 1e6 / wave_instr, ms * 1e6 / wave_instr * 2.1);
  hipFree(d_idx); hipFree(d_out);
}

int main() {
  for (int threads : {256, 1024}) {
    run<32, 0, 5>("u32 x5, random slots", threads);
    run<64, 0, 5>("u64 x5, random slots", threads);
    run<32, 1, 5>("u32 x5, conflict-free", threads);
    run<64, 1, 5>("u64 x5, conflict-free", threads);
    run<32, 2, 5>("u32 x5, 8 lanes per slot", threads);
    run<64, 2, 5>("u64 x5, 8 lanes per slot", threads);
    run<32, 0, 11>("u32 x11, random slots", threads);
  }
  return 0;
}
