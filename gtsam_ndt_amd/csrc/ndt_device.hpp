// Device-side building blocks shared by the 2D/3D NDT kernels (gfx950 only).
// wave64 DPP reductions, 128-bit integer helper, ordered-float atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ndt {

// ---------------------------------------------------------------- wave64 reductions
// Fixed-tree DPP sum over the 64 lanes of a wave (no LDS, no bpermute).  After the six
// steps lane 63 holds the total; the tree is the same on every launch, so float sums are
// bitwise reproducible (SURVEY.md section 4 "Determinism").
//   quad_perm [1,0,3,2] (0xB1), quad_perm [2,3,0,1] (0x4E), row_ror:4 (0x124),
//   row_ror:8 (0x128), row_bcast:15 rows 1,3 (0x142), row_bcast:31 rows 2,3 (0x143)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the wave; result valid in lane 63 only.
template <typename T>
__device__ __forceinline__ T wave_sum_lane63(T v) {
  v += dpp_mov<0xB1, 0xf>(v);
  v += dpp_mov<0x4E, 0xf>(v);
  v += dpp_mov<0x124, 0xf>(v);
  v += dpp_mov<0x128, 0xf>(v);
  v += dpp_mov<0x142, 0xa>(v);
  v += dpp_mov<0x143, 0xc>(v);
  return v;
}

__device__ __forceinline__ float read_lane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double read_lane63(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// Sum over the wave, broadcast to every lane (wave-uniform value).
template <typename T>
__device__ __forceinline__ T wave_sum(T v) { return read_lane63(wave_sum_lane63(v)); }

// exact integer sum over the wave, in every lane (cold paths)
__device__ __forceinline__ unsigned int wave_sum_u32(unsigned int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += (unsigned int)__shfl_xor((int)v, m, 64);
  return v;
}

// min / max over the wave via xor shuffles (cold path: bounds kernel only)
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// ---------------------------------------------------------------- sharded event counters
// A build's "valid cells / overflowed cells" counts are summed with global atomics.  Adds on ONE word serialise at
// the memory side (about 10 ns each, after the adding waves have long ended): the 2 700 per-wave adds of a 1M-point
// tile build were 22 us of a 39 us kernel.  The counts live in kCountShards pairs instead - a workgroup adds to pair
// blockIdx.x % kCountShards - and the host adds the pairs up.
constexpr int kCountShards = 16;
constexpr int kCountInts = 2 * kCountShards;
__device__ __forceinline__ int* count_shard(int* counters) { return counters + 2 * (blockIdx.x & (kCountShards - 1)); }
// one add per WORKGROUP: every thread of the workgroup must call it (it has barriers); a, b = this thread's counts
__device__ __forceinline__ void block_count_add(int* counters, int a, int b) {
  __shared__ int s_cnt[2];
  if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
  __syncthreads();
  const unsigned long long ma = __ballot(a != 0), mb = __ballot(b != 0);
  if (ma | mb) {                          // (rare enough per wave that the LDS adds of single lanes are cheap)
    if (a) atomicAdd(&s_cnt[0], a);
    if (b) atomicAdd(&s_cnt[1], b);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int* c = count_shard(counters);
    if (s_cnt[0]) atomicAdd(&c[0], s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&c[1], s_cnt[1]);
  }
}

// ---------------------------------------------------------------- ordered float <-> uint
__device__ __forceinline__ unsigned int float_to_ordered(float f) {
  const unsigned int u = __builtin_bit_cast(unsigned int, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ordered_to_float(unsigned int u) {
  const unsigned int b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  __builtin_memcpy(&f, &b, 4);
  return f;
}

// ---------------------------------------------------------------- signed 128-bit helper
// Enough of __int128 for n*Sxx - Sx*Sy on exact fixed-point cell sums (device code has
// no compiler-rt for __int128 -> double).
struct i128 {
  long long hi;
  unsigned long long lo;
};
__device__ __forceinline__ i128 mul_s64(long long a, long long b) {
  i128 r;
  r.lo = (unsigned long long)a * (unsigned long long)b;
  r.hi = __mul64hi(a, b);
  return r;
}
__device__ __forceinline__ i128 sub128(i128 a, i128 b) {
  i128 r;
  r.lo = a.lo - b.lo;
  r.hi = a.hi - b.hi - (a.lo < b.lo ? 1 : 0);
  return r;
}
__device__ __forceinline__ double to_double(i128 v) {
  const bool neg = v.hi < 0;
  if (neg) {  // two's complement negate
    v.lo = ~v.lo + 1ull;
    v.hi = ~v.hi + (v.lo == 0ull ? 1 : 0);
  }
  const double d = (double)v.hi * 18446744073709551616.0 + (double)v.lo;
  return neg ? -d : d;
}

}  // namespace ndt
