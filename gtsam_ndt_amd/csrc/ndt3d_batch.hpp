// 3D loop-closure candidate batch: the 3D twin of ndt2d_batch.hpp.  Persistent 1024-thread workgroups
// (one per CU) pull scan pairs from a queue and run the whole alignment of a pair on chip:
//   - the pair's voxel grid lives in LDS for the Gauss-Newton loop: a dense u16 voxel -> slot table
//     and 36-byte records (mean | Sigma^-1) per occupied voxel.  The carve is per pair: the
//     table takes 2 bytes per voxel, the records get the rest (config 5: 44 x 44 x 9 = 17 424
//     voxels, 2 706 occupied: 35 KB + 97 KB of the 160 KB; up to 3 494 occupied voxels fit that grid);
//   - the nine exact fixed-point sums of a voxel (72 bytes per slot) do not fit next to the table,
//     so the build adds them up in LDS as many at a time as the records' region holds (five for
//     config 5: two passes over the target; four for a grid that fills the carve: three) and parks
//     each pass in a per-workgroup slab of global memory; the finalise reads them back once and
//     writes the records into LDS (a pair with a few more occupied voxels than records fit keeps
//     the rest in the slab; bigger grids go to k_batch3_fallback, every table in global memory);
//   - source points stream from HBM/L2 once per iteration; the 29 (Newton: 38) sums are reduced
//     per wave (DPP) -> LDS -> wave 0, which also does the 6x6 solve: two workgroup barriers per
//     iteration, no kernel boundary.
// Records and update rule are the single-pair 3D path's: finalise_sums3 and gn_update3 are the functions
// k_finalise3 / k_iterate3 use, the slab sums are the integers k_accumulate3 forms.  The per-point sums are
// taken in the map frame (accumulate_point3_map below: the same Hessian and gradient with a quarter fewer
// instructions), so a pair agrees with its single-pair alignment to float32 rounding, not bit for bit.
#pragma once
#include "ndt2d_batch.hpp"
#include "ndt3d_kernels.hpp"

namespace ndt {

struct Result3Dev {   // layout of ndt3d_result (include/ndt_hip.h); static_assert in the API file
  double pose[6];
  double H[36];
  double g[6];
  double score;
  int iterations, n_hit, status, reserved;
};

struct Batch3Args {
  const float* tx; const float* ty; const float* tz; const unsigned long long* toff;   // targets, concatenated SoA
  const float* sx; const float* sy; const float* sz; const unsigned long long* soff;   // sources
  const double* init;        // [n_pairs][6]
  Result3Dev* out;           // [n_pairs]
  unsigned int* queue;       // zeroed before the launch
  unsigned char* slab;       // [gridDim.x][kB3SlabBytes]
  unsigned char* gslab;      // tables of the global-memory variant: [global_blocks][kG3SlabBytes]
  int* fb_marks;             // [n_pairs], zeroed before the launches: k_batch3 sets 1 for a pair over the LDS carve,
                             // k_batch3_fallback processes exactly those (null: such pairs get NDT_ERR_CAPACITY at once)
  unsigned int* fb_seen;     // pinned host word: k_batch3_fallback counts the pairs it processes here (BatchArgs::fb_seen)
  int n_pairs;
  int min_points;
  int fixed_iterations;
  int chain;                 // as BatchArgs::chain (a later level of a coarse-to-fine run)
  double cell;
  double eig_ratio;
  SolveParams prm;
};

constexpr int kB3Threads = 1024;
constexpr int kB3Waves = kB3Threads / 64;
#ifndef NDT_B3_UNROLL
#define NDT_B3_UNROLL 1
#endif
constexpr int kB3Unroll = NDT_B3_UNROLL;                // source points per thread and register set (2 measured equal:
                                                        // the loop is bound by VALU issue, not by the LDS round trips)
constexpr int kB3MaxSlots = 4096;                       // bound of the slab; the LDS carve allows fewer
// LDS carve (bytes).  Fixed part first, then the voxel table (u16 per voxel), then the records.
constexpr int kB3Red = 0;                               // float [Waves][kNumAcc3]
constexpr int kB3Bc = kB3Red + kB3Waves * kNumAcc3 * 4; // double [48]: pose(6) | sums(38)
constexpr int kB3Misc = kB3Bc + 48 * 8;                 // int [16]
constexpr int kB3Scan = kB3Misc + 64;                   // int [16]
constexpr int kB3Ls = kB3Scan + 64;                     // LineSearch3
constexpr int kB3Idx = kB3Ls + 128;                     // u16 [ncell]
constexpr int kB3LdsBytes = 160 * 1024;
constexpr int kB3RecBytes = 36;                         // per slot: float4 (mean, xx) + float4 (xy xz yy yz) + float (zz); a valid
                                                        // record has xx > 0 (Sigma^-1 is positive definite), the dummy record 0 is all zero
static_assert(sizeof(LineSearch3) <= 128 && (kB3Idx % 16) == 0 && (kB3Bc % 16) == 0, "carve");
// slab of one workgroup: u32 slot_n[S], u32 slot_key[S], u64 sums[9][S], overflow records
constexpr int kB3SlabN = 0;
constexpr int kB3SlabKey = kB3SlabN + kB3MaxSlots * 4;
constexpr int kB3SlabSums = kB3SlabKey + kB3MaxSlots * 4;
// records beyond the LDS carve (a pair with a few more occupied voxels than fit keeps the first ones on chip, the
// rest here, gathered through L2): float4 [S], float4 [S], float [S]
constexpr int kB3SlabRecA = kB3SlabSums + 9 * kB3MaxSlots * 8;
constexpr int kB3SlabRecB = kB3SlabRecA + kB3MaxSlots * 16;
constexpr int kB3SlabRecC = kB3SlabRecB + kB3MaxSlots * 16;
constexpr int kB3SlabBytes = kB3SlabRecC + kB3MaxSlots * 4;

// Second variant, for the pairs whose voxel grid does not fit the LDS carve (a scan against a wide or finely
// gridded map): the same code with every table in a per-workgroup slab of global memory - u32 voxel -> slot
// table (also the counts of the build), per-slot counts / keys / nine sums (counted and added up a range at a time in LDS),
// 36-byte records gathered through L2.  It runs on the context's global_blocks workgroups and only on the pairs k_batch3 handed over.
constexpr int kG3MaxCells = 1 << 20;                    // 1 048 576 voxels (e.g. 256 x 256 x 16)
constexpr int kG3MaxSlots = 1 << 15;                    // occupied voxels (slot 0 is the dummy record)
constexpr int kG3BlocksStart = 8;                       // a context starts with 8 workgroups / 7.9 MB slabs of the global-table variant (63 MB)
constexpr int kG3Blocks = 256;                          // ... and grows to one per CU (2.0 GB) after the first call in which a pair needed it;
constexpr int kG3BlocksMax = 256;                       // NDT_TUNE_BATCH_GLOBAL_WORKGROUPS pins the number instead
constexpr size_t kG3Idx = 0;                                                  // u32 [MaxCells]
constexpr size_t kG3SlotN = kG3Idx + (size_t)kG3MaxCells * 4;                 // u32 [MaxSlots]
constexpr size_t kG3SlotKey = kG3SlotN + (size_t)kG3MaxSlots * 4;             // u32 [MaxSlots]
constexpr size_t kG3Sums = kG3SlotKey + (size_t)kG3MaxSlots * 4;              // u64 [9][MaxSlots]
constexpr size_t kG3RecA = kG3Sums + (size_t)9 * kG3MaxSlots * 8;             // float4 [MaxSlots]
constexpr size_t kG3RecB = kG3RecA + (size_t)kG3MaxSlots * 16;                // float4 [MaxSlots]
constexpr size_t kG3RecC = kG3RecB + (size_t)kG3MaxSlots * 16;                // float [MaxSlots] (region sized for 8 B each)
constexpr size_t kG3SlabBytes = kG3RecC + (size_t)kG3MaxSlots * 8;            // 7.9 MB

constexpr int kTgt3Unroll = 4;
template <typename F>
__device__ __forceinline__ void for_each_target_point3(const float* tx, const float* ty, const float* tz, int nt, F&& body) {
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)tx, 0, nt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)ty, 0, nt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)tz, 0, nt * 4, 0x00020000);
  for (int i = threadIdx.x; i < nt; i += kTgt3Unroll * kB3Threads) {
    float x[kTgt3Unroll], y[kTgt3Unroll], z[kTgt3Unroll];
#pragma unroll
    for (int u = 0; u < kTgt3Unroll; ++u) {
      const int off = (i + u * kB3Threads) * 4;
      x[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
      y[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
      z[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, off, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < kTgt3Unroll; ++u)
      if (i + u * kB3Threads < nt) body(x[u], y[u], z[u]);
  }
}

// The build's passes that add into LDS walk the target in ROWS of 64 points: lane l of a wave takes the points
// i = 64 row + l of the wave's contiguous block of rows, one row per step (coalesced: a row is 256 B).  A 64-beam scan
// in firing order (all beams of one azimuth, then the next azimuth) puts one beam in every lane, and consecutive rows of
// a lane are neighbouring bearings of that beam - the same voxel for tens of steps - so a lane adds a run of equal keys
// up in registers and issues its LDS atomics once per run instead of once per point (they are 23 % of a config-5 pair's
// time otherwise).  Any other order of the cloud is as correct (the sums are exact integers: every
// partition gives the same bits) and costs what the plain loop costs.  body(x, y, z) sees the lane's points in order.
template <typename F>
__device__ __forceinline__ void for_each_target_point_rows3(const float* tx, const float* ty, const float* tz, int nt, F&& body) {
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)tx, 0, nt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)ty, 0, nt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)tz, 0, nt * 4, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = (nt + 63) >> 6;
  const int per_wave = (rows + kB3Waves - 1) / kB3Waves;
  const int r0 = wave * per_wave;
  const int r1 = r0 + per_wave < rows ? r0 + per_wave : rows;
  constexpr int U = 4;                                                // rows in flight
  for (int r = r0; r < r1; r += U) {
    float x[U], y[U], z[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int off = (((r + u) << 6) + lane) * 4;                    // beyond the cloud: the buffer returns 0, unused
      x[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
      y[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
      z[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, off, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (r + u < r1 && ((r + u) << 6) + lane < nt) body(x[u], y[u], z[u]);
  }
}

struct Cfg1024 { static constexpr int kWaves = kB3Waves; };   // for block_excl_scan

__device__ __forceinline__ void write_result3(Result3Dev* o, const double* pose, const double* s21, const double* g,
                                              double score, int iter, int n_hit, int status) {
#pragma unroll
  for (int j = 0; j < 6; ++j) { o->pose[j] = pose[j]; o->g[j] = g ? g[j] : 0.0; }
  double H[36];
#pragma unroll
  for (int j = 0; j < 36; ++j) H[j] = 0.0;
  if (s21) {   // Htt(6) Htr(9) Hrr(6), as unpack_h21 on the host
    H[0] = s21[0]; H[1] = s21[1]; H[2] = s21[2]; H[7] = s21[3]; H[8] = s21[4]; H[14] = s21[5];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) H[6 * r + 3 + k] = s21[6 + 3 * r + k];
    H[21] = s21[15]; H[22] = s21[16]; H[23] = s21[17]; H[28] = s21[18]; H[29] = s21[19]; H[35] = s21[20];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < r; ++c) H[6 * r + c] = H[6 * c + r];
  }
#pragma unroll
  for (int j = 0; j < 36; ++j) o->H[j] = H[j];
  o->score = score;
  o->iterations = iter; o->n_hit = n_hit; o->status = status; o->reserved = 0;
}

// a5 + a6 for one point in the MAP frame.  The single-pair kernel forms J_k = (dR/dtheta_k) p per point and
// sums w C J_k, w J_k' C J_l, w v'J_k (accumulate_point3: 131 arithmetic instructions per point).  With
// y = R p every rotation column is a cross product, J_k = a_k x y (a_roll = R e_x, a_pitch = Rz e_y,
// a_yaw = e_z: the same for every point), so with Y = [y]x (Y b = y x b), J = -Y A and
//     Htr = -(sum w C Y) A = -P A      Hrr = A' (sum w Y' C Y) A = A' S A      g_r = A' sum w (y x v) = A' n
// The point loop sums P (9), S (6), n (3) - as many accumulators as before, 102 arithmetic instructions -
// and the contraction with A happens once per iteration on the reduced sums, in float64
// (map_sums_to_pose_frame).  Newton mode: - d2 w (J'v)(J'v)' with J'v = (v ; A'u), u = y x v, lands on the
// same three sums, and the second-derivative term uses M' = sum w v y' = M R'.
// acc: Htt(6) P(9) S(6) g_t(3) n(3) score n_hit [M'(9)].  y and p' must be finite; a point outside the grid comes with
// the all-zero record.
template <int MODE>
__device__ __forceinline__ void accumulate_point3_map(float yx, float yy, float yz, float px, float py, float pz, bool in,
                                                      const float4& A4, const float4& B4, const float C1, float d1,
                                                      float d2, float nhd2, float* acc) {
  const float cxx = A4.w, cxy = B4.x, cxz = B4.y, cyy = B4.z, cyz = B4.w, czz = C1;
  const bool hit = in & (cxx > 0.f);
  const float qx = px - A4.x, qy = py - A4.y, qz = pz - A4.z;
  const float vx = fmaf(cxx, qx, fmaf(cxy, qy, cxz * qz));
  const float vy = fmaf(cxy, qx, fmaf(cyy, qy, cyz * qz));
  const float vz = fmaf(cxz, qx, fmaf(cyz, qy, czz * qz));
  const float m = fmaf(qx, vx, fmaf(qy, vy, qz * vz));
  const float s = hit ? d1 * __builtin_amdgcn_exp2f(nhd2 * m) : 0.f;
  const float w = s * d2;
  const float wxx = w * cxx, wxy = w * cxy, wxz = w * cxz, wyy = w * cyy, wyz = w * cyz, wzz = w * czz;   // w C
  acc[0] += wxx; acc[1] += wxy; acc[2] += wxz; acc[3] += wyy; acc[4] += wyz; acc[5] += wzz;
  const float wvx = w * vx, wvy = w * vy, wvz = w * vz;
  acc[21] += wvx; acc[22] += wvy; acc[23] += wvz;
  acc[24] = fmaf(yy, wvz, fmaf(-yz, wvy, acc[24]));            // n += y x (w v)
  acc[25] = fmaf(yz, wvx, fmaf(-yx, wvz, acc[25]));
  acc[26] = fmaf(yx, wvy, fmaf(-yy, wvx, acc[26]));
  // Z = (w C) Y, Y = [[0, -yz, yy], [yz, 0, -yx], [-yy, yx, 0]]
  const float wc[3][3] = {{wxx, wxy, wxz}, {wxy, wyy, wyz}, {wxz, wyz, wzz}};
  float Z[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Z[r][0] = fmaf(wc[r][1], yz, -(wc[r][2] * yy));
    Z[r][1] = fmaf(wc[r][2], yx, -(wc[r][0] * yz));
    Z[r][2] = fmaf(wc[r][0], yy, -(wc[r][1] * yx));
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[6 + 3 * r + c] += Z[r][c];
  }
  // S = Y' Z (symmetric): row 0 of Y' = (0, yz, -yy), row 1 = (-yz, 0, yx), row 2 = (yy, -yx, 0)
  acc[15] = fmaf(yz, Z[1][0], fmaf(-yy, Z[2][0], acc[15]));
  acc[16] = fmaf(yz, Z[1][1], fmaf(-yy, Z[2][1], acc[16]));
  acc[17] = fmaf(yz, Z[1][2], fmaf(-yy, Z[2][2], acc[17]));
  acc[18] = fmaf(yx, Z[2][1], fmaf(-yz, Z[0][1], acc[18]));
  acc[19] = fmaf(yx, Z[2][2], fmaf(-yz, Z[0][2], acc[19]));
  acc[20] = fmaf(yy, Z[0][2], fmaf(-yx, Z[1][2], acc[20]));
  if (MODE == 1) {
    const float wd = -d2 * w;
    const float ux = fmaf(yy, vz, -(yz * vy)), uy = fmaf(yz, vx, -(yx * vz)), uz = fmaf(yx, vy, -(yy * vx));   // u = y x v
    const float v3[3] = {vx, vy, vz}, u3[3] = {ux, uy, uz}, y3[3] = {yx, yy, yz}, wv3[3] = {wvx, wvy, wvz};
    const float dv[3] = {wd * vx, wd * vy, wd * vz}, du[3] = {wd * ux, wd * uy, wd * uz};
    int q = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) {
        acc[q] = fmaf(dv[i], v3[j], acc[q]);                   // Htt - d2 w v v'
        acc[15 + q] = fmaf(du[i], u3[j], acc[15 + q]);         // S   - d2 w u u'
        ++q;
      }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[6 + 3 * r + c] = fmaf(-dv[r], u3[c], acc[6 + 3 * r + c]);      // Htr = -P A: P + d2 w v u'
        acc[29 + 3 * r + c] = fmaf(wv3[r], y3[c], acc[29 + 3 * r + c]);    // M' = sum w v y'
      }
  }
  acc[27] += s;
  acc[28] += hit ? 1.f : 0.f;
}

// The reduced sums of accumulate_point3_map (sr: Htt P S g_t n score n_hit [M']) -> the 6x6 matrix and the
// gradient in pose coordinates, at the pose the sums were taken at.  Float64, once per iteration.
template <int MODE>
__device__ __forceinline__ void map_sums_to_pose_frame(const double* pose, const double* sr, double* A, double* g) {
  double sa, ca, sb, cb, sg, cg;
  sincos_wrapped(pose[3], &sa, &ca);
  sincos_wrapped(pose[4], &sb, &cb);
  sincos_wrapped(pose[5], &sg, &cg);
  const double R[9] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa,
                       sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa,
                       -sb, cb * sa, cb * ca};
  const double ax[3][3] = {{R[0], -sg, 0.0}, {R[3], cg, 0.0}, {R[6], 0.0, 1.0}};     // columns a_roll, a_pitch, a_yaw
  A[0] = sr[0]; A[1] = sr[1]; A[2] = sr[2]; A[7] = sr[3]; A[8] = sr[4]; A[14] = sr[5];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) t -= sr[6 + 3 * r + c] * ax[c][k];
      A[6 * r + 3 + k] = t;
    }
  const double S[3][3] = {{sr[15], sr[16], sr[17]}, {sr[16], sr[18], sr[19]}, {sr[17], sr[19], sr[20]}};
  double SA[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int l = 0; l < 3; ++l) SA[a][l] = S[a][0] * ax[0][l] + S[a][1] * ax[1][l] + S[a][2] * ax[2][l];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int l = k; l < 3; ++l) A[6 * (3 + k) + 3 + l] = ax[0][k] * SA[0][l] + ax[1][k] * SA[1][l] + ax[2][k] * SA[2][l];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < r; ++c) A[6 * r + c] = A[6 * c + r];
  g[0] = sr[21]; g[1] = sr[22]; g[2] = sr[23];
#pragma unroll
  for (int k = 0; k < 3; ++k) g[3 + k] = ax[0][k] * sr[24] + ax[1][k] * sr[25] + ax[2][k] * sr[26];
  if (MODE == 1) {
    double M[9];                                               // M = M' R: the sums newton_rot_block3 expects
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) M[3 * a + b] = sr[29 + 3 * a] * R[b] + sr[30 + 3 * a] * R[3 + b] + sr[31 + 3 * a] * R[6 + b];
    newton_rot_block3(pose, M, A);
  }
}

// One pair, start to finish, on the calling workgroup (plain returns for the early outs: see process_pair).
template <int MODE, bool GLOBAL>
__device__ __forceinline__ void process_pair3(const Batch3Args& a, const int pair, unsigned char* smem) {
  constexpr int NA = Acc3<MODE>::kUsed;
  using IdxT = typename std::conditional<GLOBAL, unsigned int, unsigned short>::type;
  constexpr int kS = GLOBAL ? kG3MaxSlots : kB3MaxSlots;       // row stride of the sums in the slab
  float* red = reinterpret_cast<float*>(smem + kB3Red);
  double* bc = reinterpret_cast<double*>(smem + kB3Bc);
  int* misc = reinterpret_cast<int*>(smem + kB3Misc);
  int* s_scan = reinterpret_cast<int*>(smem + kB3Scan);
  LineSearch3* ls_lds = reinterpret_cast<LineSearch3*>(smem + kB3Ls);
  unsigned char* slab = GLOBAL ? a.gslab + (size_t)blockIdx.x * kG3SlabBytes : a.slab + (size_t)blockIdx.x * kB3SlabBytes;
  IdxT* idx = GLOBAL ? reinterpret_cast<IdxT*>(slab + kG3Idx) : reinterpret_cast<IdxT*>(smem + kB3Idx);
  unsigned int* slot_n = reinterpret_cast<unsigned int*>(slab + (GLOBAL ? kG3SlotN : (size_t)kB3SlabN));
  unsigned int* slot_key = reinterpret_cast<unsigned int*>(slab + (GLOBAL ? kG3SlotKey : (size_t)kB3SlabKey));
  unsigned long long* gsums = reinterpret_cast<unsigned long long*>(slab + (GLOBAL ? kG3Sums : (size_t)kB3SlabSums));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int minpts = a.min_points < 2 ? 2 : a.min_points;

  const unsigned long long t0 = uniform64(a.toff[pair]), s0 = uniform64(a.soff[pair]);
  const unsigned long long nt64 = uniform64(a.toff[pair + 1]) - t0, ns64 = uniform64(a.soff[pair + 1]) - s0;
  double pose[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) pose[j] = a.init[6 * pair + j];
  Result3Dev* out = a.out + pair;
  if (nt64 > (unsigned long long)kBatchMaxCloud || ns64 > (unsigned long long)kBatchMaxCloud) {   // uniform; also offsets out of order
    if (tid == 0) write_result3(out, pose, nullptr, nullptr, 0.0, 0, 0, kStatusInvalid);
    return;
  }
#pragma unroll
  for (int j = 3; j < 6; ++j) pose[j] = wrap_angle(pose[j]);
  const int nt = (int)nt64, ns = (int)ns64;
  const float* __restrict__ tx = a.tx + t0;
  const float* __restrict__ ty = a.ty + t0;
  const float* __restrict__ tz = a.tz + t0;
  const float* __restrict__ sx = a.sx + s0;
  const float* __restrict__ sy = a.sy + s0;
  const float* __restrict__ sz = a.sz + s0;
  int iter_base = 0;
  if (a.chain) {                                   // uniform
    const int pst = __builtin_amdgcn_readfirstlane(out->status);
    if (pst != 0 && pst != 1) return;              // the coarser level's failure is the pair's result
#pragma unroll
    for (int j = 0; j < 6; ++j) pose[j] = out->pose[j];
    iter_base = __builtin_amdgcn_readfirstlane(out->iterations);
    __syncthreads();                               // every wave has read out[pair] before anyone rewrites it
  }

#ifdef NDT_B3_PHASE_CLOCKS
  long long pc_t[8]; pc_t[0] = wall_clock64();
#endif
  // ---- a1: bounding box of the target and grid geometry (oracle/ndt3d.py grid_geometry3)
  {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for_each_target_point3(tx, ty, tz, nt, [&](float u, float v, float w) {
      if (isfinite(u) && isfinite(v) && isfinite(w)) {
        mn[0] = fminf(mn[0], u); mx[0] = fmaxf(mx[0], u);
        mn[1] = fminf(mn[1], v); mx[1] = fmaxf(mx[1], v);
        mn[2] = fminf(mn[2], w); mx[2] = fmaxf(mx[2], w);
      }
    });
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]);
      if (lane == 0) { red[wave * 6 + 2 * c] = mn[c]; red[wave * 6 + 2 * c + 1] = mx[c]; }
    }
    __syncthreads();
    if (tid == 0) {
      int st = 0, dims[3] = {0, 0, 0};
      float o[3] = {0.f, 0.f, 0.f};
      const float inv_c = (float)(1.0 / a.cell);
      double ncell_d = 1.0;
      for (int c = 0; c < 3; ++c) {
        for (int w = 1; w < kB3Waves; ++w) { mn[c] = fminf(mn[c], red[w * 6 + 2 * c]); mx[c] = fmaxf(mx[c], red[w * 6 + 2 * c + 1]); }
        if (!(mn[c] <= mx[c])) { st = 4; break; }     // no finite target point -> no valid voxel
        o[c] = (float)((floor((double)mn[c] / a.cell) - 1.0) * a.cell);
        const float k = floorf((mx[c] - o[c]) * inv_c);
        if (!(k >= 0.f) || k > 65534.f) { st = kStatusCapacity; break; }
        dims[c] = (int)k + 2;
        ncell_d *= (double)dims[c];
      }
      // the voxel table (2 B per voxel) and, during the build, the counts (4 B per voxel) behind it
      if (st == 0 && (GLOBAL ? ncell_d > (double)kG3MaxCells : kB3Idx + 16 + 6.0 * ncell_d > (double)kB3LdsBytes)) st = kStatusCapacity;
      misc[1] = dims[0]; misc[2] = dims[1]; misc[3] = dims[2]; misc[4] = st;
      reinterpret_cast<float*>(misc)[5] = o[0];
      reinterpret_cast<float*>(misc)[6] = o[1];
      reinterpret_cast<float*>(misc)[7] = o[2];
    }
    __syncthreads();
  }
  const int W = __builtin_amdgcn_readfirstlane(misc[1]), Hh = __builtin_amdgcn_readfirstlane(misc[2]),
            D = __builtin_amdgcn_readfirstlane(misc[3]);
  const int st0 = __builtin_amdgcn_readfirstlane(misc[4]);
  const float ox = uniformf(reinterpret_cast<float*>(misc)[5]), oy = uniformf(reinterpret_cast<float*>(misc)[6]),
              oz = uniformf(reinterpret_cast<float*>(misc)[7]);
  const float inv_c = (float)(1.0 / a.cell);
  const float fW = (float)W, fH = (float)Hh, fD = (float)D;
  const int ncell = W * Hh * D;
  const double fix_scale = 4194304.0 / a.cell;     // 2^kFixShift / c
  static_assert(kFixShift == 22, "fix_scale literal");
  __syncthreads();                                 // misc is rewritten below
  if (st0 != 0) {                                  // uniform
    if (tid == 0) {
      if (!GLOBAL && st0 == kStatusCapacity && a.fb_marks) a.fb_marks[pair] = 1;      // too many voxels for the LDS carve
      else write_result3(out, pose, nullptr, nullptr, 0.0, iter_base, 0, st0);
    }
    return;
  }
  // on chip: per-pair carve behind the voxel table; global variant: table (and the build's counts, converted to slot
  // indices in place) in fixed regions of the slab, and the whole carve for records - the first 4 461 of them live in
  // LDS here too (a 0.8 m grid over the 40 m room has 3 900), the rest in the slab
  const int rec_base = GLOBAL ? kB3Idx : (kB3Idx + 2 * ncell + 15) & ~15;
  const int slot_cap = (kB3LdsBytes - rec_base) / kB3RecBytes;                            // records incl. the dummy record 0
  unsigned int* cnt = GLOBAL ? reinterpret_cast<unsigned int*>(slab + kG3Idx) : reinterpret_cast<unsigned int*>(smem + rec_base);
  unsigned long long* psum = reinterpret_cast<unsigned long long*>(smem + rec_base);      // on-chip build: u64 [per][nslot]
  float4* recA = reinterpret_cast<float4*>(smem + rec_base);
  float4* recB = reinterpret_cast<float4*>(smem + rec_base + 16 * slot_cap);
  float* recC = reinterpret_cast<float*>(smem + rec_base + 32 * slot_cap);

  // (voxel keys are formed with 24-bit multiplies - every index and the key itself stay below 2^21 - which issue at
  // full rate; the 32-bit form compiles to quarter-rate v_mad_u64_u32)
  auto voxel_of = [&](float px, float py, float pz, int& ix, int& iy, int& iz) -> bool {
    const float fx = (px - ox) * inv_c, fy = (py - oy) * inv_c, fz = (pz - oz) * inv_c;
    const bool in = (fx >= 0.f) & (fx < fW) & (fy >= 0.f) & (fy < fH) & (fz >= 0.f) & (fz < fD);
    ix = (int)fx; iy = (int)fy; iz = (int)fz;
    return in;
  };

#ifdef NDT_B3_PHASE_CLOCKS
  pc_t[1] = wall_clock64();
#endif
  // ---- a2 (1/2): per-voxel counts
  if constexpr (GLOBAL) {
    // global tables: a range of voxels at a time in LDS (the kernel has the CU's LDS to itself; one pass over the target
    // per range), stored to the slab with plain stores - instead of one atomic at L2 per point
    unsigned int* pc = reinterpret_cast<unsigned int*>(smem + kB3Idx);
    constexpr int kPassCells = (kB3LdsBytes - kB3Idx) / 4;
#pragma unroll 1
    for (int c0 = 0; c0 < ncell; c0 += kPassCells) {
      const int np = ncell - c0 < kPassCells ? ncell - c0 : kPassCells;
      for (int k = tid; k < np; k += kB3Threads) pc[k] = 0u;
      __syncthreads();
      for_each_target_point3(tx, ty, tz, nt, [&](float px, float py, float pz) {        // (the row-wise walk of the on-chip
        int ix, iy, iz;                                                                  //  build measured slower here)
        if (voxel_of(px, py, pz, ix, iy, iz)) {
          const int r = __mul24(__mul24(iz, Hh) + iy, W) + ix - c0;
          if ((unsigned)r < (unsigned)np) atomicAdd(&pc[r], 1u);
        }
      });
      __syncthreads();
      for (int k = tid; k < np; k += kB3Threads) cnt[c0 + k] = pc[k];
      __syncthreads();
    }
    __threadfence();                                 // the slab's words were written by other waves of this workgroup
  } else {
    for (int k = tid; k < ncell; k += kB3Threads) cnt[k] = 0u;
    __syncthreads();
    {
      int cur = -1;                                    // the lane's current run of points of one voxel
      unsigned int run = 0u;
      for_each_target_point_rows3(tx, ty, tz, nt, [&](float px, float py, float pz) {
        int ix, iy, iz;
        const int key = voxel_of(px, py, pz, ix, iy, iz) ? __mul24(__mul24(iz, Hh) + iy, W) + ix : -1;
        if (key != cur) {
          if (cur >= 0) atomicAdd(&cnt[cur], run);
          cur = key; run = 0u;
        }
        ++run;
      });
      if (cur >= 0) atomicAdd(&cnt[cur], run);
    }
    __syncthreads();
  }

#ifdef NDT_B3_PHASE_CLOCKS
  pc_t[2] = wall_clock64();
#endif
  // ---- compaction: voxels with n >= min_points get a slot, in voxel order (deterministic)
  const int chunk = (ncell + kB3Threads - 1) / kB3Threads;
  const int c0 = tid * chunk < ncell ? tid * chunk : ncell;
  const int c1 = c0 + chunk < ncell ? c0 + chunk : ncell;
  int local = 0;
  for (int k = c0; k < c1; ++k) local += (cnt[k] >= (unsigned)minpts) ? 1 : 0;
  int nslot = 0;
  int s = block_excl_scan<Cfg1024>(local, s_scan, &nslot);
  nslot = __builtin_amdgcn_readfirstlane(nslot);
  // On chip, a pair with more occupied voxels than records fit keeps the first slot_cap records in LDS and the rest in
  // the slab ("overflow": gathered through L2, a few percent of the lookups) as long as the slab bound holds.
#ifndef NDT_B3_OVERFLOW
#define NDT_B3_OVERFLOW 1          // 0 (tools only): over-capacity pairs always go to the global-table variant
#endif
  const bool overflow = (GLOBAL || NDT_B3_OVERFLOW) && nslot + 1 > slot_cap;          // uniform
  if (nslot > (GLOBAL ? kS - 1 : kS) || nslot < 1 || (!GLOBAL && !NDT_B3_OVERFLOW && nslot + 1 > slot_cap) ||
      (!GLOBAL && 8 * nslot > kB3LdsBytes - rec_base)) {   // (record 0 is the dummy; one sum per slot must fit a build pass)
    if (tid == 0) {
      if (!GLOBAL && nslot >= 1 && a.fb_marks) a.fb_marks[pair] = 1;                  // too many occupied voxels for the carve
      else write_result3(out, pose, nullptr, nullptr, 0.0, iter_base, 0, nslot < 1 ? 4 : kStatusCapacity);
    }
    return;
  }
  float4* ovA = reinterpret_cast<float4*>(slab + (GLOBAL ? kG3RecA : (size_t)kB3SlabRecA));       // record of slot s >= slot_cap at [s - slot_cap]
  float4* ovB = reinterpret_cast<float4*>(slab + (GLOBAL ? kG3RecB : (size_t)kB3SlabRecB));
  float* ovC = reinterpret_cast<float*>(slab + (GLOBAL ? kG3RecC : (size_t)kB3SlabRecC));
  for (int k = c0; k < c1; ++k) {
    const unsigned int n = cnt[k];                 // (global variant: read before idx[k], the same word, is written)
    if (n >= (unsigned)minpts) {
      idx[k] = (IdxT)(s + 1);
      slot_n[s] = n;
      slot_key[s] = (unsigned)k;
      ++s;
    } else {
      idx[k] = 0;
    }
  }
  __syncthreads();                                 // cnt is dead; its bytes become the pass sums

#ifdef NDT_B3_PHASE_CLOCKS
  pc_t[3] = wall_clock64();
#endif
  // One pass over the target for rows [j0, j1) of the nine exact sums (s[0..2], ss[0..5] = xx xy xz yy yz zz) of the
  // slots [s0, s0 + np): base[(row - j0) * stride + slot - 1 - s0] in LDS.  A lane adds a run of points of one voxel up
  // in registers (for_each_target_point_rows3) and issues the 64-bit LDS atomics once per run; a run ends after 256
  // points at the latest, so the three coordinate sums stay in 32 bits (|u| <= 2^21).
  auto sum_pass = [&](unsigned long long* base, int stride, int j0, int j1, int s0, int np) {
    int cur = -2, r = -1, run = 0;
    double ccx = 0.0, ccy = 0.0, ccz = 0.0;
    int a0 = 0, a1 = 0, a2 = 0;
    long long pr[6] = {0, 0, 0, 0, 0, 0};
    auto flush = [&]() {
      if (r >= 0) {
        unsigned long long* q = base + r;
        const unsigned long long v[9] = {(unsigned long long)(long long)a0, (unsigned long long)(long long)a1,
                                         (unsigned long long)(long long)a2, (unsigned long long)pr[0], (unsigned long long)pr[1],
                                         (unsigned long long)pr[2], (unsigned long long)pr[3], (unsigned long long)pr[4],
                                         (unsigned long long)pr[5]};
#pragma unroll
        for (int c = 0; c < 9; ++c)
          if (c >= j0 && c < j1) atomicAdd(q + (c - j0) * stride, v[c]);      // uniform condition
      }
      a0 = a1 = a2 = 0;
#pragma unroll
      for (int c = 0; c < 6; ++c) pr[c] = 0;
      run = 0;
    };
    for_each_target_point_rows3(tx, ty, tz, nt, [&](float px, float py, float pz) {
      int ix, iy, iz;
      const int key = voxel_of(px, py, pz, ix, iy, iz) ? __mul24(__mul24(iz, Hh) + iy, W) + ix : -1;
      if (key != cur || run == 256) {
        flush();
        cur = key;
        r = -1;
        if (key >= 0) {
          r = (int)idx[key] - 1 - s0;                                        // no slot: r < 0
          if ((unsigned)r >= (unsigned)np) r = -1;
          ccx = cell_centre(ox, ix, a.cell); ccy = cell_centre(oy, iy, a.cell); ccz = cell_centre(oz, iz, a.cell);
        }
      }
      if (r >= 0) {
        const int ux = fix_coord(px, ccx, fix_scale), uy = fix_coord(py, ccy, fix_scale), uz = fix_coord(pz, ccz, fix_scale);
        a0 += ux; a1 += uy; a2 += uz;
        if (3 >= j0 && 3 < j1) pr[0] += (long long)ux * ux;                  // uniform conditions: only this pass's rows
        if (4 >= j0 && 4 < j1) pr[1] += (long long)ux * uy;
        if (5 >= j0 && 5 < j1) pr[2] += (long long)ux * uz;
        if (6 >= j0 && 6 < j1) pr[3] += (long long)uy * uy;
        if (7 >= j0 && 7 < j1) pr[4] += (long long)uy * uz;
        if (8 >= j0 && 8 < j1) pr[5] += (long long)uz * uz;
        ++run;
      }
    });
    flush();
  };

  if constexpr (GLOBAL) {
    // ---- a2 (2/2), global variant: the nine exact sums of a range of slots at a time in LDS, one pass over the target
    // per range (LDS atomics; the first version added them with 64-bit atomics at L2, which bounded the variant)
    unsigned long long* ps = reinterpret_cast<unsigned long long*>(smem + kB3Idx);
    constexpr int kPassSlots = (kB3LdsBytes - kB3Idx) / 72;
#pragma unroll 1
    for (int s0 = 0; s0 < nslot; s0 += kPassSlots) {
      const int np = nslot - s0 < kPassSlots ? nslot - s0 : kPassSlots;
      for (int j = tid; j < 9 * np; j += kB3Threads) ps[j] = 0ull;
      __syncthreads();
      for_each_target_point3(tx, ty, tz, nt, [&](float px, float py, float pz) {
        int ix, iy, iz;
        if (voxel_of(px, py, pz, ix, iy, iz)) {
          const int r = (int)idx[__mul24(__mul24(iz, Hh) + iy, W) + ix] - 1 - s0;      // no slot: r < 0
          if ((unsigned)r < (unsigned)np) {
            const int ux = fix_coord(px, cell_centre(ox, ix, a.cell), fix_scale);
            const int uy = fix_coord(py, cell_centre(oy, iy, a.cell), fix_scale);
            const int uz = fix_coord(pz, cell_centre(oz, iz, a.cell), fix_scale);
            unsigned long long* q = ps + r;
            atomicAdd(q, (unsigned long long)(long long)ux);
            atomicAdd(q + np, (unsigned long long)(long long)uy);
            atomicAdd(q + 2 * np, (unsigned long long)(long long)uz);
            atomicAdd(q + 3 * np, prod64(ux, ux));
            atomicAdd(q + 4 * np, prod64(ux, uy));
            atomicAdd(q + 5 * np, prod64(ux, uz));
            atomicAdd(q + 6 * np, prod64(uy, uy));
            atomicAdd(q + 7 * np, prod64(uy, uz));
            atomicAdd(q + 8 * np, prod64(uz, uz));
          }
        }
      });
      __syncthreads();
      for (int j = tid; j < 9 * np; j += kB3Threads) gsums[(size_t)(j / np) * kS + s0 + (j % np)] = ps[j];
      __syncthreads();
    }
  } else {
    // ---- a2 (2/2): exact fixed-point sums per slot (LDS 64-bit integer atomics).  The records' region (36 B per slot
    // of capacity) holds `per` of the nine 64-bit sums of every occupied slot at a time: config 5 (2 706 slots) five, so
    // two passes over the target; a grid that fills the carve four, so three passes
    int per = (kB3LdsBytes - rec_base) / (8 * nslot);
    per = per > 9 ? 9 : per;
#pragma unroll 1
    for (int j0 = 0; j0 < 9; j0 += per) {
      const int j1 = j0 + per < 9 ? j0 + per : 9;
      for (int j = tid; j < (j1 - j0) * nslot; j += kB3Threads) psum[j] = 0ull;
      __syncthreads();
      sum_pass(psum, nslot, j0, j1, 0, nslot);
      __syncthreads();
      for (int j = tid; j < (j1 - j0) * nslot; j += kB3Threads)
        gsums[(size_t)(j0 + j / nslot) * kS + (j % nslot)] = psum[j];
      __syncthreads();
    }
  }
  if (tid == 0) misc[8] = 0;
  __threadfence();                                 // the slab was written with plain stores by other waves of this workgroup
  __syncthreads();

#ifdef NDT_B3_PHASE_CLOCKS
  pc_t[4] = wall_clock64();
#endif
  // ---- a3: finalise, records into LDS (those beyond the carve: into the slab)
  {
    int nvalid = 0;
    for (int sl = tid; sl < nslot; sl += kB3Threads) {
      CellAcc3 c;
      c.n = slot_n[sl]; c.pad = 0u;
#pragma unroll
      for (int j = 0; j < 3; ++j) c.s[j] = (long long)gsums[(size_t)j * kS + sl];
#pragma unroll
      for (int j = 0; j < 6; ++j) c.ss[j] = (long long)gsums[(size_t)(3 + j) * kS + sl];
      const unsigned int key = slot_key[sl], w32 = (unsigned)W, h32 = (unsigned)Hh;
      const int ix = (int)(key % w32), iy = (int)((key / w32) % h32), iz = (int)(key / (w32 * h32));
      float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
      bool ok = false;
      if (c.n <= kMaxCellCount)
        ok = finalise_sums3(c, cell_centre(ox, ix, a.cell), cell_centre(oy, iy, a.cell), cell_centre(oz, iz, a.cell), fix_scale,
                            a.min_points, a.eig_ratio, ra, rb, rc);
      if (!ok) { ra = make_float4(0.f, 0.f, 0.f, 0.f); rb = ra; rc = ra; }
      // finalise_sums3's record (mean | n, xx xy xz yy, yz zz) repacked into 9 floats
      const float4 qa = make_float4(ra.x, ra.y, ra.z, rb.x), qb = make_float4(rb.y, rb.z, rb.w, rc.x);
      if (sl + 1 < slot_cap) { recA[sl + 1] = qa; recB[sl + 1] = qb; recC[sl + 1] = rc.y; }
      else { ovA[sl + 1 - slot_cap] = qa; ovB[sl + 1 - slot_cap] = qb; ovC[sl + 1 - slot_cap] = rc.y; }
      nvalid += ok ? 1 : 0;
    }
    if (tid == 0) { recA[0] = make_float4(0.f, 0.f, 0.f, 0.f); recB[0] = recA[0]; recC[0] = 0.f; }
    if (nvalid) atomicAdd(&misc[8], nvalid);
    if (GLOBAL || overflow) __threadfence();       // records (and table) are read through L1 / L2 from here on
    __syncthreads();
  }
  if (__builtin_amdgcn_readfirstlane(misc[8]) < 1) {   // uniform
    __syncthreads();
    if (tid == 0) write_result3(out, pose, nullptr, nullptr, 0.0, iter_base, 0, 4);
    return;
  }

#ifdef NDT_B3_PHASE_CLOCKS
  pc_t[5] = wall_clock64();
#endif
  // ---- a4-a8: Gauss-Newton loop, all on this CU
  if (tid == 0) { misc[9] = 0; misc[10] = 0; ls_lds->valid = 0; ls_lds->trials = 0; }
  __syncthreads();
  const float d1 = a.prm.d1, d2 = a.prm.d2;
  const float nhd2 = -0.5f * d2 * 1.44269504088896340736f;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)sx, 0, ns * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)sy, 0, ns * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)sz, 0, ns * 4, 0x00020000);
  for (;;) {
    float acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) acc[j] = 0.f;
    {
      Rot3F T;
      make_rot3(pose, T);
      // the pose is the same in every lane: SGPRs, not VGPRs held across the point loop
#pragma unroll
      for (int j = 0; j < 9; ++j) T.R[j] = uniformf(T.R[j]);
      T.tx = uniformf(T.tx); T.ty = uniformf(T.ty); T.tz = uniformf(T.tz);
      // software-pipelined source stream, as in process_pair: two register sets, one in flight while
      // the other is consumed
      constexpr int U = kB3Unroll, kTrip = U * kB3Threads;
      float xa[U], ya[U], za[U], xb[U], yb[U], zb[U];
      auto load_set = [&](int base, float* x, float* y, float* z) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int off = (base + u * kB3Threads) * 4;
          x[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
          y[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
          z[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, off, 0, 0));
        }
      };
      // the U points of a set: all voxel lookups first (their LDS round trips overlap), then the sums.  OV: some
      // records live in the slab (uniform per pair, so the common loop carries nothing of it)
      auto consume_set = [&](auto ov_tag, int base, const float* x, const float* y, const float* z) {
        constexpr bool OV = decltype(ov_tag)::value;
        float yx[U], yy[U], yz[U], px[U], py[U], pz[U];
        bool in[U];
        float4 A4[U], B4[U];
        float C1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // NaN / inf (no-return points) never reach the sums: coordinates are clamped to +-1e15 first (v_med3_f32
          // returns the finite bound for a NaN), the image then lies outside the grid and reads the all-zero record
          // 0, against which every product below is a finite number times zero
          const float cx = __builtin_amdgcn_fmed3f(x[u], -1e15f, 1e15f), cy = __builtin_amdgcn_fmed3f(y[u], -1e15f, 1e15f),
                      cz = __builtin_amdgcn_fmed3f(z[u], -1e15f, 1e15f);
          yx[u] = fmaf(T.R[0], cx, fmaf(T.R[1], cy, T.R[2] * cz));
          yy[u] = fmaf(T.R[3], cx, fmaf(T.R[4], cy, T.R[5] * cz));
          yz[u] = fmaf(T.R[6], cx, fmaf(T.R[7], cy, T.R[8] * cz));
          px[u] = yx[u] + T.tx; py[u] = yy[u] + T.ty; pz[u] = yz[u] + T.tz;
          // voxel indices by floor (one saturating v_cvt_flr_i32_f32 each; the coordinates are finite), in range iff
          // 0 <= index < extent as ONE unsigned compare per axis - the same voxel as (int)f under f >= 0 && f < extent
          const int ix = floor_to_int((px[u] - ox) * inv_c), iy = floor_to_int((py[u] - oy) * inv_c),
                    iz = floor_to_int((pz[u] - oz) * inv_c);
          in[u] = ((base + u * kB3Threads) < ns) & ((unsigned)ix < (unsigned)W) & ((unsigned)iy < (unsigned)Hh) &
                  ((unsigned)iz < (unsigned)D);
          const int key = in[u] ? (__mul24(__mul24(iz, Hh) + iy, W) + ix) : 0;
          int slot = (int)idx[key];
          if (!in[u]) slot = 0;
          if (OV && slot >= slot_cap) { A4[u] = ovA[slot - slot_cap]; B4[u] = ovB[slot - slot_cap]; C1[u] = ovC[slot - slot_cap]; }
          else { A4[u] = recA[slot]; B4[u] = recB[slot]; C1[u] = recC[slot]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          accumulate_point3_map<MODE>(yx[u], yy[u], yz[u], px[u], py[u], pz[u], in[u], A4[u], B4[u], C1[u], d1, d2, nhd2, acc);
      };
      auto point_loop = [&](auto ov_tag) {
        load_set(tid, xa, ya, za);
        for (int i = tid; i < ns; i += 2 * kTrip) {
          load_set(i + kTrip, xb, yb, zb);
          consume_set(ov_tag, i, xa, ya, za);
          load_set(i + 2 * kTrip, xa, ya, za);
          if (i + kTrip < ns) consume_set(ov_tag, i + kTrip, xb, yb, zb);     // wave-uniform except at the tail
        }
      };
      if (ns > 0) {                                  // uniform
        if (overflow) point_loop(std::true_type{});
        else point_loop(std::false_type{});
      }
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const float rsum = wave_sum_lane63(acc[j]);
      if (lane == 63) red[wave * kNumAcc3 + j] = rsum;
    }
    __syncthreads();
    if (wave == 0) {
      // lane j < NA sums column j over the waves in a fixed order, in float64, and parks it in LDS;
      // every lane reads the totals back (broadcast reads)
      if (lane < NA) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kB3Waves; ++w) tot += (double)red[w * kNumAcc3 + lane];
        bc[6 + lane] = tot;
      }
      __builtin_amdgcn_wave_barrier();               // same wave: LDS executes its operations in order
      const double* sr = bc + 6;
      double A[36], g[6];
      map_sums_to_pose_frame<MODE>(pose, sr, A, g);
      const double score = sr[27];
      const int n_hit = (int)(sr[28] + 0.5);
      int iter = misc[9], st = 0;
      const bool done = gn_update3(pose, A, g, n_hit, iter, st, a.prm, a.fixed_iterations, score, ls_lds, ls_lds, lane == 0);
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) bc[j] = pose[j];
        // the result reports H and g in pose coordinates: Htt stays, P / S / n make way for Htr / Hrr / g_r
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int k = 0; k < 3; ++k) bc[6 + 6 + 3 * r + k] = A[6 * r + 3 + k];
        bc[6 + 15] = A[21]; bc[6 + 16] = A[22]; bc[6 + 17] = A[23]; bc[6 + 18] = A[28]; bc[6 + 19] = A[29]; bc[6 + 20] = A[35];
        bc[6 + 24] = g[3]; bc[6 + 25] = g[4]; bc[6 + 26] = g[5];
        misc[11] = done ? 1 : 0;
        misc[9] = iter;
        misc[10] = st;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 6; ++j) pose[j] = bc[j];
    const int done = __builtin_amdgcn_readfirstlane(misc[11]);
    if (done) break;
    // no third barrier, as in process_pair: the next iteration writes `red` after its point loop and
    // bc / misc only after its own first barrier
  }
  if (tid == 0)
    write_result3(out, pose, bc + 6, bc + 6 + 21, bc[6 + 27], misc[9] + iter_base, (int)(bc[6 + 28] + 0.5), misc[10]);
#ifdef NDT_B3_PHASE_CLOCKS
  if (tid == 0) {                  // tools: 100 MHz ticks per phase in the unused lower triangle of H
    pc_t[6] = wall_clock64();
    for (int j = 0; j < 6; ++j) out->H[30 + j] = (double)(pc_t[j + 1] - pc_t[j]);
  }
#endif
}

template <int MODE>
__global__ __launch_bounds__(kB3Threads) void k_batch3(Batch3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* misc = reinterpret_cast<int*>(smem + kB3Misc);
  for (;;) {
    if (threadIdx.x == 0) misc[0] = (int)atomicAdd(a.queue, 1u);
    __syncthreads();
    const int pair = __builtin_amdgcn_readfirstlane(misc[0]);
    __syncthreads();
    if (pair >= a.n_pairs) break;
    process_pair3<MODE, false>(a, pair, smem);
    __syncthreads();                                 // LDS and the slab are rewritten by the next pair
  }
}

// The pairs k_batch3 handed over (fb_marks), tables in global memory.  No queue: workgroup b looks at pairs
// b, b + gridDim.x, ... (one scalar load each; with nothing handed over the launch costs a few microseconds).
template <int MODE>
__global__ __launch_bounds__(kB3Threads) void k_batch3_fallback(Batch3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Workgroup b owns the pairs b, b + gridDim.x, ...; its threads read their marks side by side (one dependent scalar
  // load per pair cost 13 us on the 8 workgroups a context starts with, with nothing marked) and process the marked
  // ones in turn.
  int* const fb_any = reinterpret_cast<int*>(smem + kB3Misc) + 15;                     // (a word of the carve no pair uses: the carve fills the CU's LDS)
  const int owned = (a.n_pairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // pairs of this workgroup
  for (int base = 0; base < owned; base += (int)blockDim.x) {
    if (threadIdx.x == 0) *fb_any = 0;
    __syncthreads();
    const int j = base + (int)threadIdx.x;
    const bool mine = j < owned && a.fb_marks[(size_t)blockIdx.x + (size_t)j * gridDim.x] != 0;
    if (mine) *fb_any = 1;                                // benign race: every writer stores 1
    __syncthreads();
    const int any = __builtin_amdgcn_readfirstlane(*fb_any);       // uniform
    __syncthreads();                                      // (process_pair rewrites the carve)
    const int last = base + (int)blockDim.x < owned ? base + (int)blockDim.x : owned;
    for (int jj = base; jj < (any ? last : base); ++jj) {
      const int pair = (int)blockIdx.x + jj * (int)gridDim.x;
      if (__builtin_amdgcn_readfirstlane(a.fb_marks[pair]) != 0) {            // uniform
        if (threadIdx.x == 0 && a.fb_seen) __hip_atomic_fetch_add(a.fb_seen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        process_pair3<MODE, true>(a, pair, smem);
        __syncthreads();
        __threadfence();                                  // the next pair rewrites this workgroup's slab
      }
    }
  }
}

}  // namespace ndt
