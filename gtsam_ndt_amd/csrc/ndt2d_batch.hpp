// Loop-closure candidate batch (BASELINE config 4): persistent workgroups (one of 1024 threads per
// CU for submap-sized pairs, two of 256 threads for lidar-sized ones: BatchCfg below) pull scan
// pairs from a queue and run the WHOLE alignment of a pair on chip:
//   - the target's NDT grid is built and kept in LDS (dense cell -> slot index table plus
//     compact per-slot records), so the per-point cell lookup of every Gauss-Newton
//     iteration is an LDS read ("LDS-staged cell stats", BASELINE.json north_star);
//   - source points are streamed from HBM/L2 with coalesced SoA loads, once per iteration;
//   - the 11 sums are reduced per wave (DPP, or through LDS in the small variant) -> LDS -> one
//     wave, which also does the 3x3 solve, so an iteration costs two workgroup barriers and no
//     kernel boundary, no grid barrier, no atomics on floats.
// Same arithmetic as the single-pair path (shared device functions of ndt2d_kernels.hpp):
// the LDS grid holds bit-identical records to k_accumulate/k_finalise.
#pragma once
#include <type_traits>

#include "ndt2d_kernels.hpp"

namespace ndt {

#ifndef NDT_BATCH_THREADS
#define NDT_BATCH_THREADS 1024
#endif
#ifndef NDT_BATCH_UNROLL
#define NDT_BATCH_UNROLL 2
#endif
constexpr int kBatchUnroll = NDT_BATCH_UNROLL;   // source points per thread and register set

struct ResultDev {   // layout of ndt2d_result (include/ndt_hip.h); static_assert in the API file
  double pose[3];
  double H[9];
  double g[3];
  double score;
  int iterations, n_hit, status, reserved;
};

struct BatchArgs {
  const float* tx; const float* ty; const unsigned long long* toff;   // targets, concatenated SoA
  const float* sx; const float* sy; const unsigned long long* soff;   // sources
  const double* init;        // [n_pairs][3]
  ResultDev* out;            // [n_pairs]
  unsigned int* queue;       // zeroed before the launch
  int* marks;                // [n_pairs], written by the small variant for every pair: 1 = exceeds its limits,
                             // left alone; the large variant then processes exactly the marked pairs
                             // (null: every pair)
  unsigned char* slab;       // global-memory tables of the third variant: [workgroup][BatchGlobal::kTabBytes]
  int* fb_marks;             // [n_pairs], zeroed before the launches: the large variant sets 1 for a pair over
                             // the on-chip capacity, the third variant processes exactly those
                             // (null: such pairs get NDT_ERR_CAPACITY at once)
  unsigned int* fb_seen;     // pinned host word: the third variant counts the pairs it processes here (the context grows
                             // its set of table slabs from the starting few to one per CU once any have been seen)
  int n_pairs;
  int min_points;
  int fixed_iterations;
  int chain;                 // 1: a later level of a coarse-to-fine run - start from out[pair].pose,
                             // add to its iteration count, leave pairs whose earlier level failed
  double cell;
  double eig_ratio;
  SolveParams prm;
};

// Workgroup shape and on-chip capacities of one kernel variant.  LDS carve in bytes: everything in
// one dynamic array, every offset a multiple of 16.
//   Large  1024 threads, one workgroup per CU: 20480 cells (143 x 143), 2559 occupied (159.8 KB of LDS) - the BASELINE config-4
//          pairs (100k-point submap scans); sums reduced by DPP (no LDS left for anything else)
//   Small   256 threads, two workgroups per CU (2 x 78.7 KB of LDS): 128 x 128 cells, 767 occupied, clouds of up to 8192
//          points - single lidar scans; the per-iteration fixed costs of a 16-wave workgroup (DPP
//          trees on four waves per SIMD) are what bounds those, so this variant keeps one wave
//          per SIMD per pair and sums through LDS (wave_reduce11_lds)
template <int THREADS, int MAXCELLS, int MAXSLOTS, int MAXPOINTS, bool LDSSUMS, bool PACKEDCOUNT, bool GLOBALTABLES = false>
struct BatchCfg {
  // Where the grid tables (cell -> slot index, per-slot counts / keys / sums / records) live: in LDS
  // (the two on-chip variants), or in a per-workgroup slab of global memory (the third variant, which
  // takes the pairs whose grid does not fit on chip: 32-bit indices, records gathered through L2).
  static constexpr bool kGlobalTables = GLOBALTABLES;
  using IdxT = typename std::conditional<GLOBALTABLES, unsigned int, unsigned short>::type;
  using KeyT = typename std::conditional<GLOBALTABLES, unsigned int, unsigned short>::type;
  static constexpr int kThreads = THREADS;
  static constexpr int kWaves = THREADS / 64;
  static constexpr int kMaxCells = MAXCELLS;      // dense index table
  static constexpr int kMaxSlots = MAXSLOTS;      // records: slot 0 is a dummy invalid record
  static constexpr int kMaxPoints = MAXPOINTS;    // per cloud; 0 = no limit
  static constexpr bool kSumsViaLds = LDSSUMS;
  // per-cell point counts of the build as u16 pairs in the index table itself (clouds of fewer
  // than 65536 points; the slot index then overwrites each count in place) instead of a u32
  // table aliased onto the sums region - what lets a 128 x 128-cell grid fit a small workgroup
  static constexpr bool kPackedCount = PACKEDCOUNT;
  static constexpr int kSlotsPerThread = (MAXSLOTS + THREADS - 1) / THREADS;
  // the tables, as byte offsets from their base (LDS, or this workgroup's slab)
  static constexpr int kTabIdx = 0;                                                  // IdxT [MaxCells]
  static constexpr int kTabSlotN = kTabIdx + MAXCELLS * (int)sizeof(IdxT);           // u32 [MaxSlots]
  static constexpr int kTabSlotKey = kTabSlotN + MAXSLOTS * 4;                       // KeyT [MaxSlots]
  static constexpr int kTabSums = (kTabSlotKey + MAXSLOTS * (int)sizeof(KeyT) + 15) / 16 * 16;   // u64 [5][MaxSlots]; aliases:
                                                                                     //   u32 cnt[MaxCells] (build)
                                                                                     //   float4 recA[MaxSlots], recB[MaxSlots]
  // global tables: the records get their own region (they are written straight from the finalise loop);
  // on chip they overwrite the sums, staged through registers
  static constexpr int kTabRec = GLOBALTABLES ? kTabSums + 5 * MAXSLOTS * 8 : kTabSums;
  static constexpr int kTabBytes = GLOBALTABLES ? kTabRec + 2 * MAXSLOTS * 16 : kTabSums + 5 * MAXSLOTS * 8;
  static constexpr int kLdsRed = GLOBALTABLES ? 0 : kTabBytes;             // float [Waves][kNumAcc]
  static constexpr int kLdsBc = kLdsRed + kWaves * kNumAcc * 4;            // double [16] broadcast
  static constexpr int kLdsMisc = kLdsBc + 16 * 8;                         // int [16]
  static constexpr int kLdsScan = kLdsMisc + 16 * 4;                       // int [16]
  static constexpr int kLdsLs = kLdsScan + 16 * 4;                         // LineSearch (line-search state)
  static constexpr int kLdsT = kLdsLs + 80;                                // float [Waves][11 * 68] if kSumsViaLds
  static_assert(sizeof(LineSearch) <= 80, "LineSearch slot");
  static constexpr int kLdsEnd = kLdsT + (LDSSUMS ? kWaves * (kNumAcc - 1) * kSumRowStride * 4 : 0);
  // global tables: the rest of the LDS is the build's sum buffer - u64 [5][kPassSlots], a range of slots per pass over
  // the target (LDS atomics instead of five 64-bit atomics at L2 per point, which is what bounded this variant)
  static constexpr int kLdsPass = (kLdsEnd + 15) / 16 * 16;
  static constexpr int kPassSlots = GLOBALTABLES ? (160 * 1024 - kLdsPass) / 40 : 0;
  static constexpr int kLdsBytes = GLOBALTABLES ? kLdsPass + 40 * kPassSlots : kLdsEnd;
  static_assert(PACKEDCOUNT || MAXCELLS * 4 <= 5 * MAXSLOTS * 8, "cnt must fit in the sums region");
  static_assert(!PACKEDCOUNT || (MAXPOINTS > 0 && MAXPOINTS < 65536), "packed counts are 16-bit");
  static_assert(MAXSLOTS * 32 <= 5 * MAXSLOTS * 8, "records must fit in the sums region");
  static_assert(kLdsBytes <= 160 * 1024, "CDNA4 LDS is 160 KiB per CU");
  static_assert((kTabSlotN % 16) == 0 && (kTabSlotKey % 16) == 0 && (kTabSums % 16) == 0 && (kLdsRed % 16) == 0 &&
                (kLdsBc % 16) == 0 && (kLdsMisc % 16) == 0 && (kLdsLs % 16) == 0 && (kLdsT % 16) == 0, "16-byte aligned carve");
  static_assert(GLOBALTABLES || (MAXSLOTS - 1 <= 0xffff && MAXCELLS <= 0x10000), "u16 slot indices and cell keys");
  static_assert(MAXCELLS <= (1 << 24), "24-bit key multiply");
};
using BatchLarge = BatchCfg<NDT_BATCH_THREADS, 20480, 2560, 0, false, false>;
using BatchSmall = BatchCfg<256, 16384, 768, 8192, true, true>;
// Third variant: 512 x 512 cells (256 m x 256 m at 0.5 m cells), 32767 occupied, tables in global memory.
// It exists so that the device-pointer entry point never fails a pair a SLAM front end may legally
// produce (a scan against a large submap); it runs on the context's global_blocks workgroups and only on the pairs
// the large variant handed over.
using BatchGlobal = BatchCfg<1024, 1 << 18, 32768, 0, false, false, true>;
constexpr int kBatchGlobalBlocksStart = 8;   // a context starts with 8 workgroups / 3.7 MB table slabs of the global-table variant (30 MB)
constexpr int kBatchGlobalBlocks = 256;      // ... and grows to one per CU (0.95 GB) after the first call in which a pair needed it;
constexpr int kBatchGlobalBlocksMax = 256;   // NDT_TUNE_BATCH_GLOBAL_WORKGROUPS pins the number instead
constexpr int kBatchThreads = BatchLarge::kThreads;       // names the host code and tools/ use
constexpr int kBatchMaxCells = BatchLarge::kMaxCells;
constexpr int kBatchMaxSlots = BatchLarge::kMaxSlots;
constexpr int kBatchLdsBytes = BatchLarge::kLdsBytes;

// exclusive scan of one int per thread over the 1024-thread workgroup; *total = sum
template <class Cfg>
__device__ __forceinline__ int block_excl_scan(int v, int* s_scan, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < Cfg::kWaves; ++w) {
    const int t = s_scan[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ void write_result(ResultDev* o, const double* pose, const double* H6, const double* g,
                                             double score, int iter, int n_hit, int status) {
  o->pose[0] = pose[0]; o->pose[1] = pose[1]; o->pose[2] = pose[2];
  o->H[0] = H6[0]; o->H[1] = H6[1]; o->H[2] = H6[3];
  o->H[3] = H6[1]; o->H[4] = H6[2]; o->H[5] = H6[4];
  o->H[6] = H6[3]; o->H[7] = H6[4]; o->H[8] = H6[5];
  o->g[0] = g[0]; o->g[1] = g[1]; o->g[2] = g[2];
  o->score = score;
  o->iterations = iter; o->n_hit = n_hit; o->status = status; o->reserved = 0;
}

// a4 with the record served from LDS: dense index table -> slot -> 32-byte record
template <typename IdxT>
__device__ __forceinline__ void lookup_point_lds(const PoseF& P, const IdxT* __restrict__ idx,
                                                 const float4* __restrict__ recA,
                                                 const float4* __restrict__ recB, float x, float y, bool live,
                                                 PointRec& r) {
  const int key = point_key(P, x, y, live, r);
#if defined(NDT_BATCH_ABLATE) && (NDT_BATCH_ABLATE & 1)      // tools/exp_batch.hip only: no LDS traffic
  r.A = make_float4(r.px - 0.01f, r.py + 0.01f, 900.f, 30.f);
  r.B = make_float4(30.f, 800.f, 9.f, 0.f);
#else
  const int slot = idx[key];                       // guard and empty cells hold 0 = the dummy record,
  r.A = recA[slot];                                // whose n = 0 marks it invalid
  r.B = recB[slot];
#endif
}

__device__ __forceinline__ float uniformf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull));
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Target passes: kTgtUnroll points per thread are requested before any is used (a one-point
// loop pays the full memory latency per trip: 98 dependent round trips per pass).  Loads go
// through a buffer descriptor of the pair's slice: lanes past the end read 0 and are skipped.
constexpr int kTgtUnroll = 8;
template <class Cfg, typename F>
__device__ __forceinline__ void for_each_target_point(const float* tx, const float* ty, int nt, F&& body) {
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)tx, 0, nt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)ty, 0, nt * 4, 0x00020000);
  for (int i = threadIdx.x; i < nt; i += kTgtUnroll * Cfg::kThreads) {
    float x[kTgtUnroll], y[kTgtUnroll];
#pragma unroll
    for (int u = 0; u < kTgtUnroll; ++u) {
      const int off = (i + u * Cfg::kThreads) * 4;
      x[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
      y[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < kTgtUnroll; ++u)
      if (i + u * Cfg::kThreads < nt) body(x[u], y[u]);
  }
}

constexpr int kStatusCapacity = -5;   // NDT_ERR_CAPACITY: pair needs the global-memory path
constexpr int kStatusInvalid = -1;    // NDT_ERR_INVALID_ARG
constexpr int kBatchMaxCloud = 1 << 29;   // points per cloud of a pair: byte offsets into it are 32-bit (x 4 < 2^31)

// One pair, start to finish, on the calling workgroup.  Early outs are plain returns: the
// caller's queue loop then has a single back edge (with `continue`s inside the loop body
// hipcc's loop restructuring produced a kernel that re-read the same queue slot forever).
// NG = 4: Biber's four overlapping grids (every grid shifted by half a cell against the others, the terms of all four
// summed: k_iterate's NG, oracle/ndt2d.py build_grids) - on the global-table variant only: on chip four grids would
// quarter the capacity to 71 x 71 cells, less than a config-4 room.  The four grids lie back to back in the tables
// (cell q * W * H + key, slots handed out in that order); one extra column and row, as on the single-pair path.
template <int MODE, class Cfg, int NG = 1>
__device__ __forceinline__ void process_pair(const BatchArgs& a, const int pair, unsigned char* smem) {
  static_assert(NG == 1 || (NG == 4 && Cfg::kGlobalTables), "overlapping grids: global-table variant only");
  using IdxT = typename Cfg::IdxT;
  using KeyT = typename Cfg::KeyT;
  // the grid tables: LDS, or this workgroup's slab of global memory
  unsigned char* tab = smem;
  if constexpr (Cfg::kGlobalTables) tab = a.slab + (size_t)blockIdx.x * Cfg::kTabBytes;
  IdxT* idx = reinterpret_cast<IdxT*>(tab + Cfg::kTabIdx);
  unsigned int* slot_n = reinterpret_cast<unsigned int*>(tab + Cfg::kTabSlotN);
  KeyT* slot_key = reinterpret_cast<KeyT*>(tab + Cfg::kTabSlotKey);
  unsigned long long* sums = reinterpret_cast<unsigned long long*>(tab + Cfg::kTabSums);
  unsigned int* cnt = reinterpret_cast<unsigned int*>(tab + Cfg::kTabSums);
  float4* recA = reinterpret_cast<float4*>(tab + Cfg::kTabRec);
  float4* recB = reinterpret_cast<float4*>(tab + Cfg::kTabRec + Cfg::kMaxSlots * 16);
  float* red = reinterpret_cast<float*>(smem + Cfg::kLdsRed);
  double* bc = reinterpret_cast<double*>(smem + Cfg::kLdsBc);
  int* misc = reinterpret_cast<int*>(smem + Cfg::kLdsMisc);
  int* s_scan = reinterpret_cast<int*>(smem + Cfg::kLdsScan);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int minpts = a.min_points < 2 ? 2 : a.min_points;
  const double zero6[6] = {0, 0, 0, 0, 0, 0};
  {
    // offsets and sizes are wave-uniform: keep them (and the cloud base pointers) in SGPRs so
    // that every point load is `saddr + 32-bit voffset` instead of a 64-bit VGPR address
    const unsigned long long t0 = uniform64(a.toff[pair]), s0 = uniform64(a.soff[pair]);
    const unsigned long long nt64 = uniform64(a.toff[pair + 1]) - t0, ns64 = uniform64(a.soff[pair + 1]) - s0;
    if (nt64 > (unsigned long long)kBatchMaxCloud || ns64 > (unsigned long long)kBatchMaxCloud) {   // uniform; also offsets out of order
      if (tid == 0 && Cfg::kMaxPoints == 0) {
        const double z6[6] = {0, 0, 0, 0, 0, 0};
        const double p3[3] = {a.init[3 * pair], a.init[3 * pair + 1], a.init[3 * pair + 2]};
        write_result(a.out + pair, p3, z6, z6, 0.0, 0, 0, kStatusInvalid);
      }
      if (tid == 0 && Cfg::kMaxPoints > 0) a.marks[pair] = 1;       // the small variant leaves it to the large one
      return;
    }
    const int nt = (int)nt64, ns = (int)ns64;
    const float* __restrict__ tx = a.tx + t0;
    const float* __restrict__ ty = a.ty + t0;
    const float* __restrict__ sx = a.sx + s0;
    const float* __restrict__ sy = a.sy + s0;
    if (Cfg::kMaxPoints > 0) {                       // the small variant leaves big pairs to the large one
      const bool over = nt > Cfg::kMaxPoints || ns > Cfg::kMaxPoints;     // uniform
      if (tid == 0) a.marks[pair] = over ? 1 : 0;    // every pair passes here once: no memset of the marks
      if (over) return;
    } else if (!Cfg::kGlobalTables && a.marks) {     // the large variant after a small pass: marked pairs only
      if (__builtin_amdgcn_readfirstlane(a.marks[pair]) == 0) return;
    }
    double pose[3] = {a.init[3 * pair], a.init[3 * pair + 1], wrap_angle(a.init[3 * pair + 2])};
    ResultDev* out = a.out + pair;
    int iter_base = 0;
    if (a.chain) {                                   // uniform
      const int pst = __builtin_amdgcn_readfirstlane(out->status);
      if (pst != 0 && pst != 1) return;              // the coarser level's failure is the pair's result
      pose[0] = out->pose[0]; pose[1] = out->pose[1]; pose[2] = out->pose[2];
      iter_base = __builtin_amdgcn_readfirstlane(out->iterations);
      __syncthreads();                               // every wave has read out[pair] before anyone rewrites it
    }

    // ---- a1: bounding box of the target and grid geometry (oracle/ndt2d.py grid_geometry)
    {
      float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
      for_each_target_point<Cfg>(tx, ty, nt, [&](float u, float v) {
        if (isfinite(u) && isfinite(v)) {
          xmin = fminf(xmin, u); xmax = fmaxf(xmax, u);
          ymin = fminf(ymin, v); ymax = fmaxf(ymax, v);
        }
      });
      xmin = wave_min(xmin); xmax = wave_max(xmax);
      ymin = wave_min(ymin); ymax = wave_max(ymax);
      if (lane == 0) { red[wave * 4 + 0] = xmin; red[wave * 4 + 1] = xmax; red[wave * 4 + 2] = ymin; red[wave * 4 + 3] = ymax; }
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < Cfg::kWaves; ++w) {
          xmin = fminf(xmin, red[w * 4 + 0]); xmax = fmaxf(xmax, red[w * 4 + 1]);
          ymin = fminf(ymin, red[w * 4 + 2]); ymax = fmaxf(ymax, red[w * 4 + 3]);
        }
        int st = 0, W = 0, H = 0;
        float ox = 0.f, oy = 0.f;
        const float inv_c = (float)(1.0 / a.cell);
        if (!(xmin <= xmax) || !(ymin <= ymax)) {
          st = 4;                                   // no finite target point -> no valid cell
        } else {
          ox = (float)((floor((double)xmin / a.cell) - 1.0) * a.cell);
          oy = (float)((floor((double)ymin / a.cell) - 1.0) * a.cell);
          const float kx = floorf((xmax - ox) * inv_c), ky = floorf((ymax - oy) * inv_c);
          constexpr float kExtent = NG > 1 ? 3.f : 2.f;        // overlapping grids: one extra column and row
          if (!(kx >= 0.f) || !(ky >= 0.f) || (double)(kx + kExtent) * (double)(ky + kExtent) * NG > (double)Cfg::kMaxCells) {
            st = kStatusCapacity;
          } else {
            W = (int)kx + (int)kExtent; H = (int)ky + (int)kExtent;
          }
          if (NG > 1) {                                        // the origins moved down by half a cell
            reinterpret_cast<float*>(misc)[11] = (float)((floor((double)xmin / a.cell) - 1.5) * a.cell);
            reinterpret_cast<float*>(misc)[12] = (float)((floor((double)ymin / a.cell) - 1.5) * a.cell);
          }
        }
        misc[1] = W; misc[2] = H; misc[3] = st;
        reinterpret_cast<float*>(misc)[4] = ox;
        reinterpret_cast<float*>(misc)[5] = oy;
      }
      __syncthreads();
    }
    const int W = __builtin_amdgcn_readfirstlane(misc[1]), Hh = __builtin_amdgcn_readfirstlane(misc[2]);
    int status = __builtin_amdgcn_readfirstlane(misc[3]);
    const float ox = uniformf(reinterpret_cast<float*>(misc)[4]), oy = uniformf(reinterpret_cast<float*>(misc)[5]);
    // grid q's origin: (ox, oy), (ox - c/2, oy), (ox, oy - c/2), (ox - c/2, oy - c/2)
    float gxq[NG], gyq[NG];
    gxq[0] = ox; gyq[0] = oy;
    if constexpr (NG > 1) {
      const float oxs = uniformf(reinterpret_cast<float*>(misc)[11]), oys = uniformf(reinterpret_cast<float*>(misc)[12]);
      gxq[1] = oxs; gyq[1] = oy; gxq[2] = ox; gyq[2] = oys; gxq[3] = oxs; gyq[3] = oys;
    }
    const float inv_c = (float)(1.0 / a.cell);
    const float fWm1 = (float)(W - 1), fHm1 = (float)(Hh - 1);
    const int ncell1 = W * Hh;                       // cells of one grid
    const int ncell = NG * ncell1;                   // ... of the tables
    const double fix_scale = 4194304.0 / a.cell;     // 2^kFixShift / c
    static_assert(kFixShift == 22, "fix_scale literal");
    __syncthreads();                                 // misc is rewritten below
    if (status != 0) {                               // uniform
      if (tid == 0) {
        if (Cfg::kMaxPoints > 0 && status == kStatusCapacity) a.marks[pair] = 1;   // too many cells for this variant
        else if (!Cfg::kGlobalTables && Cfg::kMaxPoints == 0 && status == kStatusCapacity && a.fb_marks)
          a.fb_marks[pair] = 1;                                                    // ... for the on-chip variants: third one
        else write_result(out, pose, zero6, zero6, 0.0, iter_base, 0, status);
      }
      return;
    }

    // ---- a2 (1/2): per-cell counts
    if constexpr (Cfg::kGlobalTables) {
      // global tables: a range of kPassCells cells at a time in LDS (one pass over the target per range), stored to the
      // slab with plain stores - instead of one atomic at L2 per point
      unsigned int* pc = reinterpret_cast<unsigned int*>(smem + Cfg::kLdsPass);
      constexpr int kPassCells = 10 * Cfg::kPassSlots;                // the same buffer as 32-bit counters
#pragma unroll 1
      for (int c0 = 0; c0 < ncell; c0 += kPassCells) {
        const int np = ncell - c0 < kPassCells ? ncell - c0 : kPassCells;
        for (int k = tid; k < np; k += Cfg::kThreads) pc[k] = 0u;
        __syncthreads();
        for_each_target_point<Cfg>(tx, ty, nt, [&](float px, float py) {
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            const float fx = (px - gxq[q]) * inv_c, fy = (py - gyq[q]) * inv_c;
            if ((fx >= 1.f) & (fx < fWm1) & (fy >= 1.f) & (fy < fHm1)) {   // ring cells stay empty (in_interior)
              const int r = q * ncell1 + (int)fy * W + (int)fx - c0;
              if ((unsigned)r < (unsigned)np) atomicAdd(&pc[r], 1u);
            }
          }
        });
        __syncthreads();
        for (int k = tid; k < np; k += Cfg::kThreads) cnt[c0 + k] = pc[k];
        __syncthreads();
      }
      __threadfence();                                  // the slab's words were written by other waves of this workgroup
    } else {
      unsigned int* cnt_pairs = reinterpret_cast<unsigned int*>(idx);      // kPackedCount: two u16 counters per word
      if (Cfg::kPackedCount) {
        for (int k = tid; k < (ncell + 1) / 2; k += Cfg::kThreads) cnt_pairs[k] = 0u;
      } else {
        for (int k = tid; k < ncell; k += Cfg::kThreads) cnt[k] = 0u;
      }
      __syncthreads();
      for_each_target_point<Cfg>(tx, ty, nt, [&](float px, float py) {
        const float fx = (px - ox) * inv_c, fy = (py - oy) * inv_c;
        if ((fx >= 1.f) & (fx < fWm1) & (fy >= 1.f) & (fy < fHm1)) {   // ring cells stay empty (in_interior)
          const int key = (int)fy * W + (int)fx;
          if (Cfg::kPackedCount) atomicAdd(&cnt_pairs[key >> 1], (key & 1) ? 0x10000u : 1u);   // nt < 65536: no carry
          else atomicAdd(&cnt[key], 1u);
        }
      });
      __syncthreads();
    }
    // the count of cell k, whichever table holds it (the u16 view of the packed words is idx itself)
    auto cell_count = [&](int k) -> unsigned int { return Cfg::kPackedCount ? (unsigned int)idx[k] : cnt[k]; };

    // ---- compaction: cells with n >= min_points get a slot, in cell order (deterministic)
    const int chunk = (ncell + Cfg::kThreads - 1) / Cfg::kThreads;
    const int c0 = tid * chunk < ncell ? tid * chunk : ncell;
    const int c1 = c0 + chunk < ncell ? c0 + chunk : ncell;
    int local = 0;
    for (int k = c0; k < c1; ++k) local += (cell_count(k) >= (unsigned)minpts) ? 1 : 0;
    int nslot = 0;
    int s = block_excl_scan<Cfg>(local, s_scan, &nslot);
    nslot = __builtin_amdgcn_readfirstlane(nslot);
    if (nslot > Cfg::kMaxSlots - 1 || nslot < 1) {   // uniform (record 0 is the dummy)
      if (tid == 0) {
        if (Cfg::kMaxPoints > 0 && nslot >= 1) a.marks[pair] = 1;                    // too many occupied cells for this variant
        else if (!Cfg::kGlobalTables && Cfg::kMaxPoints == 0 && nslot >= 1 && a.fb_marks)
          a.fb_marks[pair] = 1;
        else write_result(out, pose, zero6, zero6, 0.0, iter_base, 0, nslot < 1 ? 4 : kStatusCapacity);
      }
      return;
    }
    for (int k = c0; k < c1; ++k) {
      if constexpr (NG > 1) {                            // the first slot of grids 1, 2, 3 (the slots of a grid are consecutive)
        if (k == ncell1) misc[7] = s;
        else if (k == 2 * ncell1) misc[13] = s;
        else if (k == 3 * ncell1) misc[14] = s;
      }
      const unsigned int n = cell_count(k);              // read before idx[k] is overwritten (same thread, same k)
      if (n >= (unsigned)minpts) {
        idx[k] = (IdxT)(s + 1);
        slot_n[s] = n;
        slot_key[s] = (KeyT)k;
        ++s;
      } else {
        idx[k] = 0;
      }
    }
    __syncthreads();                                 // cnt is dead; its bytes become the sums
    if constexpr (Cfg::kGlobalTables) {
      // ---- a2 (2/2), global tables: the sums of kPassSlots slots at a time in LDS, one pass over the target per range
      unsigned long long* ps = reinterpret_cast<unsigned long long*>(smem + Cfg::kLdsPass);
#pragma unroll 1
      for (int s0 = 0; s0 < nslot; s0 += Cfg::kPassSlots) {
        const int np = nslot - s0 < Cfg::kPassSlots ? nslot - s0 : Cfg::kPassSlots;
        for (int j = tid; j < 5 * np; j += Cfg::kThreads) ps[j] = 0ull;
        __syncthreads();
        for_each_target_point<Cfg>(tx, ty, nt, [&](float px, float py) {
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            const float fx = (px - gxq[q]) * inv_c, fy = (py - gyq[q]) * inv_c;
            if ((fx >= 1.f) & (fx < fWm1) & (fy >= 1.f) & (fy < fHm1)) {   // ring cells stay empty (in_interior)
              const int ix = (int)fx, iy = (int)fy;
              const int r = (int)idx[q * ncell1 + iy * W + ix] - 1 - s0;    // slot 0 (no record) gives r < 0
              if ((unsigned)r < (unsigned)np) {
                const int ux = fix_coord(px, cell_centre(gxq[q], ix, a.cell), fix_scale);
                const int uy = fix_coord(py, cell_centre(gyq[q], iy, a.cell), fix_scale);
                unsigned long long* w = ps + r;
                atomicAdd(w, (unsigned long long)(long long)ux);
                atomicAdd(w + np, (unsigned long long)(long long)uy);
                atomicAdd(w + 2 * np, prod64(ux, ux));
                atomicAdd(w + 3 * np, prod64(ux, uy));
                atomicAdd(w + 4 * np, prod64(uy, uy));
              }
            }
          }
        });
        __syncthreads();
        for (int j = tid; j < 5 * np; j += Cfg::kThreads) sums[(size_t)(j / np) * Cfg::kMaxSlots + s0 + (j % np)] = ps[j];
        __syncthreads();
      }
    } else {
      for (int j = tid; j < 5 * Cfg::kMaxSlots; j += Cfg::kThreads)
        if ((j % Cfg::kMaxSlots) < nslot) sums[j] = 0ull;
      __syncthreads();

      // ---- a2 (2/2): exact fixed-point sums per slot (LDS 64-bit integer atomics)
      for_each_target_point<Cfg>(tx, ty, nt, [&](float px, float py) {
        const float fx = (px - ox) * inv_c, fy = (py - oy) * inv_c;
        if ((fx >= 1.f) & (fx < fWm1) & (fy >= 1.f) & (fy < fHm1)) {   // ring cells stay empty (in_interior)
          const int ix = (int)fx, iy = (int)fy;
          const int slot = idx[iy * W + ix];
          if (slot) {
            const int ux = fix_coord(px, cell_centre(ox, ix, a.cell), fix_scale);
            const int uy = fix_coord(py, cell_centre(oy, iy, a.cell), fix_scale);
            unsigned long long* q = sums + (slot - 1);
            atomicAdd(q, (unsigned long long)(long long)ux);
            atomicAdd(q + Cfg::kMaxSlots, (unsigned long long)(long long)uy);
            atomicAdd(q + 2 * Cfg::kMaxSlots, prod64(ux, ux));
            atomicAdd(q + 3 * Cfg::kMaxSlots, prod64(ux, uy));
            atomicAdd(q + 4 * Cfg::kMaxSlots, prod64(uy, uy));
          }
        }
      });
    }
    if (tid == 0) misc[6] = 0;
    __syncthreads();
    if constexpr (Cfg::kGlobalTables) __threadfence();       // as above, for the sums

    // ---- a3: finalise
    if constexpr (Cfg::kGlobalTables) {
      int nvalid = 0;
      for (int sl = tid; sl < nslot; sl += Cfg::kThreads) {
        int key = (int)slot_key[sl];
        const int n = (int)slot_n[sl];
        float cox = ox, coy = oy;                      // the origin of the grid the cell belongs to
        if constexpr (NG > 1) {
          const int q = key / ncell1;
          key -= q * ncell1;
          cox = (q & 1) ? gxq[1] : ox;
          coy = (q & 2) ? gyq[2] : oy;
        }
        float4 ra, rb;
        const bool ok = n <= (int)kMaxCellCount &&
                        finalise_sums(n, (long long)sums[sl], (long long)sums[sl + Cfg::kMaxSlots],
                                      (long long)sums[sl + 2 * Cfg::kMaxSlots], (long long)sums[sl + 3 * Cfg::kMaxSlots],
                                      (long long)sums[sl + 4 * Cfg::kMaxSlots], cell_centre(cox, key % W, a.cell),
                                      cell_centre(coy, key / W, a.cell), fix_scale, a.min_points, a.eig_ratio, ra, rb);
        if (!ok) { ra = make_float4(0.f, 0.f, 0.f, 0.f); rb = make_float4(0.f, 0.f, 0.f, 0.f); }
        recA[sl + 1] = ra; recB[sl + 1] = rb;
        nvalid += ok ? 1 : 0;
      }
      if (tid == 0) { recA[0] = make_float4(0.f, 0.f, 0.f, 0.f); recB[0] = make_float4(0.f, 0.f, 0.f, 0.f); }
      if (nvalid) atomicAdd(&misc[6], nvalid);
      __syncthreads();
    } else {   // <= 3 slots per thread held in registers, then overwrite the sums
      float4 ra[Cfg::kSlotsPerThread], rb[Cfg::kSlotsPerThread];
      int nvalid = 0;
#pragma unroll
      for (int j = 0; j < Cfg::kSlotsPerThread; ++j) {
        const int sl = tid + j * Cfg::kThreads;
        ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sl < nslot) {
          const int key = slot_key[sl];
          const int n = (int)slot_n[sl];
          const bool ok = n <= (int)kMaxCellCount &&
                          finalise_sums(n, (long long)sums[sl], (long long)sums[sl + Cfg::kMaxSlots],
                                        (long long)sums[sl + 2 * Cfg::kMaxSlots],
                                        (long long)sums[sl + 3 * Cfg::kMaxSlots],
                                        (long long)sums[sl + 4 * Cfg::kMaxSlots],
                                        cell_centre(ox, key % W, a.cell), cell_centre(oy, key / W, a.cell),
                                        fix_scale, a.min_points, a.eig_ratio, ra[j], rb[j]);
          nvalid += ok ? 1 : 0;
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < Cfg::kSlotsPerThread; ++j) {
        const int sl = tid + j * Cfg::kThreads;
        if (sl < nslot) { recA[sl + 1] = ra[j]; recB[sl + 1] = rb[j]; }   // record index = idx value
      }
      if (tid == 0) { recA[0] = make_float4(0.f, 0.f, 0.f, 0.f); recB[0] = make_float4(0.f, 0.f, 0.f, 0.f); }
      if (nvalid) atomicAdd(&misc[6], nvalid);
      __syncthreads();
    }
    if (__builtin_amdgcn_readfirstlane(misc[6]) < 1) {   // uniform
      __syncthreads();
      if (tid == 0) write_result(out, pose, zero6, zero6, 0.0, iter_base, 0, 4);
      return;
    }

    // ---- a4-a8: Gauss-Newton loop, all on this CU.  Only wave 0 touches the float64 sums and
    // the solve; what the other waves need (pose, done) and what the final result needs
    // (H, g, score, counters) lives in LDS, not in registers that would be held across the
    // point loop.
    if (tid == 0) { misc[9] = 0; misc[10] = 0; }     // iter, status
    __syncthreads();
    // line-search state lives in LDS so that it is not held in registers across the point loop
    LineSearch* ls_lds = reinterpret_cast<LineSearch*>(smem + Cfg::kLdsLs);
    if (tid == 0) { ls_lds->valid = 0; ls_lds->trials = 0; }
    // Overlapping grids: the four grids TAKE TURNS on chip.  Per iteration and grid the grid's slice of the index table
    // (as u16 local slots) and its records are copied from the slab into the LDS the build's passes used, and the
    // source streams past them as in the on-chip variants - four streams of the source per iteration (L2-resident)
    // instead of three dependent gathers through L2 per point and grid (measured 56 ms per config-4 pair that way,
    // against 1.6 ms for one grid on chip).  A grid too large for that keeps the gathers.
    bool turns = false;
    unsigned short* lidx = nullptr;
    float4 *lrecA = nullptr, *lrecB = nullptr;
    if constexpr (NG > 1) {
      const int b1 = __builtin_amdgcn_readfirstlane(misc[7]), b2 = __builtin_amdgcn_readfirstlane(misc[13]),
                b3 = __builtin_amdgcn_readfirstlane(misc[14]);
      int n_max = b1;
      n_max = b2 - b1 > n_max ? b2 - b1 : n_max;
      n_max = b3 - b2 > n_max ? b3 - b2 : n_max;
      n_max = nslot - b3 > n_max ? nslot - b3 : n_max;
      const int idx_bytes = (2 * ncell1 + 15) & ~15;
      turns = n_max + 1 <= 0xffff && (long long)idx_bytes + 32ll * (n_max + 1) <= 40ll * Cfg::kPassSlots;      // uniform
      lidx = reinterpret_cast<unsigned short*>(smem + Cfg::kLdsPass);
      lrecA = reinterpret_cast<float4*>(smem + Cfg::kLdsPass + idx_bytes);
      lrecB = lrecA + (n_max + 1);
      __syncthreads();                               // (every wave has read the bases before s_scan is rewritten)
      if (tid == 0) { s_scan[0] = 0; s_scan[1] = b1; s_scan[2] = b2; s_scan[3] = b3; s_scan[4] = nslot; }
      __threadfence();                               // the records were stored to the slab by other waves of this workgroup
      __syncthreads();
    }
    for (;;) {
      float acc[kNumAcc];
      {
        double sn_d, cs_d;
        sincos_wrapped(pose[2], &sn_d, &cs_d);
        // the pose is the same in every lane: SGPRs, not 13 VGPRs held across the point loop
        const PoseF P = make_pose(uniformf((float)cs_d), uniformf((float)sn_d), uniformf((float)pose[0]),
                                  uniformf((float)pose[1]), ox, oy, inv_c, W, Hh, a.prm.d1, a.prm.d2);
        Acc2D A;
        acc_zero(A);
        // Software-pipelined source stream: two register sets (a, b) of kBatchUnroll points each;
        // while one set is consumed the other set's loads are in flight.  No register copies
        // between trips (a copy is a use and would put the wait right behind the loads).
        // Every load of a wave is 256 contiguous bytes; lanes past the end re-read the last
        // point and are masked by `live`.
        constexpr int kTrip = kBatchUnroll * Cfg::kThreads;
        float xa[kBatchUnroll], ya[kBatchUnroll], xb[kBatchUnroll], yb[kBatchUnroll];
        // Source points come through buffer descriptors: 32-bit byte offsets instead of 64-bit
        // address arithmetic, and the hardware range check returns 0 past the end of the pair's
        // slice, so no clamp either (those lanes are masked by `live` anyway).
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)sx, 0, ns * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)sy, 0, ns * 4, 0x00020000);
        auto load_set = [&](int base, float* xs, float* ys) {
#pragma unroll
          for (int u = 0; u < kBatchUnroll; ++u) {
            const int off = (base + u * Cfg::kThreads) * 4;
#if defined(NDT_BATCH_ABLATE) && (NDT_BATCH_ABLATE & 2)      // tools only: no global point loads
            xs[u] = (float)((off >> 2) & 1023) * 0.04f - 20.f;
            ys[u] = (float)(off >> 12) * 0.4f - 20.f;
#else
            xs[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
            ys[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
#endif
          }
        };
        auto consume_set = [&](int base, const float* xs, const float* ys) {
          PointRec r[kBatchUnroll];
          if constexpr (NG == 1) {
#pragma unroll
            for (int u = 0; u < kBatchUnroll; ++u)
              lookup_point_lds(P, idx, recA, recB, xs[u], ys[u], (base + u * Cfg::kThreads) < ns, r[u]);
#pragma unroll
            for (int u = 0; u < kBatchUnroll; ++u) accumulate_point<MODE>(P, r[u], A);
          } else {                                     // the same image point scores against every grid
#pragma unroll
            for (int u = 0; u < kBatchUnroll; ++u) image_point(P, xs[u], ys[u], r[u]);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
#pragma unroll
              for (int u = 0; u < kBatchUnroll; ++u) {
                const int slot = idx[q * ncell1 + image_key(P, gxq[q], gyq[q], r[u], (base + u * Cfg::kThreads) < ns)];
                r[u].A = recA[slot];
                r[u].B = recB[slot];
              }
#pragma unroll
              for (int u = 0; u < kBatchUnroll; ++u) accumulate_point<MODE>(P, r[u], A);
            }
          }
        };
        auto stream_source = [&](auto&& consume) {
          if (ns > 0) {                                // uniform; an empty source must not touch sx[-1]
            load_set(tid, xa, ya);
            for (int i = tid; i < ns; i += 2 * kTrip) {
              load_set(i + kTrip, xb, yb);
              consume(i, xa, ya);
              load_set(i + 2 * kTrip, xa, ya);
              if (i + kTrip < ns) consume(i + kTrip, xb, yb);       // wave-uniform except at the tail
            }
          }
        };
        if constexpr (NG > 1) {
          if (turns) {                                 // uniform
#pragma unroll 1
            for (int q = 0; q < NG; ++q) {
              __syncthreads();                         // the lookups of the grid before are done
              const int b0 = __builtin_amdgcn_readfirstlane(s_scan[q]), nq = __builtin_amdgcn_readfirstlane(s_scan[q + 1]) - b0;
              const IdxT* gi = idx + q * ncell1;
              for (int k = tid; k < ncell1; k += Cfg::kThreads) {
                const unsigned int v = gi[k];
                lidx[k] = (unsigned short)(v ? v - (unsigned)b0 : 0u);
              }
              for (int j = tid; j <= nq; j += Cfg::kThreads) {      // record 0: the dummy
                lrecA[j] = j ? recA[b0 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                lrecB[j] = j ? recB[b0 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
              }
              __syncthreads();
              PoseF Pq = P;
              Pq.ox = (q & 1) ? gxq[1] : ox;
              Pq.oy = (q & 2) ? gyq[2] : oy;
              stream_source([&](int base, const float* xs, const float* ys) {
                PointRec r[kBatchUnroll];
#pragma unroll
                for (int u = 0; u < kBatchUnroll; ++u)
                  lookup_point_lds(Pq, lidx, lrecA, lrecB, xs[u], ys[u], (base + u * Cfg::kThreads) < ns, r[u]);
#pragma unroll
                for (int u = 0; u < kBatchUnroll; ++u) accumulate_point<MODE>(Pq, r[u], A);
              });
            }
          } else {
            stream_source(consume_set);
          }
        } else {
          stream_source(consume_set);
        }
        acc_store(A, a.prm.d2, acc);
        acc[11] = 0.f;
      }
      if (Cfg::kSumsViaLds) {
        float* s_t = reinterpret_cast<float*>(smem + Cfg::kLdsT) + wave * ((kNumAcc - 1) * kSumRowStride);
        const float rsum = wave_reduce11_lds(acc, s_t, lane);
        if ((lane & 3) == 0 && lane < 4 * (kNumAcc - 1)) red[wave * kNumAcc + (lane >> 2)] = rsum;
      } else {
#pragma unroll
        for (int j = 0; j < kNumAcc - 1; ++j) {
          const float rsum = wave_sum_lane63(acc[j]);
          if (lane == 63) red[wave * kNumAcc + j] = rsum;
        }
      }
      __syncthreads();
      if (wave == 0) {
        // lane j < 11 sums column j over the waves in a fixed order, in float64, and parks it in
        // LDS; every lane reads the totals back (broadcast reads, one wait) - H(6) g(3) score n_hit
        // of this evaluation stay there for the result
        if (lane < kNumAcc - 1) {
          double tot = 0.0;
#pragma unroll
          for (int w = 0; w < Cfg::kWaves; ++w) tot += (double)red[w * kNumAcc + lane];
          bc[3 + lane] = tot;
        }
        __builtin_amdgcn_wave_barrier();               // same wave: LDS executes its operations in order
        double H[6], g[3];
#pragma unroll
        for (int j = 0; j < 6; ++j) H[j] = bc[3 + j];
#pragma unroll
        for (int j = 0; j < 3; ++j) g[j] = bc[9 + j];
        const int n_hit = (int)(bc[13] + 0.5);
        int iter = misc[9], st = 0;
        const bool done = gn_update(pose, H, g, n_hit, iter, st, a.prm, a.fixed_iterations, bc[12], ls_lds, ls_lds,
                                    lane == 0);
        if (lane == 0) {
          bc[0] = pose[0]; bc[1] = pose[1]; bc[2] = pose[2];
          misc[8] = done ? 1 : 0;
          misc[9] = iter;
          misc[10] = st;
        }
      }
      __syncthreads();
      pose[0] = bc[0]; pose[1] = bc[1]; pose[2] = bc[2];
      const int done = __builtin_amdgcn_readfirstlane(misc[8]);
      if (done) break;
      // No third barrier: the next iteration writes `red` (every wave) and bc/misc (wave 0) only
      // after its own first barrier, which no wave passes before all have finished the reads above;
      // wave 0 read `red` before the second barrier.
    }
    if (tid == 0)
      write_result(out, pose, &bc[3], &bc[9], bc[12], misc[9] + iter_base, (int)(bc[13] + 0.5), misc[10]);
  }
}

template <int MODE, class Cfg = BatchLarge>
__global__ __launch_bounds__(Cfg::kThreads) void k_batch(BatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* misc = reinterpret_cast<int*>(smem + Cfg::kLdsMisc);
  for (;;) {
    // dequeue one pair (every wave reaches this; the loop ends for all of them together)
    if (threadIdx.x == 0) misc[0] = (int)atomicAdd(a.queue, 1u);
    __syncthreads();
    const int pair = __builtin_amdgcn_readfirstlane(misc[0]);   // wave-uniform by construction
    __syncthreads();
    if (pair >= a.n_pairs) break;
    process_pair<MODE, Cfg>(a, pair, smem);
    __syncthreads();                                 // LDS is rewritten by the next pair
  }
}

// Third variant: the pairs the large variant handed over (fb_marks), tables in global memory.  No
// queue: workgroup b looks at pairs b, b + gridDim.x, ... (one scalar load each; with nothing handed
// over the launch costs a few microseconds).
template <int MODE, int NG = 1>
__global__ __launch_bounds__(BatchGlobal::kThreads) void k_batch_fallback(BatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Workgroup b owns the pairs b, b + gridDim.x, ...; its threads read their marks side by side (one dependent scalar
  // load per pair cost 13 us on the 8 workgroups a context starts with, with nothing marked) and process the marked
  // ones in turn.
  int* const fb_any = reinterpret_cast<int*>(smem + BatchGlobal::kLdsMisc) + 15;                     // (a word of the carve no pair uses: the carve fills the CU's LDS)
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
        process_pair<MODE, BatchGlobal, NG>(a, pair, smem);
        __syncthreads();
        __threadfence();                                  // the next pair rewrites this workgroup's slab
      }
    }
  }
}

}  // namespace ndt
