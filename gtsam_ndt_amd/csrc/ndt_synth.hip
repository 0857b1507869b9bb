// Device-side generator of the synthetic workloads (include/ndt_synth.h): the HIP twin of
// gtsam_ndt_amd/synth.py.  Workload generator only - nothing of the matcher lives here, and the
// matcher library does not link it.
//
// Bit-for-bit rule (synth.py's "design rule"): after the integer RNG only exactly-rounded IEEE
// float64 operations (+ - * / sqrt, conversions, comparisons) in the order synth.py writes them.
// This file is compiled with -ffp-contract=off (host and device), float64 division and sqrt are
// correctly rounded on gfx950 as compiled by hipcc, and the one libm call of the generator (cos/sin
// of a pair's rotation) is made on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ndt_synth.h"

namespace {

std::string& synth_error() {
  thread_local std::string e;
  return e;
}
#define SYNTH_TRY(expr)                                                        \
  do {                                                                         \
    const hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) {                                                    \
      synth_error() = std::string(#expr) + ": " + hipGetErrorString(_e);       \
      (void)hipGetLastError();                                                 \
      return -3;                                                               \
    }                                                                          \
  } while (0)

constexpr int kSeg = NDT_SYNTH_ROOM_SEGMENTS;
constexpr double kSqrt3 = 1.7320508075688772;   // synth._SQRT3

__host__ __device__ inline unsigned long long splitmix64(unsigned long long seed, unsigned long long counter) {
  unsigned long long z = (counter + 1ull) * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline double uniform01(unsigned long long seed, unsigned long long counter) {
  return (double)(splitmix64(seed, counter) >> 11) * (1.0 / 9007199254740992.0);
}

// synth.room_scene, statement by statement
void room_scene(unsigned long long seed, double L, double x0, double y0, double* seg /*[76][4]*/) {
  int n = 0;
  auto put = [&](double x1, double y1, double x2, double y2) {
    seg[4 * n] = x1; seg[4 * n + 1] = y1; seg[4 * n + 2] = x2; seg[4 * n + 3] = y2;
    ++n;
  };
  put(x0, y0, x0 + L, y0);
  put(x0 + L, y0, x0 + L, y0 + L);
  put(x0 + L, y0 + L, x0, y0 + L);
  put(x0, y0 + L, x0, y0);
  unsigned long long k = 0;
  const double margin = 0.12 * L;
  const double span = L - 2.0 * margin;
  const double ext = 0.2 * L;
  const double hmin = 0.006 * L;
  const double hspan = 0.024 * L;
  for (int i = 0; i < 24; ++i) {
    const double cx = x0 + margin + uniform01(seed, k) * span; ++k;
    const double cy = y0 + margin + uniform01(seed, k) * span; ++k;
    const double dx = (uniform01(seed, k) - 0.5) * ext; ++k;
    const double dy = (uniform01(seed, k) - 0.5) * ext; ++k;
    put(cx - dx, cy - dy, cx + dx, cy + dy);
  }
  for (int b = 0; b < 12; ++b) {
    const double cx = x0 + margin + uniform01(seed, k) * span; ++k;
    const double cy = y0 + margin + uniform01(seed, k) * span; ++k;
    const double hx = hmin + uniform01(seed, k) * hspan; ++k;
    const double hy = hmin + uniform01(seed, k) * hspan; ++k;
    if (b % 2 == 0) {
      put(cx - hx, cy - hy, cx + hx, cy - hy);
      put(cx + hx, cy - hy, cx + hx, cy + hy);
      put(cx + hx, cy + hy, cx - hx, cy + hy);
      put(cx - hx, cy + hy, cx - hx, cy - hy);
    } else {
      put(cx - hx, cy, cx, cy - hy);
      put(cx, cy - hy, cx + hx, cy);
      put(cx + hx, cy, cx, cy + hy);
      put(cx, cy + hy, cx - hx, cy);
    }
  }
}

// what the sampling kernel needs of one scene: SoA segments, lengths, running length (synth._cumlen)
struct SceneDev {
  const double* ax; const double* ay; const double* bx; const double* by;   // [n_seg]
  const double* ln;                                                          // [n_seg]
  const double* cum;                                                         // [n_seg + 1]
  int n_seg;
};

// host arrays of one scene in the order [ax | ay | bx | by | ln | cum]: 5 n + (n + 1) doubles
void pack_scene(const double* seg, int n, double* out) {
  double* ax = out; double* ay = out + n; double* bx = out + 2 * n; double* by = out + 3 * n;
  double* ln = out + 4 * n; double* cum = out + 5 * n;
  double acc = 0.0;
  cum[0] = 0.0;
  for (int i = 0; i < n; ++i) {
    ax[i] = seg[4 * i]; ay[i] = seg[4 * i + 1]; bx[i] = seg[4 * i + 2]; by[i] = seg[4 * i + 3];
    const double dx = bx[i] - ax[i], dy = by[i] - ay[i];
    ln[i] = std::sqrt(dx * dx + dy * dy);
    acc = acc + ln[i];
    cum[i + 1] = acc;
  }
}
constexpr size_t scene_doubles(int n) { return 6 * (size_t)n + 1; }

struct Frame {          // synth.to_source_frame: applied when on != 0
  double tx, ty, c, s;
  int on;
};

// synth.sample_scene for point `i` (+ to_source_frame), rounded to float32
__device__ __forceinline__ void sample_point(const SceneDev& sc, unsigned long long seed, unsigned long long i,
                                             double noise /* sqrt3 * sigma */, const Frame& f, float* ox, float* oy) {
  const unsigned long long c0 = i * 16ull;
  const double total = sc.cum[sc.n_seg];
  const double s = uniform01(seed, c0) * total;
  // np.searchsorted(cum, s, side="right") - 1: the number of entries <= s, minus one; clipped
  int lo = 0, hi = sc.n_seg + 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sc.cum[mid] <= s) lo = mid + 1; else hi = mid;
  }
  int k = lo - 1;
  k = k < 0 ? 0 : (k > sc.n_seg - 1 ? sc.n_seg - 1 : k);
  const double t = (s - sc.cum[k]) / sc.ln[k];
  double x = sc.ax[k] + t * (sc.bx[k] - sc.ax[k]);
  double y = sc.ay[k] + t * (sc.by[k] - sc.ay[k]);
  const double nx = ((uniform01(seed, c0 + 1) + uniform01(seed, c0 + 2)) + (uniform01(seed, c0 + 3) + uniform01(seed, c0 + 4))) - 2.0;
  const double ny = ((uniform01(seed, c0 + 5) + uniform01(seed, c0 + 6)) + (uniform01(seed, c0 + 7) + uniform01(seed, c0 + 8))) - 2.0;
  x = x + nx * noise;
  y = y + ny * noise;
  if (f.on) {
    const double dx = x - f.tx, dy = y - f.ty;
    const double xs = f.c * dx + f.s * dy;
    const double ys = (-f.s) * dx + f.c * dy;
    x = xs; y = ys;
  }
  *ox = (float)x;
  *oy = (float)y;
}

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void k_sample(SceneDev sc, size_t n, unsigned long long seed, unsigned long long first,
                                                      double noise, Frame f, float* __restrict__ x, float* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads)
    sample_point(sc, seed, first + i, noise, f, &x[i], &y[i]);
}

// config 4: one scene per pair (blockIdx.y), staged in LDS; thread i makes target point i and
// source point i of its pair
struct PairDev {
  unsigned long long seed_t, seed_s;
  Frame frame;
};

__global__ __launch_bounds__(kThreads) void k_config4(const double* __restrict__ scenes /*[n_pairs][scene_doubles(76)]*/,
                                                       const PairDev* __restrict__ pairs, size_t n_tgt, size_t n_src,
                                                       double noise, float* __restrict__ tx, float* __restrict__ ty,
                                                       float* __restrict__ sx, float* __restrict__ sy) {
  __shared__ double s_scene[scene_doubles(kSeg)];
  const size_t pair = blockIdx.y;
  const double* src = scenes + pair * scene_doubles(kSeg);
  for (int j = threadIdx.x; j < (int)scene_doubles(kSeg); j += kThreads) s_scene[j] = src[j];
  __syncthreads();
  SceneDev sc;
  sc.ax = s_scene; sc.ay = s_scene + kSeg; sc.bx = s_scene + 2 * kSeg; sc.by = s_scene + 3 * kSeg;
  sc.ln = s_scene + 4 * kSeg; sc.cum = s_scene + 5 * kSeg; sc.n_seg = kSeg;
  const PairDev p = pairs[pair];
  Frame none{};
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n_tgt; i += stride)
    sample_point(sc, p.seed_t, i, noise, none, &tx[pair * n_tgt + i], &ty[pair * n_tgt + i]);
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n_src; i += stride)
    sample_point(sc, p.seed_s, i, noise, p.frame, &sx[pair * n_src + i], &sy[pair * n_src + i]);
}

__global__ void k_offsets(unsigned long long* toff, unsigned long long* soff, size_t n_pairs, size_t n_tgt, size_t n_src) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k <= n_pairs) { toff[k] = k * n_tgt; soff[k] = k * n_src; }
}

// keeps an upload's pinned source alive until the copy has left it
struct Staging {
  void* host = nullptr;
  void* dev = nullptr;
  ~Staging() {
    if (host) (void)hipHostFree(host);
    if (dev) (void)hipFree(dev);
  }
};

}  // namespace

// ---- 3D: synth3d.lidar_scan on the device (ray cast of the box room; workload generator for config 5) ----------
constexpr int kMaxBoxes = NDT_SYNTH_MAX_BOXES;
struct Lidar3dArgs {
  double lo[kMaxBoxes][3], hi[kMaxBoxes][3];     // clutter boxes
  double rlo[3], rhi[3];                         // the room
  double o[3];                                   // ray origin (sensor position in the map frame)
  double R[9];                                   // sensor orientation
  double el0, el_step, az_step, amp;             // beam pattern; amp = sqrt(3) * sigma
  unsigned long long seed;
  int n_box, n_elev, n_azim;
  int firing_order;   // 0: ring by ring (ring-major); 1: all beams of a bearing, then the next bearing (as a driver delivers them)
};
__global__ __launch_bounds__(256) void k_lidar3d(Lidar3dArgs a, float* __restrict__ x, float* __restrict__ y,
                                                 float* __restrict__ z) {
  const int n = a.n_elev * a.n_azim;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int e = i / a.n_azim, j = i - e * a.n_azim;
    // np.deg2rad(np.linspace(-24, 20, n_elev)): start + e * step (the last beam is the stop value itself)
    const double el_deg = e == a.n_elev - 1 && a.n_elev > 1 ? 20.0 : a.el0 + (double)e * a.el_step;
    const double el = el_deg * (3.141592653589793 / 180.0);
    const double az = ((double)j + 0.5) * a.az_step;
    double se, ce, sa, ca;
    sincos(el, &se, &ce);
    sincos(az, &sa, &ca);
    const double ds[3] = {ce * ca, ce * sa, se};
    double t_hit = INFINITY, inv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double d = a.R[3 * c] * ds[0] + a.R[3 * c + 1] * ds[1] + a.R[3 * c + 2] * ds[2];
      if (fabs(d) < 1e-12) d = 1e-12;
      inv[c] = 1.0 / d;
      t_hit = fmin(t_hit, fmax((a.rlo[c] - a.o[c]) * inv[c], (a.rhi[c] - a.o[c]) * inv[c]));   // inside the room: exit distance
    }
    for (int b = 0; b < a.n_box; ++b) {
      double tn = -INFINITY, tf = INFINITY;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double t1 = (a.lo[b][c] - a.o[c]) * inv[c], t2 = (a.hi[b][c] - a.o[c]) * inv[c];
        tn = fmax(tn, fmin(t1, t2));
        tf = fmin(tf, fmax(t1, t2));
      }
      if (tn <= tf && tn > 0.0 && tn < t_hit) t_hit = tn;
    }
    const unsigned long long k = 4ull * (unsigned long long)i;
    const double noise = (uniform01(a.seed, k) + uniform01(a.seed, k + 1) + uniform01(a.seed, k + 2) + uniform01(a.seed, k + 3) - 2.0) * a.amp;
    const double r = t_hit + noise;
    const int o = a.firing_order ? j * a.n_elev + e : i;          // the same points either way: only their order differs
    x[o] = (float)(ds[0] * r); y[o] = (float)(ds[1] * r); z[o] = (float)(ds[2] * r);
  }
}


extern "C" {

const char* ndt_synth_last_error(void) { return synth_error().c_str(); }

int32_t ndt_synth_room_scene(uint64_t seed, double L, double x0, double y0, double* segments) {
  if (!segments || !(L > 0.0)) return -1;
  room_scene(seed, L, x0, y0, segments);
  return 0;
}

int32_t ndt_synth_sample_dev(const double* segments, int32_t n_seg, size_t n, uint64_t seed, double sigma, uint64_t first,
                             const double* pose, const double* cs_sn, float* d_x, float* d_y, void* stream) {
  if (!segments || n_seg < 1 || n == 0 || !d_x || !d_y || (pose && !cs_sn)) return -1;
  hipStream_t st = (hipStream_t)stream;
  Staging s;
  const size_t nd = scene_doubles(n_seg);
  SYNTH_TRY(hipHostMalloc(&s.host, nd * sizeof(double), hipHostMallocDefault));
  SYNTH_TRY(hipMalloc(&s.dev, nd * sizeof(double)));
  pack_scene(segments, n_seg, (double*)s.host);
  SYNTH_TRY(hipMemcpyAsync(s.dev, s.host, nd * sizeof(double), hipMemcpyHostToDevice, st));
  const double* d = (const double*)s.dev;
  SceneDev sc{d, d + n_seg, d + 2 * (size_t)n_seg, d + 3 * (size_t)n_seg, d + 4 * (size_t)n_seg, d + 5 * (size_t)n_seg, n_seg};
  Frame f{};
  if (pose) { f.tx = pose[0]; f.ty = pose[1]; f.c = cs_sn[0]; f.s = cs_sn[1]; f.on = 1; }
  size_t blocks = (n + kThreads - 1) / kThreads;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_sample, dim3((unsigned)blocks), dim3(kThreads), 0, st, sc, n, (unsigned long long)seed,
                     (unsigned long long)first, kSqrt3 * sigma, f, d_x, d_y);
  SYNTH_TRY(hipGetLastError());
  SYNTH_TRY(hipStreamSynchronize(st));      // the scene's staging buffers are released on return
  return 0;
}

int32_t ndt_synth_config4_dev(uint64_t first_pair, size_t n_pairs, size_t n_tgt, size_t n_src, double sigma, float* d_tx,
                              float* d_ty, float* d_sx, float* d_sy, uint64_t* d_toff, uint64_t* d_soff, double* d_init,
                              double* d_pose, void* stream) {
  if (n_pairs == 0 || n_pairs > 65535 || n_tgt == 0 || n_src == 0 || !d_tx || !d_ty || !d_sx || !d_sy || !d_toff || !d_soff ||
      !d_init)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  const size_t nd = scene_doubles(kSeg);
  const size_t bytes_scene = n_pairs * nd * sizeof(double), bytes_pair = n_pairs * sizeof(PairDev),
               bytes_pose = n_pairs * 3 * sizeof(double);
  Staging s;
  SYNTH_TRY(hipHostMalloc(&s.host, bytes_scene + bytes_pair + bytes_pose, hipHostMallocDefault));
  SYNTH_TRY(hipMalloc(&s.dev, bytes_scene + bytes_pair));
  double* h_scene = (double*)s.host;
  PairDev* h_pair = (PairDev*)((char*)s.host + bytes_scene);
  double* h_pose = (double*)((char*)s.host + bytes_scene + bytes_pair);
  const double L = 50.0;
  for (size_t j = 0; j < n_pairs; ++j) {
    // synth.make_pair(4, pair_index = k): scene seed 9000 + k, offset from stream 7000 + k
    const unsigned long long k = first_pair + j, S = 9000ull + k;
    double seg[4 * kSeg];
    room_scene(S, L, -0.5 * L, -0.5 * L, seg);
    pack_scene(seg, kSeg, h_scene + j * nd);
    const double e0 = (uniform01(7000ull + k, 0) - 0.5) * 0.2;
    const double e1 = (uniform01(7000ull + k, 1) - 0.5) * 0.2;
    const double e2 = (uniform01(7000ull + k, 2) - 0.5) * 0.02;
    PairDev& p = h_pair[j];
    p.seed_t = S * 7919ull + 11ull;
    p.seed_s = S * 7919ull + 12ull;
    p.frame.tx = 0.0 + e0; p.frame.ty = 0.0 + e1;           // sensor at the origin
    p.frame.c = std::cos(e2); p.frame.s = std::sin(e2);     // the host's libm, as math.cos / math.sin
    p.frame.on = 1;
    h_pose[3 * j] = p.frame.tx; h_pose[3 * j + 1] = p.frame.ty; h_pose[3 * j + 2] = e2;
  }
  SYNTH_TRY(hipMemcpyAsync(s.dev, s.host, bytes_scene + bytes_pair, hipMemcpyHostToDevice, st));
  if (d_pose) SYNTH_TRY(hipMemcpyAsync(d_pose, h_pose, bytes_pose, hipMemcpyHostToDevice, st));
  SYNTH_TRY(hipMemsetAsync(d_init, 0, bytes_pose, st));      // init = (sensor, 0) = (0, 0, 0)
  hipLaunchKernelGGL(k_offsets, dim3((unsigned)((n_pairs + 256) / 256)), dim3(256), 0, st, (unsigned long long*)d_toff,
                     (unsigned long long*)d_soff, n_pairs, n_tgt, n_src);
  const size_t most = n_tgt > n_src ? n_tgt : n_src;
  size_t bx = (most + kThreads - 1) / kThreads;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_config4, dim3((unsigned)bx, (unsigned)n_pairs), dim3(kThreads), 0, st, (const double*)s.dev,
                     (const PairDev*)((char*)s.dev + bytes_scene), n_tgt, n_src, kSqrt3 * sigma, d_tx, d_ty, d_sx, d_sy);
  SYNTH_TRY(hipGetLastError());
  SYNTH_TRY(hipStreamSynchronize(st));      // staging is released on return
  return 0;
}

int32_t ndt_synth_lidar3d_dev(const double* boxes_lo, const double* boxes_hi, int32_t n_box, double L, double height,
                              double sensor_z, uint64_t seed, const double pose[6], int32_t n_elev, int32_t n_azim,
                              double sigma, int32_t firing_order, float* d_x, float* d_y, float* d_z, void* stream) {
  if (!boxes_lo || !boxes_hi || n_box < 0 || n_box > kMaxBoxes || !pose || n_elev < 1 || n_azim < 1 || !d_x || !d_y || !d_z ||
      !(L > 0.0) || !(height > 0.0))
    return -1;
  Lidar3dArgs a{};
  for (int b = 0; b < n_box; ++b)
    for (int c = 0; c < 3; ++c) { a.lo[b][c] = boxes_lo[3 * b + c]; a.hi[b][c] = boxes_hi[3 * b + c]; }
  a.rlo[0] = -0.5 * L; a.rlo[1] = -0.5 * L; a.rlo[2] = 0.0;
  a.rhi[0] = 0.5 * L; a.rhi[1] = 0.5 * L; a.rhi[2] = height;
  a.o[0] = pose[0]; a.o[1] = pose[1]; a.o[2] = pose[2] + sensor_z;
  const double ca = std::cos(pose[3]), sa = std::sin(pose[3]), cb = std::cos(pose[4]), sb = std::sin(pose[4]),
               cg = std::cos(pose[5]), sg = std::sin(pose[5]);
  const double R[9] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa,
                       sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa,
                       -sb, cb * sa, cb * ca};
  for (int j = 0; j < 9; ++j) a.R[j] = R[j];
  a.el0 = -24.0;
  a.el_step = n_elev > 1 ? 44.0 / (double)(n_elev - 1) : 0.0;
  a.az_step = 2.0 * 3.141592653589793 / (double)n_azim;
  a.amp = kSqrt3 * sigma;
  a.seed = seed;
  a.n_box = n_box; a.n_elev = n_elev; a.n_azim = n_azim;
  a.firing_order = firing_order ? 1 : 0;
  const size_t n = (size_t)n_elev * (size_t)n_azim;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_lidar3d, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, d_x, d_y, d_z);
  SYNTH_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
