// ndt_matcher_hip.hpp - header-only C++ adapter over the C ABI (ndt_hip.h).
//
// This is the host-side mirror of the scan-matcher interface named by BASELINE.json's
// north_star ("the repo's existing scan-matcher -> GTSAM-factor interface").  The reference
// checkout shows no such interface (/root/reference/README.md:1 is its only line), so the
// class below is this repo's proposal of the smallest one a SLAM front end needs:
//   setTarget(scan or submap)  ->  align(scan, initial guess)  ->  pose + information.
// INTEGRATION.md shows how a maintainer maps it onto the real interface.
//
// No GPU code here: everything goes through extern "C" entry points of libndt_hip.so.
#ifndef NDT_MATCHER_HIP_HPP_
#define NDT_MATCHER_HIP_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "ndt_hip.h"

namespace ndt {

struct Pose2 {           // tx, ty, theta: maps source-frame points into the target frame
  double x = 0.0, y = 0.0, theta = 0.0;
};

struct MatchResult {
  Pose2 pose;
  std::array<double, 9> information{};   // row-major 3x3 Hessian of -score at the last evaluation
  std::array<double, 9> covariance{};    // calibrated pose covariance S H^-1 S (ndt2d_calibrated_covariance;
                                         // zero if the Hessian is not positive definite)
  double score = 0.0;
  int iterations = 0;
  int n_hit = 0;
  int status = NDT_OK;                   // NDT_OK / NDT_NOT_CONVERGED / ...
  bool converged() const { return status == NDT_OK; }
};

class NdtError : public std::runtime_error {
 public:
  NdtError(int32_t code, const std::string& where)
      : std::runtime_error(where + ": " + ndt_status_string(code) + " (" + std::to_string(code) + ") " +
                           ndt_last_error()),
        code_(code) {}
  int32_t code() const { return code_; }

 private:
  int32_t code_;
};

inline bool invert3(const double* H, double* C) {
  const double a = H[0], b = H[1], c = H[2], d = H[4], e = H[5], f = H[8];   // symmetric
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(det > 0.0) && !(det < 0.0)) { for (int i = 0; i < 9; ++i) C[i] = 0.0; return false; }
  const double id = 1.0 / det;
  C[0] = c00 * id; C[1] = c01 * id; C[2] = c02 * id;
  C[3] = C[1];     C[4] = (a * f - c * c) * id; C[5] = (b * c - a * e) * id;
  C[6] = C[2];     C[7] = C[5];                 C[8] = (a * d - b * b) * id;
  return true;
}

inline MatchResult to_match_result(const ndt2d_result& r, int32_t hessian_mode = NDT_HESSIAN_GAUSS_NEWTON) {
  MatchResult m;
  m.pose = {r.pose[0], r.pose[1], r.pose[2]};
  for (int i = 0; i < 9; ++i) m.information[i] = r.H[i];
  (void)ndt2d_calibrated_covariance(r.H, hessian_mode, m.covariance.data());
  m.score = r.score; m.iterations = r.iterations; m.n_hit = r.n_hit; m.status = r.status;
  return m;
}

// The calibrated covariance in the tangent frame of the measured pose, which is what a
// gtsam::BetweenFactor<Pose2> noise model expects (its error is Logmap(measured^-1 * predicted), i.e.
// translation errors expressed in the measured pose's own axes): C_local = A C A' with
// A = diag(R(theta)', 1).  MatchResult::covariance itself is for additive errors on (tx, ty, theta)
// in the target frame.
inline std::array<double, 9> covarianceInLocalFrame(const MatchResult& m) {
  const double c = std::cos(m.pose.theta), s = std::sin(m.pose.theta);
  const double A[9] = {c, s, 0.0, -s, c, 0.0, 0.0, 0.0, 1.0};
  double T[9];
  std::array<double, 9> out{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      T[3 * i + j] = 0.0;
      for (int k = 0; k < 3; ++k) T[3 * i + j] += A[3 * i + k] * m.covariance[3 * k + j];
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) out[3 * i + j] += T[3 * i + k] * A[3 * j + k];
  return out;
}

// One matcher = one device stream + one cached target grid.  Not thread-safe; use one
// instance per thread (distinct instances are independent).
class NdtMatcherHip {
 public:
  static ndt2d_params defaultParams() { ndt2d_params p; ndt2d_default_params(&p); return p; }

  explicit NdtMatcherHip(const ndt2d_params& params = defaultParams(), int device = 0) : mode_(params.hessian_mode) {
    const int32_t st = ndt2d_create(&params, device, &h_);
    if (st != NDT_OK) throw NdtError(st, "ndt2d_create");
  }
  ~NdtMatcherHip() { ndt2d_destroy(h_); }
  NdtMatcherHip(const NdtMatcherHip&) = delete;
  NdtMatcherHip& operator=(const NdtMatcherHip&) = delete;

  // (i) target grid from SoA float arrays
  void setTarget(const float* x, const float* y, size_t n) { check(ndt2d_set_target(h_, x, y, n), "ndt2d_set_target"); }
  void setTarget(const std::vector<float>& x, const std::vector<float>& y) { setTarget(x.data(), y.data(), x.size()); }
  // empty grid over the extent the submap may grow into; fill with addTargetPoints(Dev)
  void reserveTarget(double xmin, double ymin, double xmax, double ymax) {
    check(ndt2d_reserve_target(h_, xmin, ymin, xmax, ymax), "ndt2d_reserve_target");
  }
  // incremental submap update: returns the number of points outside the cached extent
  size_t addTargetPoints(const float* x, const float* y, size_t n) {
    size_t outside = 0;
    check(ndt2d_add_target_points(h_, x, y, n, &outside), "ndt2d_add_target_points");
    return outside;
  }
  // points already on the device, moved into the map frame by `pose` first (nullptr: as they are)
  size_t addTargetPointsDev(const float* d_x, const float* d_y, size_t n, const Pose2* pose = nullptr,
                            void* producer_stream = nullptr) {
    size_t outside = 0;
    const double p[3] = {pose ? pose->x : 0.0, pose ? pose->y : 0.0, pose ? pose->theta : 0.0};
    check(ndt2d_add_target_points_dev(h_, d_x, d_y, n, pose ? p : nullptr, &outside, producer_stream),
          "ndt2d_add_target_points_dev");
    return outside;
  }
  ndt2d_grid_info gridInfo() const { ndt2d_grid_info g; check(ndt2d_get_grid_info(h_, &g), "ndt2d_get_grid_info"); return g; }
  // submap persistence: the cached grid as a flat buffer (ndt_map_header + exact per-cell sums), and back - bit for
  // bit the same grid when the loading matcher has the saving one's parameters; it goes on taking points
  std::vector<unsigned char> saveMap() const {
    std::vector<unsigned char> buf(ndt2d_map_size(h_));
    check(ndt2d_save_map(h_, buf.data(), buf.size(), nullptr), "ndt2d_save_map");
    return buf;
  }
  void loadMap(const std::vector<unsigned char>& buf) { check(ndt2d_load_map(h_, buf.data(), buf.size()), "ndt2d_load_map"); }

  // (ii)+(iii)+solve: full alignment from an initial guess
  MatchResult align(const float* sx, const float* sy, size_t n, const Pose2& guess = Pose2()) {
    const double init[3] = {guess.x, guess.y, guess.theta};
    ndt2d_result r;
    check(ndt2d_align(h_, sx, sy, n, init, &r), "ndt2d_align");
    return to_match_result(r, mode_);
  }
  MatchResult align(const std::vector<float>& sx, const std::vector<float>& sy, const Pose2& guess = Pose2()) {
    return align(sx.data(), sy.data(), sx.size(), guess);
  }
  // Source scan already on the device (e.g. from ndt2d_polar_to_points_dev).  producer_stream is
  // the stream that wrote d_sx / d_sy: the handle's own stream is ordered behind it first
  // (ndt2d_wait_stream; nullptr = the legacy default stream); complete = true when those arrays are
  // known to be complete (the producer was synchronised), which skips the ordering.
  MatchResult alignDev(const float* d_sx, const float* d_sy, size_t n, const Pose2& guess, void* producer_stream,
                       bool complete = false) {
    if (!complete) check(ndt2d_wait_stream(h_, producer_stream), "ndt2d_wait_stream");
    const double init[3] = {guess.x, guess.y, guess.theta};
    ndt2d_result r;
    check(ndt2d_align_dev(h_, d_sx, d_sy, n, init, &r), "ndt2d_align_dev");
    return to_match_result(r, mode_);
  }
  // Several starts around a poor guess in one launch chain (ndt2d_align_multi_start_dev, at most 64):
  // result k is what alignDev(guesses[k]) returns; pick e.g. the best score among the converged ones.
  std::vector<MatchResult> alignMultiStartDev(const float* d_sx, const float* d_sy, size_t n, const std::vector<Pose2>& guesses,
                                              void* producer_stream, bool complete = false) {
    if (!complete) check(ndt2d_wait_stream(h_, producer_stream), "ndt2d_wait_stream");
    std::vector<double> init(3 * guesses.size());
    for (size_t k = 0; k < guesses.size(); ++k) { init[3 * k] = guesses[k].x; init[3 * k + 1] = guesses[k].y; init[3 * k + 2] = guesses[k].theta; }
    std::vector<ndt2d_result> r(guesses.size());
    check(ndt2d_align_multi_start_dev(h_, d_sx, d_sy, n, init.data(), (int32_t)guesses.size(), r.data()),
          "ndt2d_align_multi_start_dev");
    std::vector<MatchResult> out;
    for (const ndt2d_result& q : r) out.push_back(to_match_result(q, mode_));
    return out;
  }
  // Several different device scans against the cached grid in one launch chain (ndt2d_align_multi_scan_dev,
  // at most 64): scans[k] = {d_x, d_y, n}; result k is what alignDev(scans[k], guesses[k]) returns.
  struct DeviceScan { const float* x; const float* y; size_t n; };
  std::vector<MatchResult> alignMultiScanDev(const std::vector<DeviceScan>& scans, const std::vector<Pose2>& guesses,
                                             void* producer_stream, bool complete = false) {
    if (scans.size() != guesses.size() || scans.empty()) throw NdtError(NDT_ERR_INVALID_ARG, "alignMultiScanDev");
    if (!complete) check(ndt2d_wait_stream(h_, producer_stream), "ndt2d_wait_stream");
    const size_t m = scans.size();
    std::vector<const float*> px(m), py(m);
    std::vector<size_t> n(m);
    std::vector<double> init(3 * m);
    for (size_t k = 0; k < m; ++k) {
      px[k] = scans[k].x; py[k] = scans[k].y; n[k] = scans[k].n;
      init[3 * k] = guesses[k].x; init[3 * k + 1] = guesses[k].y; init[3 * k + 2] = guesses[k].theta;
    }
    std::vector<ndt2d_result> r(m);
    check(ndt2d_align_multi_scan_dev(h_, px.data(), py.data(), n.data(), init.data(), (int32_t)m, r.data()),
          "ndt2d_align_multi_scan_dev");
    std::vector<MatchResult> out;
    for (const ndt2d_result& q : r) out.push_back(to_match_result(q, mode_));
    return out;
  }
  // one evaluation at a fixed pose, for callers with their own optimiser
  ndt2d_eval evaluate(const float* sx, const float* sy, size_t n, const Pose2& at) {
    const double p[3] = {at.x, at.y, at.theta};
    ndt2d_eval e;
    check(ndt2d_evaluate(h_, sx, sy, n, p, &e), "ndt2d_evaluate");
    return e;
  }
  ndt2d_handle* raw() { return h_; }

 private:
  static void check(int32_t st, const char* where) { if (st < 0) throw NdtError(st, where); }
  ndt2d_handle* h_ = nullptr;
  int32_t mode_ = NDT_HESSIAN_GAUSS_NEWTON;   // the form of the Hessian the results carry (covariance calibration)
};

// Coarse-to-fine alignment: levels of (cell multiplier, eig_ratio)
// above the caller's finest grid widen the convergence basin; each level starts from the
// previous level's pose and the last level runs with the caller's parameters.
class NdtPyramidHip {
 public:
  struct Level { double cell_mult, eig_ratio; };
  explicit NdtPyramidHip(const ndt2d_params& fine = NdtMatcherHip::defaultParams(), int device = 0,
                         std::vector<Level> coarse = {{4.0, 0.1}, {2.0, 0.03}}) {
    for (const Level& l : coarse) {
      ndt2d_params p = fine;
      p.cell_size = fine.cell_size * l.cell_mult;
      p.eig_ratio = l.eig_ratio;
      p.eps_trans = 1e-3; p.eps_rot = 1e-4; p.max_iterations = 30; p.fixed_iterations = 0;
      p.step_max_trans = fine.step_max_trans * l.cell_mult;
      levels_.emplace_back(new NdtMatcherHip(p, device));
    }
    levels_.emplace_back(new NdtMatcherHip(fine, device));
  }
  ~NdtPyramidHip() { for (NdtMatcherHip* m : levels_) delete m; }
  NdtPyramidHip(const NdtPyramidHip&) = delete;
  NdtPyramidHip& operator=(const NdtPyramidHip&) = delete;

  void setTarget(const float* x, const float* y, size_t n) { for (NdtMatcherHip* m : levels_) m->setTarget(x, y, n); }
  MatchResult align(const float* sx, const float* sy, size_t n, const Pose2& guess = Pose2()) {
    Pose2 pose = guess;
    MatchResult r;
    int total = 0;
    for (NdtMatcherHip* m : levels_) {
      r = m->align(sx, sy, n, pose);
      total += r.iterations;
      if (r.status != NDT_OK && r.status != NDT_NOT_CONVERGED) break;
      pose = r.pose;
    }
    r.iterations = total;
    return r;
  }

 private:
  std::vector<NdtMatcherHip*> levels_;
};

namespace detail {
struct CloudView { const float* x; const float* y; size_t n; };

// concatenates the pairs into the SoA + offsets form of ndt2d_batch_align / ndt2d_multi_align
template <class Clouds, class Call>
std::vector<MatchResult> align_pairs(const Clouds& targets, const Clouds& sources, const std::vector<Pose2>& guesses,
                                     const char* what, int32_t hessian_mode, Call&& call) {
  const size_t n = targets.size();
  if (sources.size() != n || guesses.size() != n || n == 0) throw NdtError(NDT_ERR_INVALID_ARG, what);
  std::vector<uint64_t> toff(n + 1, 0), soff(n + 1, 0);
  for (size_t k = 0; k < n; ++k) { toff[k + 1] = toff[k] + targets[k].n; soff[k + 1] = soff[k] + sources[k].n; }
  std::vector<float> tx(toff[n]), ty(toff[n]), sx(soff[n]), sy(soff[n]);
  std::vector<double> init(3 * n);
  for (size_t k = 0; k < n; ++k) {
    std::copy(targets[k].x, targets[k].x + targets[k].n, tx.begin() + toff[k]);
    std::copy(targets[k].y, targets[k].y + targets[k].n, ty.begin() + toff[k]);
    std::copy(sources[k].x, sources[k].x + sources[k].n, sx.begin() + soff[k]);
    std::copy(sources[k].y, sources[k].y + sources[k].n, sy.begin() + soff[k]);
    init[3 * k] = guesses[k].x; init[3 * k + 1] = guesses[k].y; init[3 * k + 2] = guesses[k].theta;
  }
  std::vector<ndt2d_result> res(n);
  const int32_t st = call(tx.data(), ty.data(), toff.data(), sx.data(), sy.data(), soff.data(), init.data(), n, res.data());
  if (st < 0) throw NdtError(st, what);
  std::vector<MatchResult> out;
  out.reserve(n);
  for (const auto& r : res) out.push_back(to_match_result(r, hessian_mode));
  return out;
}
}  // namespace detail

// Loop-closure candidates: many independent pairs in one call.
class NdtBatchHip {
 public:
  struct Cloud { const float* x; const float* y; size_t n; };

  explicit NdtBatchHip(const ndt2d_params& params = NdtMatcherHip::defaultParams(), int device = 0)
      : mode_(params.hessian_mode) {
    const int32_t st = ndt2d_batch_create(&params, device, &b_);
    if (st != NDT_OK) throw NdtError(st, "ndt2d_batch_create");
  }
  // coarse-to-fine: `levels` ordered coarse to fine (see standardPyramid)
  NdtBatchHip(const std::vector<ndt2d_params>& levels, int device)
      : mode_(levels.empty() ? NDT_HESSIAN_GAUSS_NEWTON : levels.back().hessian_mode) {
    const int32_t st = ndt2d_batch_create_pyramid(levels.data(), static_cast<int32_t>(levels.size()), device, &b_);
    if (st != NDT_OK) throw NdtError(st, "ndt2d_batch_create_pyramid");
  }
  // the library's 3-level schedule (4c, 2c, c) around `fine`
  static std::vector<ndt2d_params> standardPyramid(const ndt2d_params& fine = NdtMatcherHip::defaultParams()) {
    std::vector<ndt2d_params> lv(3);
    const int32_t st = ndt2d_default_pyramid(&fine, lv.data());
    if (st != NDT_OK) throw NdtError(st, "ndt2d_default_pyramid");
    return lv;
  }
  ~NdtBatchHip() { ndt2d_batch_destroy(b_); }
  // NDT_TUNE_BATCH_SMALL_VARIANT / NDT_TUNE_BATCH_GLOBAL_WORKGROUPS (memory of the global-table variant against its rate)
  void setTuning(int32_t knob, int64_t value) {
    const int32_t st = ndt2d_batch_set_tuning(b_, knob, value);
    if (st != NDT_OK) throw NdtError(st, "ndt2d_batch_set_tuning");
  }
  NdtBatchHip(const NdtBatchHip&) = delete;
  NdtBatchHip& operator=(const NdtBatchHip&) = delete;

  std::vector<MatchResult> align(const std::vector<Cloud>& targets, const std::vector<Cloud>& sources,
                                 const std::vector<Pose2>& guesses) {
    return detail::align_pairs(targets, sources, guesses, "ndt2d_batch_align", mode_,
                               [&](const float* tx, const float* ty, const uint64_t* toff, const float* sx, const float* sy,
                                   const uint64_t* soff, const double* init, size_t n, ndt2d_result* res) {
                                 return ndt2d_batch_align(b_, tx, ty, toff, sx, sy, soff, init, n, res);
                               });
  }

 private:
  ndt2d_batch* b_ = nullptr;
  int32_t mode_ = NDT_HESSIAN_GAUSS_NEWTON;
};

// The same over several GPUs owned by this process: pairs are split into contiguous
// work-balanced shards, one host thread and one batch context per device.
class NdtMultiHip {
 public:
  using Cloud = NdtBatchHip::Cloud;

  // devices empty = every visible device
  explicit NdtMultiHip(const ndt2d_params& params = NdtMatcherHip::defaultParams(), const std::vector<int32_t>& devices = {})
      : mode_(params.hessian_mode) {
    const int32_t st = ndt2d_multi_create(&params, devices.empty() ? nullptr : devices.data(),
                                          static_cast<int32_t>(devices.size()), &m_);
    if (st != NDT_OK) throw NdtError(st, "ndt2d_multi_create");
  }
  NdtMultiHip(const std::vector<ndt2d_params>& levels, const std::vector<int32_t>& devices)
      : mode_(levels.empty() ? NDT_HESSIAN_GAUSS_NEWTON : levels.back().hessian_mode) {
    const int32_t st = ndt2d_multi_create_pyramid(levels.data(), static_cast<int32_t>(levels.size()),
                                                  devices.empty() ? nullptr : devices.data(),
                                                  static_cast<int32_t>(devices.size()), &m_);
    if (st != NDT_OK) throw NdtError(st, "ndt2d_multi_create_pyramid");
  }
  ~NdtMultiHip() { ndt2d_multi_destroy(m_); }
  NdtMultiHip(const NdtMultiHip&) = delete;
  NdtMultiHip& operator=(const NdtMultiHip&) = delete;

  int deviceCount() const { return ndt2d_multi_device_count(m_); }

  std::vector<MatchResult> align(const std::vector<Cloud>& targets, const std::vector<Cloud>& sources,
                                 const std::vector<Pose2>& guesses) {
    return detail::align_pairs(targets, sources, guesses, "ndt2d_multi_align", mode_,
                               [&](const float* tx, const float* ty, const uint64_t* toff, const float* sx, const float* sy,
                                   const uint64_t* soff, const double* init, size_t n, ndt2d_result* res) {
                                 return ndt2d_multi_align(m_, tx, ty, toff, sx, sy, soff, init, n, res);
                               });
  }

  // Shards already resident on their devices (one DeviceShard per context, in context order; the
  // pointers are device pointers laid out as ndt2d_batch_align_dev takes them).  Every context aligns
  // its shard, the result rows are exchanged with one RCCL all-gather on the contexts' streams
  // (ndt2d_multi_align_dev), and the rows come back in global pair order.
  struct DeviceShard {
    const float* tx = nullptr; const float* ty = nullptr; const uint64_t* toff = nullptr;
    const float* sx = nullptr; const float* sy = nullptr; const uint64_t* soff = nullptr;
    const double* init = nullptr;
    size_t n_pairs = 0;
  };
  std::vector<MatchResult> alignDev(const std::vector<DeviceShard>& shards) {
    const size_t nd = shards.size();
    std::vector<const float*> tx(nd), ty(nd), sx(nd), sy(nd);
    std::vector<const uint64_t*> toff(nd), soff(nd);
    std::vector<const double*> init(nd);
    std::vector<size_t> n(nd);
    size_t total = 0;
    for (size_t d = 0; d < nd; ++d) {
      tx[d] = shards[d].tx; ty[d] = shards[d].ty; toff[d] = shards[d].toff; sx[d] = shards[d].sx; sy[d] = shards[d].sy;
      soff[d] = shards[d].soff; init[d] = shards[d].init; n[d] = shards[d].n_pairs;
      total += n[d];
    }
    if (static_cast<int>(nd) != deviceCount()) throw NdtError(NDT_ERR_INVALID_ARG, "NdtMultiHip::alignDev: one shard per device");
    std::vector<ndt2d_result> rows(total);
    const int32_t st = ndt2d_multi_align_dev(m_, tx.data(), ty.data(), toff.data(), sx.data(), sy.data(), soff.data(),
                                             init.data(), n.data(), nullptr, nullptr, rows.data());
    if (st < 0) throw NdtError(st, "ndt2d_multi_align_dev");
    std::vector<MatchResult> out;
    out.reserve(total);
    for (const ndt2d_result& r : rows) out.push_back(to_match_result(r, mode_));
    return out;
  }

 private:
  ndt2d_multi* m_ = nullptr;
  int32_t mode_ = NDT_HESSIAN_GAUSS_NEWTON;
};

// ---- 3D (SE(3), BASELINE config 5): the same shape over the ndt3d_* entry points ----------------
struct Pose3 {           // translation and roll/pitch/yaw of R = Rz(yaw) Ry(pitch) Rx(roll)
  double x = 0.0, y = 0.0, z = 0.0, roll = 0.0, pitch = 0.0, yaw = 0.0;
};

struct MatchResult3 {
  Pose3 pose;
  std::array<double, 36> information{};  // row-major 6x6 Gauss-Newton Hessian of -score (tx ty tz roll pitch yaw)
  std::array<double, 36> covariance{};   // its inverse (zero if singular)
  double score = 0.0;
  int iterations = 0, n_hit = 0, status = NDT_OK;
  bool converged() const { return status == NDT_OK; }
};

// symmetric positive definite 6x6 inverse by Gauss-Jordan with partial pivoting; false if singular
inline bool invert6(const double* H, double* C) {
  double a[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) { a[i][j] = H[6 * i + j]; a[i][6 + j] = i == j ? 1.0 : 0.0; }
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r) if (std::abs(a[r][c]) > std::abs(a[p][c])) p = r;
    if (!(std::abs(a[p][c]) > 0.0)) { for (int i = 0; i < 36; ++i) C[i] = 0.0; return false; }
    if (p != c) for (int j = 0; j < 12; ++j) std::swap(a[p][j], a[c][j]);
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 12; ++j) a[c][j] *= inv;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      if (f != 0.0) for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) C[6 * i + j] = a[i][6 + j];
  return true;
}

// R = Rz(yaw) Ry(pitch) Rx(roll), row-major
inline std::array<double, 9> rotationOf(const Pose3& p) {
  const double ca = std::cos(p.roll), sa = std::sin(p.roll), cb = std::cos(p.pitch), sb = std::sin(p.pitch),
               cg = std::cos(p.yaw), sg = std::sin(p.yaw);
  return {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa,
          sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa,
          -sb, cb * sa, cb * ca};
}

// MatchResult3::covariance is for additive errors on (tx, ty, tz, roll, pitch, yaw).  A factor on SE(3) (e.g.
// gtsam::BetweenFactor<gtsam::Pose3>, whose tangent vector is (rotation; translation) in the measured pose's own
// frame) wants C_local = J C J' with, to first order, omega_body = E (d roll, d pitch, d yaw)',
// E = [[1, 0, -sin pitch], [0, cos roll, sin roll cos pitch], [0, -sin roll, cos roll cos pitch]], and rho = R' dt.
// Returned row-major 6x6 in the order (omega_x, omega_y, omega_z, rho_x, rho_y, rho_z).
inline std::array<double, 36> covarianceInLocalFrame3(const Pose3& pose, const std::array<double, 36>& cov) {
  const std::array<double, 9> R = rotationOf(pose);
  const double ca = std::cos(pose.roll), sa = std::sin(pose.roll), cb = std::cos(pose.pitch), sb = std::sin(pose.pitch);
  double J[36] = {0};
  const double E[9] = {1.0, 0.0, -sb, 0.0, ca, sa * cb, 0.0, -sa, ca * cb};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      J[6 * i + 3 + j] = E[3 * i + j];            // omega <- euler increments
      J[6 * (3 + i) + j] = R[3 * j + i];          // rho <- R' dt
    }
  double T[36];
  std::array<double, 36> out{};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      T[6 * i + j] = 0.0;
      for (int k = 0; k < 6; ++k) T[6 * i + j] += J[6 * i + k] * cov[6 * k + j];
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      for (int k = 0; k < 6; ++k) out[6 * i + j] += T[6 * i + k] * J[6 * j + k];
  return out;
}

class NdtMatcherHip3 {
 public:
  static ndt3d_params defaultParams() { ndt3d_params p; ndt3d_default_params(&p); return p; }

  explicit NdtMatcherHip3(const ndt3d_params& params = defaultParams(), int device = 0) {
    const int32_t st = ndt3d_create(&params, device, &h_);
    if (st != NDT_OK) throw NdtError(st, "ndt3d_create");
  }
  ~NdtMatcherHip3() { ndt3d_destroy(h_); }
  NdtMatcherHip3(const NdtMatcherHip3&) = delete;
  NdtMatcherHip3& operator=(const NdtMatcherHip3&) = delete;

  void setTarget(const float* x, const float* y, const float* z, size_t n) { check(ndt3d_set_target(h_, x, y, z, n), "ndt3d_set_target"); }
  // execution-strategy knobs of a 3D handle (NDT_TUNE_SINGLE_SYNC_BUILD)
  void setTuning(int32_t knob, int64_t value) { check(ndt3d_set_tuning(h_, knob, value), "ndt3d_set_tuning"); }
  // incremental voxel-grid update: returns the number of points outside the cached extent
  size_t addTargetPoints(const float* x, const float* y, const float* z, size_t n) {
    size_t outside = 0;
    check(ndt3d_add_target_points(h_, x, y, z, n, &outside), "ndt3d_add_target_points");
    return outside;
  }
  // empty voxel grid over a chosen box (a submap that grows scan by scan)
  void reserveTarget(const std::array<double, 3>& lo, const std::array<double, 3>& hi) {
    check(ndt3d_reserve_target(h_, lo.data(), hi.data()), "ndt3d_reserve_target");
  }
  // submap persistence, as NdtMatcherHip::saveMap / loadMap
  std::vector<unsigned char> saveMap() const {
    std::vector<unsigned char> buf(ndt3d_map_size(h_));
    check(ndt3d_save_map(h_, buf.data(), buf.size(), nullptr), "ndt3d_save_map");
    return buf;
  }
  void loadMap(const std::vector<unsigned char>& buf) { check(ndt3d_load_map(h_, buf.data(), buf.size()), "ndt3d_load_map"); }
  // device points, optionally moved into the map frame by `pose` first (the pose an alignment returned);
  // producer_stream = the stream that wrote them (nullptr: complete)
  size_t addTargetPointsDev(const float* d_x, const float* d_y, const float* d_z, size_t n, const Pose3* pose = nullptr,
                            void* producer_stream = nullptr) {
    size_t outside = 0;
    double p[6] = {0, 0, 0, 0, 0, 0};
    if (pose) { p[0] = pose->x; p[1] = pose->y; p[2] = pose->z; p[3] = pose->roll; p[4] = pose->pitch; p[5] = pose->yaw; }
    check(ndt3d_add_target_points_dev(h_, d_x, d_y, d_z, n, pose ? p : nullptr, &outside, producer_stream), "ndt3d_add_target_points_dev");
    return outside;
  }
  MatchResult3 alignDev(const float* d_sx, const float* d_sy, const float* d_sz, size_t n, const Pose3& guess = Pose3()) {
    const double init[6] = {guess.x, guess.y, guess.z, guess.roll, guess.pitch, guess.yaw};
    ndt3d_result r;
    check(ndt3d_align_dev(h_, d_sx, d_sy, d_sz, n, init, &r), "ndt3d_align_dev");
    return toMatchResult(r);
  }
  // up to 64 different device scans against the cached voxel grid in one launch chain; result k is bit for bit
  // what alignDev returns for scan k and guesses[k]
  struct DeviceScan3 { const float* x; const float* y; const float* z; size_t n; };
  std::vector<MatchResult3> alignMultiScanDev(const std::vector<DeviceScan3>& scans, const std::vector<Pose3>& guesses) {
    const size_t m = scans.size();
    if (m == 0 || guesses.size() != m) throw NdtError(NDT_ERR_INVALID_ARG, "ndt3d_align_multi_scan_dev");
    std::vector<const float*> px(m), py(m), pz(m);
    std::vector<size_t> ns(m);
    std::vector<double> init(6 * m);
    for (size_t k = 0; k < m; ++k) {
      px[k] = scans[k].x; py[k] = scans[k].y; pz[k] = scans[k].z; ns[k] = scans[k].n;
      const Pose3& g = guesses[k];
      const double p[6] = {g.x, g.y, g.z, g.roll, g.pitch, g.yaw};
      std::copy(p, p + 6, init.begin() + 6 * k);
    }
    std::vector<ndt3d_result> res(m);
    check(ndt3d_align_multi_scan_dev(h_, px.data(), py.data(), pz.data(), ns.data(), init.data(), static_cast<int32_t>(m), res.data()),
          "ndt3d_align_multi_scan_dev");
    std::vector<MatchResult3> out;
    out.reserve(m);
    for (const auto& r : res) out.push_back(toMatchResult(r));
    return out;
  }
  // one device scan from several initial poses (a lattice of hypotheses around a poor guess)
  std::vector<MatchResult3> alignMultiStartDev(const float* d_sx, const float* d_sy, const float* d_sz, size_t n,
                                               const std::vector<Pose3>& guesses) {
    std::vector<DeviceScan3> scans(guesses.size(), DeviceScan3{d_sx, d_sy, d_sz, n});
    return alignMultiScanDev(scans, guesses);
  }
  ndt3d_grid_info gridInfo() const { ndt3d_grid_info g; check(ndt3d_get_grid_info(h_, &g), "ndt3d_get_grid_info"); return g; }

  MatchResult3 align(const float* sx, const float* sy, const float* sz, size_t n, const Pose3& guess = Pose3()) {
    const double init[6] = {guess.x, guess.y, guess.z, guess.roll, guess.pitch, guess.yaw};
    ndt3d_result r;
    check(ndt3d_align(h_, sx, sy, sz, n, init, &r), "ndt3d_align");
    return toMatchResult(r);
  }
  ndt3d_handle* raw() { return h_; }
  static MatchResult3 toMatchResult(const ndt3d_result& r) {
    MatchResult3 m;
    m.pose = {r.pose[0], r.pose[1], r.pose[2], r.pose[3], r.pose[4], r.pose[5]};
    for (int i = 0; i < 36; ++i) m.information[i] = r.H[i];
    invert6(r.H, m.covariance.data());
    m.score = r.score; m.iterations = r.iterations; m.n_hit = r.n_hit; m.status = r.status;
    return m;
  }

 private:
  static void check(int32_t st, const char* where) { if (st < 0) throw NdtError(st, where); }
  ndt3d_handle* h_ = nullptr;
};

// 3D loop-closure candidates: many independent 3D pairs in one call (ndt3d_batch_*; the pair's voxel grid
// lives in LDS for its whole alignment).  `levels` coarse to fine for a pyramid, one entry otherwise.
class NdtBatchHip3 {
 public:
  struct Cloud { const float* x; const float* y; const float* z; size_t n; };

  explicit NdtBatchHip3(const ndt3d_params& params = NdtMatcherHip3::defaultParams(), int device = 0) {
    const int32_t st = ndt3d_batch_create(&params, device, &b_);
    if (st != NDT_OK) throw NdtError(st, "ndt3d_batch_create");
  }
  NdtBatchHip3(const std::vector<ndt3d_params>& levels, int device) {
    const int32_t st = ndt3d_batch_create_pyramid(levels.data(), static_cast<int32_t>(levels.size()), device, &b_);
    if (st != NDT_OK) throw NdtError(st, "ndt3d_batch_create_pyramid");
  }
  ~NdtBatchHip3() { ndt3d_batch_destroy(b_); }
  // NDT_TUNE_BATCH_GLOBAL_WORKGROUPS: 1..256 table slabs of 7.9 MB for the pairs whose voxel grid does not fit on chip
  void setTuning(int32_t knob, int64_t value) {
    const int32_t st = ndt3d_batch_set_tuning(b_, knob, value);
    if (st != NDT_OK) throw NdtError(st, "ndt3d_batch_set_tuning");
  }
  NdtBatchHip3(const NdtBatchHip3&) = delete;
  NdtBatchHip3& operator=(const NdtBatchHip3&) = delete;

  std::vector<MatchResult3> align(const std::vector<Cloud>& targets, const std::vector<Cloud>& sources,
                                  const std::vector<Pose3>& guesses) {
    const size_t n = targets.size();
    if (sources.size() != n || guesses.size() != n || n == 0) throw NdtError(NDT_ERR_INVALID_ARG, "ndt3d_batch_align");
    std::vector<uint64_t> toff(n + 1, 0), soff(n + 1, 0);
    for (size_t k = 0; k < n; ++k) { toff[k + 1] = toff[k] + targets[k].n; soff[k + 1] = soff[k] + sources[k].n; }
    std::vector<float> t[3], s[3];
    for (int c = 0; c < 3; ++c) { t[c].resize(toff[n]); s[c].resize(soff[n]); }
    std::vector<double> init(6 * n);
    for (size_t k = 0; k < n; ++k) {
      const float* tc[3] = {targets[k].x, targets[k].y, targets[k].z};
      const float* sc[3] = {sources[k].x, sources[k].y, sources[k].z};
      for (int c = 0; c < 3; ++c) {
        std::copy(tc[c], tc[c] + targets[k].n, t[c].begin() + toff[k]);
        std::copy(sc[c], sc[c] + sources[k].n, s[c].begin() + soff[k]);
      }
      const Pose3& g = guesses[k];
      const double p[6] = {g.x, g.y, g.z, g.roll, g.pitch, g.yaw};
      std::copy(p, p + 6, init.begin() + 6 * k);
    }
    std::vector<ndt3d_result> res(n);
    const int32_t st = ndt3d_batch_align(b_, t[0].data(), t[1].data(), t[2].data(), toff.data(), s[0].data(), s[1].data(),
                                         s[2].data(), soff.data(), init.data(), n, res.data());
    if (st < 0) throw NdtError(st, "ndt3d_batch_align");
    std::vector<MatchResult3> out;
    out.reserve(n);
    for (const auto& r : res) out.push_back(NdtMatcherHip3::toMatchResult(r));
    return out;
  }
  // device arrays laid out as ndt3d_batch_align_dev takes them; asynchronous on `stream` (nullptr: the context's own)
  void alignDev(const float* d_tx, const float* d_ty, const float* d_tz, const uint64_t* d_toff, const float* d_sx,
                const float* d_sy, const float* d_sz, const uint64_t* d_soff, const double* d_init, size_t n_pairs,
                ndt3d_result* d_results, void* stream = nullptr) {
    const int32_t st = ndt3d_batch_align_dev(b_, d_tx, d_ty, d_tz, d_toff, d_sx, d_sy, d_sz, d_soff, d_init, n_pairs, d_results, stream);
    if (st < 0) throw NdtError(st, "ndt3d_batch_align_dev");
  }
  void* stream() { return ndt3d_batch_stream(b_); }

 private:
  ndt3d_batch* b_ = nullptr;
};

// The 3D candidates over several GPUs owned by this process (ndt3d_multi_*): host pairs split into contiguous
// work-balanced shards, one context and one host thread per device.
class NdtMultiHip3 {
 public:
  explicit NdtMultiHip3(const ndt3d_params& params = NdtMatcherHip3::defaultParams(), const std::vector<int32_t>& devices = {}) {
    const int32_t st = ndt3d_multi_create(&params, devices.empty() ? nullptr : devices.data(), static_cast<int32_t>(devices.size()), &m_);
    if (st != NDT_OK) throw NdtError(st, "ndt3d_multi_create");
  }
  ~NdtMultiHip3() { ndt3d_multi_destroy(m_); }
  NdtMultiHip3(const NdtMultiHip3&) = delete;
  NdtMultiHip3& operator=(const NdtMultiHip3&) = delete;
  int deviceCount() const { return ndt3d_multi_device_count(m_); }
  ndt3d_multi* raw() { return m_; }     // ndt3d_multi_align / ndt3d_multi_align_dev (RCCL gather) take it

 private:
  ndt3d_multi* m_ = nullptr;
};

}  // namespace ndt
#endif  // NDT_MATCHER_HIP_HPP_
