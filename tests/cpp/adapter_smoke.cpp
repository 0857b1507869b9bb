// C++ adapter smoke test: reads two scans (raw float32 files written by the pytest driver),
// aligns them through ndt::NdtMatcherHip / ndt::NdtBatchHip and prints the result as JSON-ish
// text that the driver compares with the oracle.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndt_matcher_hip.hpp"

// Device-pointer entry points: the program owns a few device buffers itself.  Only the C API of the HIP
// runtime is needed for that (no device code here: this file is compiled by plain g++).
extern "C" {
int hipMalloc(void** ptr, size_t size);
int hipFree(void* ptr);
int hipMemcpy(void* dst, const void* src, size_t size, int kind);   // kind 1 = host to device
int hipDeviceSynchronize(void);
}
template <class T>
static T* to_device(const std::vector<T>& v) {
  void* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T)) != 0 || hipMemcpy(d, v.data(), v.size() * sizeof(T), 1) != 0) {
    std::fprintf(stderr, "device upload failed\n");
    std::exit(2);
  }
  return static_cast<T*>(d);
}

// A box room sampled on a lattice with a deterministic jitter, and the same surfaces seen from a
// frame displaced by a known rigid motion: the 3D adapter must recover that motion.
static void room3d(std::vector<float>& x, std::vector<float>& y, std::vector<float>& z, unsigned seed) {
  auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
  const float L = 12.f, H = 4.f, step = 0.11f;
  for (float a = -L; a <= L; a += step)
    for (float b = -L; b <= L; b += step) {
      x.push_back(a + 0.03f * rnd()); y.push_back(b + 0.03f * rnd()); z.push_back(0.f + 0.03f * rnd());      // floor
      if (std::fabs(a) < 6.f && std::fabs(b) < 6.f) { x.push_back(a); y.push_back(b); z.push_back(H + 0.03f * rnd()); }
    }
  for (float a = -L; a <= L; a += step)
    for (float c = 0.f; c <= H; c += step) {
      x.push_back(a); y.push_back(-L + 0.03f * rnd()); z.push_back(c);
      x.push_back(L + 0.03f * rnd()); y.push_back(a); z.push_back(c);
      x.push_back(3.f + 0.03f * rnd()); y.push_back(0.25f * a); z.push_back(0.5f * c);                       // an inner wall
    }
}

static std::vector<float> load(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f) / 4;
  std::fseek(f, 0, SEEK_SET);
  std::vector<float> v(n);
  if (std::fread(v.data(), 4, n, f) != (size_t)n) std::exit(2);
  std::fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 8) { std::fprintf(stderr, "usage: tx ty sx sy ix iy itheta\n"); return 2; }
  const auto tx = load(argv[1]), ty = load(argv[2]), sx = load(argv[3]), sy = load(argv[4]);
  const ndt::Pose2 guess{std::atof(argv[5]), std::atof(argv[6]), std::atof(argv[7])};
  try {
    ndt::NdtMatcherHip m;
    m.setTarget(tx, ty);
    const ndt::MatchResult r = m.align(sx, sy, guess);
    std::printf("single %.17g %.17g %.17g %d %d %d\n", r.pose.x, r.pose.y, r.pose.theta, r.iterations, r.n_hit, r.status);
    // covariance = S H^-1 S: row 0 of (S^-1 H S^-1) times column 0 of the covariance is 1
    const double sc[3] = {std::sqrt(NDT_COV_SCALE_GN_TRANS), std::sqrt(NDT_COV_SCALE_GN_TRANS), std::sqrt(NDT_COV_SCALE_GN_ROT)};
    double prod = 0.0;
    for (int k = 0; k < 3; ++k) prod += r.information[k] / (sc[0] * sc[k]) * r.covariance[3 * k];
    std::printf("infocov %.6f\n", prod);
    const auto loc = ndt::covarianceInLocalFrame(r);   // a rotation of the translation block: trace and rotation variance unchanged
    std::printf("localcov %.6g %.6g\n", (loc[0] + loc[4]) / (r.covariance[0] + r.covariance[4]), loc[8] / r.covariance[8]);
    {   // the submap through a buffer into a second matcher: the same alignment, bit for bit
      ndt::NdtMatcherHip m2;
      m2.loadMap(m.saveMap());
      const ndt::MatchResult r2 = m2.align(sx, sy, guess);
      std::printf("maprt %d\n", (r2.pose.x == r.pose.x && r2.pose.y == r.pose.y && r2.pose.theta == r.pose.theta &&
                                 r2.iterations == r.iterations && r2.information == r.information) ? 1 : 0);
    }
    ndt::NdtBatchHip b;
    const ndt::NdtBatchHip::Cloud t{tx.data(), ty.data(), tx.size()}, s{sx.data(), sy.data(), sx.size()};
    const auto rs = b.align({t, t}, {s, s}, {guess, guess});
    std::printf("batch %.17g %.17g %.17g %d %d\n", rs[1].pose.x, rs[1].pose.y, rs[1].pose.theta, rs[1].iterations, rs[1].status);
    ndt::NdtBatchHip bp(ndt::NdtBatchHip::standardPyramid(), 0);
    const auto rp = bp.align({t}, {s}, {ndt::Pose2{guess.x + 0.5, guess.y - 0.4, guess.theta + 0.04}});
    std::printf("pyramid %.17g %.17g %.17g %d %d\n", rp[0].pose.x, rp[0].pose.y, rp[0].pose.theta, rp[0].iterations, rp[0].status);
    {   // Biber's four overlapping grids: the single-pair path and the batch path with the same option
      ndt2d_params po = ndt::NdtMatcherHip::defaultParams();
      po.overlap_grids = 4;
      ndt::NdtMatcherHip mo(po);
      mo.setTarget(tx, ty);
      const ndt::MatchResult ro = mo.align(sx, sy, guess);
      ndt::NdtBatchHip bo(po);
      const auto rbo = bo.align({t}, {s}, {guess});
      std::printf("overlap %.17g %.17g %.17g %d %d %.17g %.17g %.17g %d %d\n", ro.pose.x, ro.pose.y, ro.pose.theta, ro.iterations,
                  ro.status, rbo[0].pose.x, rbo[0].pose.y, rbo[0].pose.theta, rbo[0].iterations, rbo[0].status);
    }
    ndt::NdtMultiHip mm(ndt::NdtMatcherHip::defaultParams(), {0, 0});   // two contexts on device 0
    const auto rm = mm.align({t, t, t}, {s, s, s}, {guess, guess, guess});
    std::printf("multi %.17g %.17g %.17g %d %d\n", rm[2].pose.x, rm[2].pose.y, rm[2].pose.theta, rm[2].iterations, rm[2].status);
    {
      // the device-pointer forms: the scan uploaded once, aligned where it lies (the upload above is complete:
      // hipMemcpy is synchronous), several starts in one chain, and the multi-device context with the RCCL
      // gather (one device here: a one-rank communicator)
      float* d_sx = to_device(sx); float* d_sy = to_device(sy);
      float* d_tx = to_device(tx); float* d_ty = to_device(ty);
      const ndt::MatchResult rd = m.alignDev(d_sx, d_sy, sx.size(), guess, nullptr, /*complete=*/true);
      std::printf("dev %.17g %.17g %.17g %d %d\n", rd.pose.x, rd.pose.y, rd.pose.theta, rd.iterations, rd.status);
      const std::vector<ndt::Pose2> starts = {guess, ndt::Pose2{guess.x + 0.02, guess.y - 0.01, guess.theta + 0.001},
                                              ndt::Pose2{guess.x - 0.02, guess.y + 0.02, guess.theta - 0.002}};
      const auto rmulti = m.alignMultiStartDev(d_sx, d_sy, sx.size(), starts, nullptr, true);
      std::printf("multistart %.17g %.17g %.17g %d %d %d\n", rmulti[0].pose.x, rmulti[0].pose.y, rmulti[0].pose.theta,
                  rmulti[0].iterations, rmulti[0].status, (int)rmulti.size());
      const auto rscan = m.alignMultiScanDev({{d_sx, d_sy, sx.size()}, {d_sx, d_sy, sx.size() - 100}}, {guess, guess}, nullptr, true);
      std::printf("multiscan %.17g %.17g %.17g %d %d\n", rscan[0].pose.x, rscan[0].pose.y, rscan[0].pose.theta,
                  rscan[0].iterations, rscan[0].status);
      const std::vector<uint64_t> toff = {0, tx.size(), 2 * tx.size()}, soff = {0, sx.size(), 2 * sx.size()};
      std::vector<float> tx2(tx), ty2(ty), sx2(sx), sy2(sy);
      tx2.insert(tx2.end(), tx.begin(), tx.end()); ty2.insert(ty2.end(), ty.begin(), ty.end());
      sx2.insert(sx2.end(), sx.begin(), sx.end()); sy2.insert(sy2.end(), sy.begin(), sy.end());
      const std::vector<double> init2 = {guess.x, guess.y, guess.theta, guess.x, guess.y, guess.theta};
      ndt::NdtMultiHip one(ndt::NdtMatcherHip::defaultParams(), {0});
      ndt::NdtMultiHip::DeviceShard sh;
      sh.tx = to_device(tx2); sh.ty = to_device(ty2); sh.toff = to_device(toff);
      sh.sx = to_device(sx2); sh.sy = to_device(sy2); sh.soff = to_device(soff);
      sh.init = to_device(init2); sh.n_pairs = 2;
      const auto rg = one.alignDev({sh});
      std::printf("rccl %.17g %.17g %.17g %d %d\n", rg[1].pose.x, rg[1].pose.y, rg[1].pose.theta, rg[1].iterations, rg[1].status);
      (void)hipDeviceSynchronize();
      void* bufs[] = {d_sx, d_sy, d_tx, d_ty, (void*)sh.tx, (void*)sh.ty, (void*)sh.toff, (void*)sh.sx, (void*)sh.sy, (void*)sh.soff, (void*)sh.init};
      for (void* b2 : bufs) (void)hipFree(b2);
    }
    {
      std::vector<float> x, y, z, qx, qy, qz;
      room3d(x, y, z, 1u);
      room3d(qx, qy, qz, 7u);                                    // an independent sample of the same surfaces
      const double t[3] = {0.20, -0.15, 0.05}, yaw = 0.02;      // target = R(yaw) source + t
      const double c = std::cos(yaw), s3 = std::sin(yaw);
      for (size_t i = 0; i < qx.size(); ++i) {                   // source = R^-1 (world - t)
        const double wx = qx[i] - t[0], wy = qy[i] - t[1], wz = qz[i] - t[2];
        qx[i] = (float)(c * wx + s3 * wy); qy[i] = (float)(-s3 * wx + c * wy); qz[i] = (float)wz;
      }
      ndt::NdtMatcherHip3 m3;
      m3.setTarget(x.data(), y.data(), z.data(), x.size() / 2);
      m3.addTargetPoints(x.data() + x.size() / 2, y.data() + x.size() / 2, z.data() + x.size() / 2, x.size() - x.size() / 2);
      const ndt::MatchResult3 r3 = m3.align(qx.data(), qy.data(), qz.data(), qx.size());
      double ic = 0.0;                                           // information * covariance ~ identity
      for (int k = 0; k < 6; ++k) ic += r3.information[k] * r3.covariance[6 * k];
      std::printf("three_d %.9g %.9g %.9g %.9g %.9g %.9g %d %d %.6f\n", r3.pose.x, r3.pose.y, r3.pose.z, r3.pose.roll,
                  r3.pose.pitch, r3.pose.yaw, r3.iterations, r3.status, ic);
      {
        // covarianceInLocalFrame3: its Jacobian against a finite perturbation of the pose - the body-frame rotation
        // vector of R0' R1 and R0' (t1 - t0) for a small additive step d on (t, roll, pitch, yaw)
        ndt::Pose3 p0; p0.x = 0.3; p0.y = -0.2; p0.z = 0.1; p0.roll = 0.2; p0.pitch = -0.3; p0.yaw = 0.7;
        const double d[6] = {1e-6, -2e-6, 1.5e-6, 2e-6, -1e-6, 1.2e-6};
        ndt::Pose3 p1 = p0; p1.x += d[0]; p1.y += d[1]; p1.z += d[2]; p1.roll += d[3]; p1.pitch += d[4]; p1.yaw += d[5];
        const auto R0 = ndt::rotationOf(p0), R1 = ndt::rotationOf(p1);
        double M[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { M[3 * i + j] = 0; for (int k = 0; k < 3; ++k) M[3 * i + j] += R0[3 * k + i] * R1[3 * k + j]; }
        const double want[6] = {0.5 * (M[7] - M[5]), 0.5 * (M[2] - M[6]), 0.5 * (M[3] - M[1]),
                                R0[0] * d[0] + R0[3] * d[1] + R0[6] * d[2], R0[1] * d[0] + R0[4] * d[1] + R0[7] * d[2],
                                R0[2] * d[0] + R0[5] * d[1] + R0[8] * d[2]};
        std::array<double, 36> dd{};                         // covariance of a deterministic step: d d'
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) dd[6 * i + j] = d[i] * d[j];
        const auto loc3 = ndt::covarianceInLocalFrame3(p0, dd);
        double worst = 0.0;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) worst = std::max(worst, std::fabs(loc3[6 * i + j] - want[i] * want[j]));
        std::printf("localcov3 %.3g\n", worst / 1e-12);       // relative to |d|^2
      }
      // the same pair twice through the 3D batch (one of them from a displaced guess): both must land on the single-pair pose
      ndt::NdtBatchHip3 b3;
      b3.setTuning(NDT_TUNE_BATCH_GLOBAL_WORKGROUPS, 8);      // 63 MB of table slabs instead of 2 GB: the results do not depend on it
      const ndt::NdtBatchHip3::Cloud tc{x.data(), y.data(), z.data(), x.size()}, sc{qx.data(), qy.data(), qz.data(), qx.size()};
      ndt::Pose3 off; off.x = 0.02; off.y = -0.02; off.yaw = 0.002;
      const std::vector<ndt::MatchResult3> rb = b3.align({tc, tc}, {sc, sc}, {ndt::Pose3(), off});
      for (size_t k = 0; k < rb.size(); ++k)
        std::printf("three_d_batch%zu %.9g %.9g %.9g %.9g %.9g %.9g %d %d\n", k, rb[k].pose.x, rb[k].pose.y, rb[k].pose.z,
                    rb[k].pose.roll, rb[k].pose.pitch, rb[k].pose.yaw, rb[k].iterations, rb[k].status);
    }
  } catch (const ndt::NdtError& e) {
    std::printf("error %d %s\n", e.code(), e.what());
    return e.code() == NDT_ERR_NO_DEVICE ? 3 : 1;
  }
  return 0;
}
