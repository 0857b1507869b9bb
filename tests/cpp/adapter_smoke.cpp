// C++ adapter smoke test: reads two scans (raw float32 files written by the pytest driver),
// aligns them through ndt::NdtMatcherHip / ndt::NdtBatchHip and prints the result as JSON-ish
// text that the driver compares with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndt_matcher_hip.hpp"

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
    ndt::NdtBatchHip b;
    const ndt::NdtBatchHip::Cloud t{tx.data(), ty.data(), tx.size()}, s{sx.data(), sy.data(), sx.size()};
    const auto rs = b.align({t, t}, {s, s}, {guess, guess});
    std::printf("batch %.17g %.17g %.17g %d %d\n", rs[1].pose.x, rs[1].pose.y, rs[1].pose.theta, rs[1].iterations, rs[1].status);
    ndt::NdtBatchHip bp(ndt::NdtBatchHip::standardPyramid(), 0);
    const auto rp = bp.align({t}, {s}, {ndt::Pose2{guess.x + 0.5, guess.y - 0.4, guess.theta + 0.04}});
    std::printf("pyramid %.17g %.17g %.17g %d %d\n", rp[0].pose.x, rp[0].pose.y, rp[0].pose.theta, rp[0].iterations, rp[0].status);
    ndt::NdtMultiHip mm(ndt::NdtMatcherHip::defaultParams(), {0, 0});   // two contexts on device 0
    const auto rm = mm.align({t, t, t}, {s, s, s}, {guess, guess, guess});
    std::printf("multi %.17g %.17g %.17g %d %d\n", rm[2].pose.x, rm[2].pose.y, rm[2].pose.theta, rm[2].iterations, rm[2].status);
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
    }
  } catch (const ndt::NdtError& e) {
    std::printf("error %d %s\n", e.code(), e.what());
    return e.code() == NDT_ERR_NO_DEVICE ? 3 : 1;
  }
  return 0;
}
