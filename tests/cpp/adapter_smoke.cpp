// C++ adapter smoke test: reads two scans (raw float32 files written by the pytest driver),
// aligns them through ndt::NdtMatcherHip / ndt::NdtBatchHip and prints the result as JSON-ish
// text that the driver compares with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ndt_matcher_hip.hpp"

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
    double prod = 0.0;   // information * covariance ~ identity
    for (int k = 0; k < 3; ++k) prod += r.information[k] * r.covariance[3 * k];
    std::printf("infocov %.6f\n", prod);
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
  } catch (const ndt::NdtError& e) {
    std::printf("error %d %s\n", e.code(), e.what());
    return e.code() == NDT_ERR_NO_DEVICE ? 3 : 1;
  }
  return 0;
}
