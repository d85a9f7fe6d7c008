// What the reference's LM loop sees from C++ (no Python in the way): wall time of blocking
// linearize / computeCost calls on a HIP cost and of a full LevenbergMarquadtDynamic::minimize.
//   bench_blocking [N = 10000000]
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "moptimizer_amd/cost_function_hip.hpp"
#include "moptimizer_caller/levenberg_marquadt.hpp"
#include "moptimizer_amd/so3.hpp"

namespace mh = moptimizer::hip;
using Clock = std::chrono::steady_clock;

int main(int argc, char **argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 10000000L;
  std::vector<double> src(size_t(n) * 3), tgt(size_t(n) * 3);
  std::mt19937_64 gen(42);
  std::uniform_real_distribution<double> uni(0.0, 10.0);
  std::normal_distribution<double> noise(0.0, 0.01);
  const double xt[6] = {10.5, 10.2, 0.1, 0.38994502377414, 0.31542006718654, 0.54962215934141};
  double T[16];
  moptimizer::so3::convert6DOFParameterToMatrix<double>(xt, T);
  for (long i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) src[3 * i + k] = uni(gen);
    for (int r = 0; r < 3; ++r)
      tgt[3 * i + r] = T[0 * 4 + r] * src[3 * i] + T[1 * 4 + r] * src[3 * i + 1] +
                       T[2 * 4 + r] * src[3 * i + 2] + T[3 * 4 + r] + noise(gen);
  }
  try {
    auto t0 = Clock::now();
    auto model = std::make_shared<mh::Point2PointDeviceModel<double>>(src.data(), tgt.data(), size_t(n));
    mh::CostFunctionNumericalDynamic<double> cost(model, 6, 3, int(n));
    std::printf("N = %ld: cost construction (PCIe copy + re-layout) %.2f ms\n", n,
                std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    mopt_cost_set_speculation(cost.handle(), 0);  // time real sweeps
    double x[6] = {0.5, -0.3, 0.2, 0.1, -0.2, 0.3}, H[36], b[6];
    for (int warm = 0; warm < 50; ++warm) cost.linearize(x, H, b);  // code objects loaded, clocks up
    const int reps = 300;
    t0 = Clock::now();
    for (int k = 0; k < reps; ++k) {
      x[0] = 0.5 + 1e-4 * (k % 16);
      cost.linearize(x, H, b);
    }
    const double lin_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / reps;
    t0 = Clock::now();
    for (int k = 0; k < reps; ++k) {
      x[0] = 0.5 + 1e-4 * (k % 16);
      cost.computeCost(x);
    }
    const double cost_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / reps;
    std::printf("blocking linearize (forward differences): %.1f us/call = %.3e correspondences/s\n", lin_us,
                n / (lin_us * 1e-6));
    std::printf("blocking computeCost:                     %.1f us/call\n", cost_us);

    for (int spec = 0; spec < 2; ++spec) {
      mopt_cost_set_speculation(cost.handle(), spec);
      moptimizer::LevenbergMarquadtDynamic<double> lm(6);
      lm.setMaximumIterations(50);
      lm.addCost(&cost);
      double x0[6] = {0, 0, 0, 0, 0, 0};
      std::int64_t s0 = 0, h0 = 0, s1 = 0, h1 = 0;
      mopt_cost_stats(cost.handle(), &s0, &h0);
      t0 = Clock::now();
      const auto status = lm.minimize(x0);
      const double ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
      mopt_cost_stats(cost.handle(), &s1, &h1);
      double e = 0;
      for (int i = 0; i < 6; ++i) e = std::max(e, std::fabs(x0[i] - xt[i]));
      std::printf("LM minimize (speculation %s): status %d, %u outer iterations, %lld sweeps, %.2f ms, "
                  "|x - truth| = %.1e\n", spec ? "on " : "off", int(status), lm.getExecutedIterations(),
                  (long long)(s1 - s0), ms, e);
    }
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
  return 0;
}
