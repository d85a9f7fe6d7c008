// What the reference's LM loop sees from C++ (no Python in the way): wall time of blocking
// linearize / computeCost calls on a HIP cost and of a full LevenbergMarquadtDynamic::minimize.
//   bench_blocking [N = 10000000]   |   bench_blocking camera
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "moptimizer_amd/cost_function_hip.hpp"
#include "moptimizer_caller/levenberg_marquadt.hpp"
#include "moptimizer_amd/so3.hpp"

namespace mh = moptimizer::hip;
using Clock = std::chrono::steady_clock;

// BASELINE config 5 from C++: 100 000 reprojection elements as two costs (40 k + 60 k), Geman-McClure,
// forward differences; the optimizer's loop over its costs (levenberg_marquadt_dyn.cpp:48-60), unlinked
// and linked.   bench_blocking camera
static int cameraSteps() {
  const int n = 100000, split = 40000;
  std::vector<double> pts(size_t(n) * 4);
  std::vector<std::int32_t> pix(size_t(n) * 2);
  std::mt19937_64 gen(17);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  for (int i = 0; i < n; ++i) {
    // points in front of the camera of tst/camera_calibration.cpp:22-30 (depth along the laser's x;
    // the frame maps (X, Y, Z) to the camera's (-Y, -Z, X)) and the pixels they land on at x = 0
    const double X = 1.5 + 2.5 * uni(gen), Y = 2 * uni(gen) - 1, Z = -0.5 + 1.3 * uni(gen);
    pts[4 * i + 0] = X; pts[4 * i + 1] = Y; pts[4 * i + 2] = Z; pts[4 * i + 3] = 1.0;
    pix[2 * i + 0] = std::int32_t(std::lround(586.122314453125 * (-Y) / X + 638.8477694496105));
    pix[2 * i + 1] = std::int32_t(std::lround(722.3973388671875 * (-Z) / X + 323.031267074588));
  }
  try {
    auto first = std::make_shared<mh::ReprojectionDeviceModel>(pts.data(), pix.data(), size_t(split));
    auto rest = std::make_shared<mh::ReprojectionDeviceModel>(pts.data() + 4 * size_t(split),
                                                              pix.data() + 2 * size_t(split), size_t(n - split));
    mh::CostFunctionNumericalDynamic<double> a(first, 6, 2, split), b(rest, 6, 2, n - split);
    auto loss = std::make_shared<moptimizer::loss::GemmanMCClure<double>>(100.0);
    a.setLossFunction(loss);
    b.setLossFunction(loss);
    double x[6] = {0, 0, 0, 0, 0, 0}, H[36], g[6], Hs[36], gs[6];
    auto step = [&](int k) {
      x[0] = 1e-4 * (k % 16);
      double y = 0.0;
      for (int q = 0; q < 36; ++q) Hs[q] = 0.0;
      for (int q = 0; q < 6; ++q) gs[q] = 0.0;
      for (auto *c : {&a, &b}) {
        y += c->linearize(x, H, g);
        for (int q = 0; q < 36; ++q) Hs[q] += H[q];
        for (int q = 0; q < 6; ++q) gs[q] += g[q];
      }
      return y;
    };
    for (int linked = 0; linked < 2; ++linked) {
      if (linked) mh::linkCosts<double>({&a, &b});
      for (int warm = 0; warm < 100; ++warm) step(warm);
      const int reps = 2000;
      const auto t0 = Clock::now();
      double y = 0.0;
      for (int k = 0; k < reps; ++k) y = step(k);
      const double us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / reps;
      std::printf("camera (40 k + 60 k elements, Geman-McClure, forward differences), costs %s: %.1f us per step "
                  "(both costs linearized and summed), sum r^T r = %.6e\n", linked ? "linked  " : "unlinked", us, y);
    }
    mh::linkCosts<double>({});
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "camera") return cameraSteps();
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
