// Time to solution of the reference's registration problem (tst/point2point.cpp:192-217: LM from
// x0 = 0 to the fixture pose with the numerical cost), same LM loop, two cost implementations:
//   CPU  oracle::CostFunctionNumericalDynamic  (restatement of the reference cost, its linearize
//        single-threaded like the original; its cost-only sweep on one worker per 4096 elements, at
//        most every core, a single worker on the calling thread — the reference's TBB pool does not
//        start threads per call, so neither may the baseline; best of three solves up to 100 k)
//   HIP  moptimizer::hip::CostFunctionNumericalDynamic (construction = PCIe copy included), driven
//        (a) by that same host loop through the boundary, (b) by the device-resident loop
//        (moptimizer::hip::LevenbergMarquadtDevice = mopt_lm_minimize)
//   bench_solve [N ...]     default 30000 1000000 10000000
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "moptimizer_amd/cost_function_hip.hpp"
#include "moptimizer_amd/levenberg_marquadt_device.hpp"
#include "moptimizer_caller/levenberg_marquadt.hpp"
#include "moptimizer_amd/so3.hpp"

#include "cpu_costs.hpp"
#include "test_models.hpp"

namespace mh = moptimizer::hip;
using Clock = std::chrono::steady_clock;
static double msSince(Clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}

int main(int argc, char **argv) {
  std::vector<long> sizes;
  for (int a = 1; a < argc; ++a) sizes.push_back(std::atol(argv[a]));
  if (sizes.empty()) sizes = {30000L, 1000000L, 10000000L};
  const double xt[6] = {10.5, 10.2, 0.1, 0.38994502377414, 0.31542006718654, 0.54962215934141};
  double T[16];
  moptimizer::so3::convert6DOFParameterToMatrix<double>(xt, T);
  try {
    {  // HIP context creation is a per-process cost, not a per-solve one
      double p[6] = {0, 0, 0, 1, 1, 1};
      auto warm = std::make_shared<mh::Point2PointDeviceModel<double>>(p, p + 3, size_t(1));
      mh::CostFunctionNumericalDynamic<double> cost(warm, 6, 3, 1);
      double x[6] = {0, 0, 0, 0, 0, 0}, H[36], b[6];
      for (int k = 0; k < 50; ++k) cost.linearize(x, H, b);
    }
    std::printf("| N | CPU solve ms (iterations) | HIP construct ms | HIP solve ms, host loop (iterations, sweeps) | "
                "HIP solve ms, device-resident loop (iterations, sweeps) | us per sweep host / device | "
                "CPU cost-sweep workers | max |x_cpu - x_hip| host / device loop |\n"
                "|---|---|---|---|---|---|---|---|\n");
    for (long n : sizes) {
      std::vector<double> src(size_t(n) * 3), tgt(size_t(n) * 3);
      std::mt19937_64 gen(42);
      std::uniform_real_distribution<double> uni(0.0, 10.0);
      std::normal_distribution<double> noise(0.0, 0.01);
      for (long i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) src[3 * i + k] = uni(gen);
        for (int r = 0; r < 3; ++r)
          tgt[3 * i + r] = T[0 * 4 + r] * src[3 * i] + T[1 * 4 + r] * src[3 * i + 1] +
                           T[2 * 4 + r] * src[3 * i + 2] + T[3 * 4 + r] + noise(gen);
      }
      double x_cpu[6] = {0, 0, 0, 0, 0, 0}, x_hip[6] = {0, 0, 0, 0, 0, 0}, x_dev[6] = {0, 0, 0, 0, 0, 0};
      unsigned it_cpu = 0, it_hip = 0, it_dev = 0;
      double cpu_ms = 0, build_ms = 0, hip_ms = 0, dev_ms = 1e30;
      long long sweeps = 0, dev_sweeps = 0;
      {
        auto model = std::make_shared<oracle::Point2Point<double>>(src.data(), tgt.data());
        oracle::CostFunctionNumericalDynamic<double> cost(model, 6, 3, int(n));
        moptimizer::LevenbergMarquadtDynamic<double> lm(6);
        lm.setMaximumIterations(50);
        lm.addCost(&cost);
        cpu_ms = 1e30;
        for (int rep = 0; rep < (n <= 100000 ? 3 : 1); ++rep) {
          for (double &v : x_cpu) v = 0.0;
          const auto t0 = Clock::now();
          lm.minimize(x_cpu);
          cpu_ms = std::min(cpu_ms, msSince(t0));
        }
        it_cpu = lm.getExecutedIterations();
      }
      {
        auto t0 = Clock::now();
        auto model = std::make_shared<mh::Point2PointDeviceModel<double>>(src.data(), tgt.data(), size_t(n));
        mh::CostFunctionNumericalDynamic<double> cost(model, 6, 3, int(n));
        build_ms = msSince(t0);
        moptimizer::LevenbergMarquadtDynamic<double> lm(6);
        lm.setMaximumIterations(50);
        lm.addCost(&cost);
        std::int64_t s0 = 0, s1 = 0, h = 0;
        mopt_cost_stats(cost.handle(), &s0, &h);
        t0 = Clock::now();
        lm.minimize(x_hip);
        hip_ms = msSince(t0);
        mopt_cost_stats(cost.handle(), &s1, &h);
        sweeps = s1 - s0;
        it_hip = lm.getExecutedIterations();
        // the same cost under the device-resident loop; best of a few solves from x0 = 0 (the first
        // one also uploads the cost's static constants)
        mh::LevenbergMarquadtDevice<double> dlm(6);
        dlm.setMaximumIterations(50);
        dlm.addCost(&cost);
        for (int rep = 0; rep < 5; ++rep) {
          for (double &v : x_dev) v = 0.0;
          t0 = Clock::now();
          dlm.minimize(x_dev);
          dev_ms = std::min(dev_ms, msSince(t0));
        }
        dev_sweeps = dlm.sweeps();
        it_dev = dlm.getExecutedIterations();
      }
      double diff = 0, diff_dev = 0;
      for (int i = 0; i < 6; ++i) diff = std::max(diff, std::fabs(x_cpu[i] - x_hip[i]));
      for (int i = 0; i < 6; ++i) diff_dev = std::max(diff_dev, std::fabs(x_cpu[i] - x_dev[i]));
      const long hw = long(std::thread::hardware_concurrency());
      const long by_grain = (n + oracle::CostComputation<double>::kParallelGrain - 1) /
                            oracle::CostComputation<double>::kParallelGrain;
      std::printf("| %ld | %.3f (%u) | %.2f | %.3f (%u, %lld) | %.3f (%u, %lld) | %.1f / %.1f | %ld | %.1e / %.1e |\n",
                  n, cpu_ms, it_cpu, build_ms, hip_ms, it_hip, sweeps, dev_ms, it_dev, dev_sweeps,
                  1e3 * hip_ms / double(sweeps), 1e3 * dev_ms / double(dev_sweeps),
                  std::max(1L, std::min(hw, by_grain)), diff, diff_dev);
      std::fflush(stdout);
    }
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
  return 0;
}
