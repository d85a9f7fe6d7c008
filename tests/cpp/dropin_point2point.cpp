// Drop-in check, C++ side: the reference's point2point tests (tst/point2point.cpp:142-217) with
// the HIP-backed cost classes in place of the CPU ones, driven by the unchanged LM loop.
//
//   ConsistencyOverCostsClasses  — HIP analytic (as-written layout) and HIP numeric linearize vs
//                                  the CPU restatement, 1e-6 relative; costs equal across classes
//   Optimization                 — LevenbergMarquadtDynamic + CostFunctionNumericalHip converges
//                                  to the fixture pose (t, log R); same pose as the CPU cost
//   Covariance / loss            — setCovariance / setLossFunction reach the device
//   Camera calibration           — tst/camera_calibration.cpp:101-122 with the HIP cost
//
// Input: a raw little-endian float64 file of packed xyz (the façade cloud), argv[1].
// The oracle headers are used here as the checker only (this is a test program).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "cpu_costs.hpp"     // oracle (test infrastructure)
#include "test_models.hpp"   // oracle (test infrastructure)
#include "moptimizer_amd/cost_function_hip.hpp"
#include "moptimizer_caller/levenberg_marquadt.hpp"
#include "moptimizer_amd/so3.hpp"

using Scalar = double;
namespace mh = moptimizer::hip;

static int g_fail = 0;
static void expectTrue(bool ok, const char *what, double a = 0, double b = 0) {
  std::printf("%s %s (%.12g vs %.12g)\n", ok ? "PASS" : "FAIL", what, a, b);
  if (!ok) ++g_fail;
}
static double relErr(const Scalar *a, const Scalar *b, int n) {
  double scale = 0, err = 0;
  for (int i = 0; i < n; ++i) {
    scale = std::max(scale, std::fabs(double(b[i])));
    err = std::max(err, std::fabs(double(a[i]) - double(b[i])));
  }
  return err / (scale > 0 ? scale : 1.0);
}

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s facade_xyz.f64\n", argv[0]);
    return 2;
  }
  std::vector<Scalar> src;
  {
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) {
      std::perror("open");
      return 2;
    }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    src.resize(bytes / sizeof(Scalar));
    if (std::fread(src.data(), 1, bytes, f) != size_t(bytes)) return 2;
    std::fclose(f);
  }
  const int n = int(src.size() / 3);
  // tst/point2point.cpp:93-123: Rx(0.3) Ry(0.4) Rz(0.5), t = (10.5, 10.2, 0.1)
  std::vector<Scalar> tgt(src.size());
  {
    const double a = 0.3, b = 0.4, c = 0.5;
    const double Rx[3][3] = {{1, 0, 0}, {0, std::cos(a), -std::sin(a)}, {0, std::sin(a), std::cos(a)}};
    const double Ry[3][3] = {{std::cos(b), 0, std::sin(b)}, {0, 1, 0}, {-std::sin(b), 0, std::cos(b)}};
    const double Rz[3][3] = {{std::cos(c), -std::sin(c), 0}, {std::sin(c), std::cos(c), 0}, {0, 0, 1}};
    double Rxy[3][3], R[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        Rxy[i][j] = 0;
        for (int k = 0; k < 3; ++k) Rxy[i][j] += Rx[i][k] * Ry[k][j];
      }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        R[i][j] = 0;
        for (int k = 0; k < 3; ++k) R[i][j] += Rxy[i][k] * Rz[k][j];
      }
    const double t[3] = {10.5, 10.2, 0.1};
    for (int i = 0; i < n; ++i)
      for (int r = 0; r < 3; ++r)
        tgt[3 * i + r] = ((R[r][0] * src[3 * i] + R[r][1] * src[3 * i + 1]) + R[r][2] * src[3 * i + 2]) + t[r];
  }
  std::printf("Loaded : %d points.\n", n);

  try {
    // ---- ConsistencyOverCostsClasses ----------------------------------------------------
    double x0[6] = {0};
    auto cpu_model = std::make_shared<oracle::Point2Point<Scalar>>(src.data(), tgt.data());
    auto gpu_model = std::make_shared<mh::Point2PointDeviceModel<Scalar>>(src.data(), tgt.data(), n);
    oracle::CostFunctionAnalyticalDynamic<Scalar> cost_an_cpu(cpu_model, 6, 3, n);
    oracle::CostFunctionNumericalDynamic<Scalar> cost_num_cpu(cpu_model, 6, 3, n);
    mh::CostFunctionAnalyticalTstLayoutHip<Scalar> cost_an_gpu(gpu_model, 6, 3, n);
    mh::CostFunctionNumericalHip<Scalar> cost_num_gpu(gpu_model, 6, 3, n);

    Scalar H_an_c[36], H_an_g[36], H_nu_c[36], H_nu_g[36], b_c[6], b_g[6], b2_c[6], b2_g[6];
    const Scalar s_an_c = cost_an_cpu.linearize(x0, H_an_c, b_c);
    const Scalar s_an_g = cost_an_gpu.linearize(x0, H_an_g, b_g);
    const Scalar s_nu_c = cost_num_cpu.linearize(x0, H_nu_c, b2_c);
    const Scalar s_nu_g = cost_num_gpu.linearize(x0, H_nu_g, b2_g);
    expectTrue(std::fabs(s_an_g - s_an_c) <= 1e-6 * s_an_c, "sum analytic gpu==cpu", s_an_g, s_an_c);
    expectTrue(std::fabs(s_nu_g - s_an_g) <= 1e-6 * s_an_c, "sum numeric==analytic (gpu)", s_nu_g, s_an_g);
    expectTrue(std::fabs(s_nu_g - s_nu_c) <= 1e-6 * s_an_c, "sum numeric gpu==cpu", s_nu_g, s_nu_c);
    expectTrue(relErr(H_an_g, H_an_c, 36) <= 1e-6, "H analytic(tst layout) gpu==cpu", relErr(H_an_g, H_an_c, 36), 1e-6);
    expectTrue(relErr(b_g, b_c, 6) <= 1e-6, "b analytic(tst layout) gpu==cpu", relErr(b_g, b_c, 6), 1e-6);
    expectTrue(relErr(H_nu_g, H_nu_c, 36) <= 1e-6, "H numeric gpu==cpu", relErr(H_nu_g, H_nu_c, 36), 1e-6);
    expectTrue(relErr(b2_g, b2_c, 6) <= 1e-6, "b numeric gpu==cpu", relErr(b2_g, b2_c, 6), 1e-6);
    const Scalar c_g = cost_num_gpu.computeCost(x0), c_c = cost_num_cpu.computeCost(x0);
    expectTrue(std::fabs(c_g - c_c) <= 1e-6 * c_c, "computeCost gpu==cpu", c_g, c_c);

    // ---- Optimization --------------------------------------------------------------------
    const double truth[6] = {10.5, 10.2, 0.1, 0.38994502377414, 0.31542006718654, 0.54962215934141};
    double xg[6] = {0}, xc[6] = {0};
    {
      moptimizer::LevenbergMarquadtDynamic<Scalar> lm(6);
      lm.setMaximumIterations(50);
      lm.addCost(&cost_num_gpu);
      const auto st = lm.minimize(xg);
      lm.clearCosts();
      std::printf("GPU LM status %d after %u iterations: %.10f %.10f %.10f %.10f %.10f %.10f\n", int(st),
                  lm.getExecutedIterations(), xg[0], xg[1], xg[2], xg[3], xg[4], xg[5]);
      lm.addCost(&cost_num_cpu);
      lm.minimize(xc);
    }
    double e_truth = 0, e_cpu = 0;
    for (int i = 0; i < 6; ++i) {
      e_truth = std::max(e_truth, std::fabs(xg[i] - truth[i]));
      e_cpu = std::max(e_cpu, std::fabs(xg[i] - xc[i]));
    }
    expectTrue(e_truth < 1e-6, "LM(HIP numeric cost) reaches the fixture pose", e_truth, 1e-6);
    expectTrue(e_cpu < 1e-6, "LM(HIP) pose == LM(CPU) pose", e_cpu, 1e-6);

    // ---- covariance + loss reach the device ------------------------------------------------
    {
      auto cov = std::make_shared<moptimizer::covariance::Matrix<Scalar>>();
      cov->resize(3, 3);
      cov->setIdentity();
      *cov *= 0.5;
      (*cov)(0, 1) = (*cov)(1, 0) = 0.1;
      mh::CostFunctionAnalyticalHip<Scalar> g(gpu_model, 6, 3, n);
      oracle::CostFunctionAnalyticalDynamic<Scalar> c(
          std::make_shared<oracle::Point2Point<Scalar>>(src.data(), tgt.data(),
                                                        oracle::P2PJacobianLayout::kRowMajor),
          6, 3, n);
      g.setCovariance(cov);
      c.setCovariance(cov);
      g.setLossFunction(std::make_shared<moptimizer::loss::GemmanMCClure<Scalar>>(100.0));
      c.setLossFunction(std::make_shared<moptimizer::loss::GemmanMCClure<Scalar>>(100.0));
      Scalar Hg[36], Hc[36], bg[6], bc[6];
      double x1[6] = {0.5, -0.3, 0.2, 0.1, -0.2, 0.3};
      g.linearize(x1, Hg, bg);
      c.linearize(x1, Hc, bc);
      expectTrue(relErr(Hg, Hc, 36) <= 1e-6, "H with covariance + Geman-McClure", relErr(Hg, Hc, 36), 1e-6);
      expectTrue(relErr(bg, bc, 6) <= 1e-6, "b with covariance + Geman-McClure", relErr(bg, bc, 6), 1e-6);
    }

    // ---- camera calibration (tst/camera_calibration.cpp:101-122) -------------------------
    {
      const double points[20] = {2.055643, 0.065643, 0.684357, 1, 1.963083, -0.765833, 0.653833, 1,
                                 2.927500, 0.707000, 0.125250, 1, 2.957833, 0.384667,  0.123667, 1,
                                 2.756000, 0.712000, -0.298000, 1};
      const std::int32_t pixels[10] = {621, 67, 878, 76, 491, 279, 559, 282, 481, 388};
      const double ceres[6] = {-0.0101064, 0.0206767, -0.0582803, 0.0183564, -0.00130745, 0.027414};
      auto cam = std::make_shared<mh::ReprojectionDeviceModel>(points, pixels, 5);
      mh::CostFunctionNumericalHip<double> cost(cam, 6, 2, 5);
      moptimizer::LevenbergMarquadtDynamic<double> lm(6);
      lm.addCost(&cost);
      double x[6] = {0};
      lm.minimize(x);
      double e = 0;
      for (int i = 0; i < 6; ++i) e = std::max(e, std::fabs(x[i] - ceres[i]));
      expectTrue(e < 5e-5, "CameraCalibration.GoodWeather with the HIP cost", e, 5e-5);
    }

    // ---- the compile-time-dimension twins (tst/point2point.cpp:149,151,211) ------------------
    {
      mh::CostFunctionAnalytical<Scalar, 6, 3> an_s(gpu_model, n);
      mh::CostFunctionNumerical<Scalar, 6, 3> nu_s(gpu_model, n);
      mh::CostFunctionAnalyticalDynamic<Scalar> an_d(gpu_model, 6, 3, n);
      mh::CostFunctionNumericalDynamic<Scalar> nu_d(gpu_model, 6, 3, n);
      Scalar Ha[36], Hb[36], ba[6], bb[6];
      const Scalar sa = an_s.linearize(x0, Ha, ba), sb = an_d.linearize(x0, Hb, bb);
      expectTrue(sa == sb && relErr(Ha, Hb, 36) == 0.0, "static == dynamic analytic (HIP)", sa, sb);
      const Scalar sc = nu_s.linearize(x0, Ha, ba), sd = nu_d.linearize(x0, Hb, bb);
      expectTrue(sc == sd && relErr(Ha, Hb, 36) == 0.0, "static == dynamic numeric (HIP)", sc, sd);
      expectTrue(std::fabs(sa - sc) <= 1e-7 * sa, "sum analytic == sum numeric (1e-7 rel)", sa, sc);
    }

    // ---- the cost sharded over a device list (here the one GPU three times: shards combined by
    // the group's host path; on a multi-GPU node the same constructor runs one RCCL all-reduce) --
    {
      const std::vector<int> devices = {0, 0, 0};
      mh::CostFunctionNumericalDynamic<Scalar> sharded(gpu_model, 6, 3, n, devices);
      mh::CostFunctionNumericalDynamic<Scalar> single(gpu_model, 6, 3, n);
      sharded.setLossFunction(std::make_shared<moptimizer::loss::GemmanMCClure<Scalar>>(1000.0));
      single.setLossFunction(std::make_shared<moptimizer::loss::GemmanMCClure<Scalar>>(1000.0));
      Scalar Hs[36], H1[36], bs[6], b1[6];
      double x1[6] = {0.5, -0.3, 0.2, 0.1, -0.2, 0.3};
      const Scalar ss = sharded.linearize(x1, Hs, bs), s1 = single.linearize(x1, H1, b1);
      expectTrue(std::fabs(ss - s1) <= 1e-12 * s1, "sharded cost sum == single-device sum", ss, s1);
      expectTrue(relErr(Hs, H1, 36) <= 1e-12, "sharded H == single-device H", relErr(Hs, H1, 36), 1e-12);
      expectTrue(relErr(bs, b1, 6) <= 1e-12, "sharded b == single-device b", relErr(bs, b1, 6), 1e-12);
      expectTrue(std::fabs(sharded.computeCost(x1) - single.computeCost(x1)) <= 1e-12 * s1,
                 "sharded computeCost == single-device", sharded.computeCost(x1), single.computeCost(x1));
      moptimizer::LevenbergMarquadtDynamic<Scalar> lm(6);
      lm.setMaximumIterations(50);
      mh::CostFunctionNumerical<Scalar, 6, 3> sharded_static(gpu_model, n, devices);
      lm.addCost(&sharded_static);
      double xs[6] = {0};
      lm.minimize(xs);
      double e = 0;
      for (int i = 0; i < 6; ++i) e = std::max(e, std::fabs(xs[i] - truth[i]));
      expectTrue(e < 1e-6, "LM over the sharded cost reaches the fixture pose", e, 1e-6);
    }

    // ---- ICP: correspondences unknown, re-searched by model->update(x) inside the LM loop ----
    {
      // target = the cloud moved by a small pose, in reversed order (index alignment destroyed)
      const double xt[6] = {0.15, -0.10, 0.05, 0.02, -0.01, 0.03};
      double Tt[16];
      moptimizer::so3::convert6DOFParameterToMatrix<double>(xt, Tt);
      std::vector<Scalar> moved(src.size());
      for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
          moved[3 * (n - 1 - i) + r] = Tt[0 * 4 + r] * src[3 * i] + Tt[1 * 4 + r] * src[3 * i + 1] +
                                       Tt[2 * 4 + r] * src[3 * i + 2] + Tt[3 * 4 + r];
      auto icp = std::make_shared<mh::IcpDeviceModel<Scalar>>(src.data(), n, moved.data(), n, 1.0);
      mh::CostFunctionAnalyticalHip<Scalar> cost(icp, 6, 3, n);
      moptimizer::LevenbergMarquadtDynamic<Scalar> lm(6);
      lm.setMaximumIterations(60);
      lm.addCost(&cost);
      double x[6] = {0, 0, 0, 0, 0, 0};
      lm.minimize(x);   // every outer iteration: cost.update(x) -> GPU nearest-neighbour search
      double e = 0;
      for (int i = 0; i < 6; ++i) e = std::max(e, std::fabs(x[i] - xt[i]));
      expectTrue(e < 1e-5, "ICP with GPU correspondence search recovers the pose", e, 1e-5);
    }

    // ---- a host model is refused, not silently run on the CPU ----------------------------
    {
      bool threw = false;
      try {
        mh::CostFunctionNumericalHip<Scalar> bad(cpu_model, 6, 3, n);
      } catch (const moptimizer::Exception &) {
        threw = true;
      }
      expectTrue(threw, "host IBaseModel rejected by the HIP cost (no CPU fallback)");
    }
  } catch (const std::exception &e) {
    std::printf("FAIL exception: %s\n", e.what());
    return 1;
  }
  std::printf("SUMMARY failures=%d\n", g_fail);
  return g_fail == 0 ? 0 : 1;
}
