// The binding compiled against the REFERENCE'S OWN interface headers.
//
// Every other C++ test here compiles include/moptimizer_amd/cost_function_hip.hpp over
// host_api.hpp, this repository's Eigen-free mirror of the reference's declarations.  A maintainer
// of the reference would compile the other branch: the reference's <moptimizer/cost_function.h>
// (:15-59), model.h (:11-104), loss_function/geman_mcclure.h (:6-19), covariance/covariance.h
// (:10-13), exception.h first, then MOPTIMIZER_AMD_USE_REFERENCE_HEADERS.  This
// program is that branch, built by tests/test_reference_headers_binding.py with
//   -I/root/reference/include  -Itests/support/eigen_decl  (a declaration of the one Eigen type those
//    headers name; Eigen3 itself is not in this image)
// and linked against libmoptimizer_hip.so.  It instantiates every cost class and device model of
// the binding in float and double, sets the reference's own loss and covariance objects on them,
// and runs:
//   * without a HIP device (the CPU suite): every construction must end in moptimizer::Exception
//     carrying the library's MOPT_ERR_NO_DEVICE message — the error path through the reference's
//     exception type, no CPU substitute behind it;
//   * with a device: the costs are built, linearized with GemmanMCClure + a covariance set through
//     the reference's setters, and checked for finite, symmetric H.
// Host-only checks in both cases: the reference's GemmanMCClure keeps its threshold private, the
// binding recovers it from weight(1) (gemanMcClureThreshold) — here against the real class.
//
// Built a second time WITHOUT the reference (-DBINDING_OVER_HOST_API, tests/cpp/Makefile ->
// _build/binding_all_classes; /root/reference does not exist on the GPU box and nothing compiled
// from it travels there): the same program over host_api.hpp, run by tests/test_gpu_dropin_cpp.py
// on the GPU, so that the with-a-device branch below is exercised too.
#ifndef BINDING_OVER_HOST_API
#include <moptimizer/cost_function.h>
#include <moptimizer/loss_function/geman_mcclure.h>
#define MOPTIMIZER_AMD_USE_REFERENCE_HEADERS
#endif
#include <moptimizer_amd/cost_function_hip.hpp>
#include <moptimizer_amd/levenberg_marquadt_device.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <type_traits>
#include <vector>

namespace mh = moptimizer::hip;

static int g_fail = 0, g_checks = 0, g_no_device = 0, g_built = 0;
static void expectTrue(const char *what, bool ok) {
  ++g_checks;
  if (!ok) ++g_fail;
  std::printf("%s %s\n", ok ? "PASS" : "FAIL", what);
}

// the binding's classes ARE the reference's types: these are compile-time facts of this branch
static_assert(std::is_base_of<moptimizer::CostFunctionBase<double>, mh::CostFunctionAnalyticalHip<double>>::value,
              "derives from the reference's CostFunctionBase");
static_assert(std::is_base_of<moptimizer::CostFunctionBase<float>, mh::CostFunctionNumericalHip<float>>::value,
              "derives from the reference's CostFunctionBase (float)");
static_assert(std::is_base_of<moptimizer::IBaseModel<double>, mh::Point2PointDeviceModel<double>>::value,
              "device models are the reference's IBaseModel");
#ifndef BINDING_OVER_HOST_API
static_assert(std::is_same<moptimizer::covariance::Matrix<double>,
                           Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic>>::value,
              "covariance::Matrix is the Eigen type the reference names");
#endif
static_assert(!std::is_copy_constructible<mh::CostFunctionAnalyticalHip<double>>::value,
              "copying stays deleted (cost_function.h:33-34)");

// every member of the device-resident optimizer compiled against the reference's types
// (OptimizationStatus of types.h:6-12, CostFunctionBase of cost_function.h)
template class moptimizer::hip::LevenbergMarquadtDevice<double>;
template class moptimizer::hip::LevenbergMarquadtDevice<float>;

// Builds a cost, and either finds no device (moptimizer::Exception with the library's message) or
// exercises it through the reference's own virtuals and setters.
template <class Scalar, class Cost, class MakeCost>
static void exercise(const char *name, int n, int m, MakeCost make) {
  std::unique_ptr<Cost> cost;
  try {
    cost = make();
  } catch (const moptimizer::Exception &e) {
    const bool no_device = std::strstr(e.what(), "no HIP device") != nullptr;
    char what[256];
    std::snprintf(what, sizeof what, "%-44s no device -> moptimizer::Exception: %.120s", name, e.what());
    expectTrue(what, no_device);
    ++g_no_device;
    return;
  }
  ++g_built;
  moptimizer::CostFunctionBase<Scalar> *base = cost.get();  // what Optimizer::addCost takes (optimizer.h:58)
  base->setLossFunction(std::make_shared<moptimizer::loss::GemmanMCClure<Scalar>>(Scalar(100)));
  auto cov = std::make_shared<moptimizer::covariance::Matrix<Scalar>>();
  cov->resize(m, m);
  cov->setIdentity();
  (*cov)(0, 0) = Scalar(2);
  base->setCovariance(cov);
  std::vector<Scalar> x(std::size_t(n), Scalar(0.01)), H(std::size_t(n * n)), b((std::size_t(n)));
  Scalar sum = 0, again = 0;
  try {
    base->update(x.data());
    sum = base->linearize(x.data(), H.data(), b.data());
    again = base->computeCost(x.data());
  } catch (const std::exception &e) {
    char what[320];
    std::snprintf(what, sizeof what, "%-44s on the device: threw %.200s", name, e.what());
    expectTrue(what, false);
    return;
  }
  bool ok = std::isfinite(double(sum)) && std::fabs(double(sum - again)) <= 1e-5 * std::fabs(double(sum)) + 1e-12;
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      const double a = double(H[std::size_t(c * n + r)]), t = double(H[std::size_t(r * n + c)]);
      ok = ok && std::isfinite(a) && std::fabs(a - t) <= 1e-4 * (std::fabs(a) + std::fabs(t)) + 1e-9;
    }
  char what[256];
  std::snprintf(what, sizeof what, "%-44s on the device: finite symmetric H, cost %.6g", name, double(sum));
  expectTrue(what, ok);
}

template <class Scalar>
static void point2pointFamily(const char *tag) {
  const int count = 700;
  static std::vector<Scalar> src, tgt;
  src.assign(std::size_t(3 * count), 0);
  tgt.assign(std::size_t(3 * count), 0);
  for (int i = 0; i < 3 * count; ++i) {
    src[std::size_t(i)] = Scalar(0.37 * (i % 29) - 1.0);
    tgt[std::size_t(i)] = src[std::size_t(i)] + Scalar(0.05 * ((i % 7) - 3));
  }
  auto model = std::make_shared<mh::Point2PointDeviceModel<Scalar>>(src.data(), tgt.data(), std::size_t(count));
  char name[96];
#define P2P_CASE(Class, ...)                                                                       \
  std::snprintf(name, sizeof name, #Class "<%s>", tag);                                            \
  exercise<Scalar, mh::Class<Scalar, ##__VA_ARGS__>>(name, 6, 3, [&] {                             \
    return std::make_unique<mh::Class<Scalar, ##__VA_ARGS__>>(model, 6, 3, count);                 \
  })
  P2P_CASE(CostFunctionAnalyticalHip);
  P2P_CASE(CostFunctionNumericalHip);
  P2P_CASE(CostFunctionAnalyticalTstLayoutHip);
  P2P_CASE(CostFunctionAnalyticalLeftHip);
  P2P_CASE(CostFunctionAnalyticalRightHip);
  P2P_CASE(CostFunctionAnalyticalDynamic);
  P2P_CASE(CostFunctionNumericalDynamic);
#undef P2P_CASE
  // the static twins: (model, N) as cost_function_analytical.h:21-25 / cost_function_numerical.h:24-28
  std::snprintf(name, sizeof name, "CostFunctionAnalytical<%s, 6, 3>", tag);
  exercise<Scalar, mh::CostFunctionAnalytical<Scalar, 6, 3>>(name, 6, 3, [&] {
    return std::make_unique<mh::CostFunctionAnalytical<Scalar, 6, 3>>(model, count);
  });
  std::snprintf(name, sizeof name, "CostFunctionNumerical<%s, 6, 3>", tag);
  exercise<Scalar, mh::CostFunctionNumerical<Scalar, 6, 3>>(name, 6, 3, [&] {
    return std::make_unique<mh::CostFunctionNumerical<Scalar, 6, 3>>(model, count);
  });
  // correspondence search behind model->update(x) (cost_function.h:44 -> model.h:24-26)
  auto icp = std::make_shared<mh::IcpDeviceModel<Scalar>>(src.data(), std::size_t(count), tgt.data(),
                                                           std::size_t(count), 0.5);
  std::snprintf(name, sizeof name, "IcpDeviceModel<%s> numeric", tag);
  exercise<Scalar, mh::CostFunctionNumericalHip<Scalar>>(name, 6, 3, [&] {
    return std::make_unique<mh::CostFunctionNumericalHip<Scalar>>(icp, 6, 3, count);
  });
  // rational model of tst/test_models.h:7-20 and a user-written model compiled at run time
  static std::vector<Scalar> t, y;
  t.assign(std::size_t(count), 0);
  y.assign(std::size_t(count), 0);
  for (int i = 0; i < count; ++i) {
    t[std::size_t(i)] = Scalar(0.1 + 0.01 * i);
    y[std::size_t(i)] = Scalar(0.4 * t[std::size_t(i)] / (0.6 + t[std::size_t(i)]));
  }
  auto rational = std::make_shared<mh::RationalDeviceModel<Scalar>>(t.data(), y.data());
  std::snprintf(name, sizeof name, "RationalDeviceModel<%s> analytic", tag);
  exercise<Scalar, mh::CostFunctionAnalytical<Scalar, 2, 1>>(name, 2, 1, [&] {
    return std::make_unique<mh::CostFunctionAnalytical<Scalar, 2, 1>>(rational, count);
  });
  auto jit = std::make_shared<mh::JitDeviceModel<Scalar>>(
      2, 1, "r[0] = d[1] - x[0] * d[0] / (x[1] + d[0]);",
      "J[0] = -d[0] / (x[1] + d[0]); J[1] = x[0] * d[0] / ((x[1] + d[0]) * (x[1] + d[0]));",
      std::vector<const Scalar *>{t.data(), y.data()});
  std::snprintf(name, sizeof name, "JitDeviceModel<%s> numeric", tag);
  exercise<Scalar, mh::CostFunctionNumericalHip<Scalar>>(name, 2, 1, [&] {
    return std::make_unique<mh::CostFunctionNumericalHip<Scalar>>(jit, 2, 1, count);
  });
}

int main() {
  std::setvbuf(stdout, nullptr, _IOLBF, 0);
  // ---- host-only: the private threshold of the reference's GemmanMCClure, recovered ----------
  {
    moptimizer::loss::GemmanMCClure<double> gm(100.0);
    moptimizer::loss::GemmanMCClure<float> gmf(0.8f);
    moptimizer::loss::GemmanMCClure<double> tiny(1e-3);
    expectTrue("gemanMcClureThreshold(reference GemmanMCClure<double>(100)) = 100",
               std::fabs(mh::gemanMcClureThreshold(&gm, 0) - 100.0) < 1e-9);
    expectTrue("gemanMcClureThreshold(reference GemmanMCClure<float>(0.8)) = 0.8",
               std::fabs(mh::gemanMcClureThreshold(&gmf, 0) - 0.8) < 1e-5);
    expectTrue("gemanMcClureThreshold(reference GemmanMCClure<double>(1e-3)) = 1e-3",
               std::fabs(mh::gemanMcClureThreshold(&tiny, 0) - 1e-3) < 1e-12);
    // a host model of the reference's own kind is refused: there is no CPU path behind the classes
    struct HostModel : moptimizer::BaseModel<double, HostModel> {
      bool f(const double *, double *f_x, unsigned int) const override { f_x[0] = 0; return true; }
    };
    bool refused = false;
    try {
      mh::CostFunctionNumericalHip<double> c(std::make_shared<HostModel>(), 6, 3, 10);
    } catch (const moptimizer::Exception &e) {
      refused = std::strstr(e.what(), "device model") != nullptr;
    }
    expectTrue("a host IBaseModel handed to a HIP cost class throws moptimizer::Exception", refused);
    // device models keep the reference's per-index virtuals only to refuse them
    double one = 0;
    mh::PowellDeviceModel powell;
    bool threw = false;
    try {
      static_cast<moptimizer::IBaseModel<double> &>(powell).f(&one, &one, 0);
    } catch (const moptimizer::Exception &) {
      threw = true;
    }
    expectTrue("DeviceModel::f on the host throws moptimizer::Exception", threw);
    expectTrue("DeviceModel::clone returns an IBaseModel::Ptr of the same model",
               dynamic_cast<mh::PowellDeviceModel *>(powell.clone().get()) != nullptr);
  }

  // ---- host-only: the optimizer surface of optimizer.h:26-54 over the reference's types ------
  {
    mh::LevenbergMarquadtDevice<double> lm(6);
    double x[6] = {0, 0, 0, 0, 0, 0};
    bool no_costs = false, negative = false, host_cost = false;
    try {
      lm.minimize(x);
    } catch (const std::runtime_error &) {
      no_costs = true;
    }
    try {
      lm.setMaximumIterations(-1);
    } catch (const std::invalid_argument &) {
      negative = true;
    }
    struct HostCost : moptimizer::CostFunctionBase<double> {
      HostCost() : CostFunctionBase<double>(nullptr, 0) {}
      double computeCost(const double *) override { return 0; }
      double linearize(const double *, double *, double *) override { return 0; }
    } host;
    try {
      lm.addCost(&host);
    } catch (const moptimizer::Exception &) {
      host_cost = true;
    }
    expectTrue("LevenbergMarquadtDevice::minimize without costs throws std::runtime_error (optimizer.h:48-54)", no_costs);
    expectTrue("setMaximumIterations(-1) throws std::invalid_argument (optimizer.h:33-37)", negative);
    expectTrue("a host CostFunctionBase is refused by the device-resident optimizer", host_cost);
    expectTrue("step() is the reference's stub", lm.step(x) == moptimizer::OptimizationStatus::NUMERIC_ERROR);
  }

  point2pointFamily<double>("double");
  point2pointFamily<float>("float");

  // ---- the fp64-only models ------------------------------------------------------------------
  {
    const int count = 300;
    static std::vector<double> pts(std::size_t(4 * count));
    static std::vector<std::int32_t> pix(std::size_t(2 * count));
    for (int i = 0; i < count; ++i) {
      pts[std::size_t(4 * i)] = 0.01 * (i % 17) - 0.05;
      pts[std::size_t(4 * i + 1)] = 0.01 * (i % 13) - 0.04;
      pts[std::size_t(4 * i + 2)] = 1.0 + 0.002 * i;
      pts[std::size_t(4 * i + 3)] = 1.0;
      pix[std::size_t(2 * i)] = 300 + i % 40;
      pix[std::size_t(2 * i + 1)] = 200 + i % 30;
    }
    auto camera = std::make_shared<mh::ReprojectionDeviceModel>(pts.data(), pix.data(), std::size_t(count));
    exercise<double, mh::CostFunctionNumerical<double, 6, 2>>("ReprojectionDeviceModel numeric", 6, 2, [&] {
      return std::make_unique<mh::CostFunctionNumerical<double, 6, 2>>(camera, count);
    });
    static std::vector<double> ty(std::size_t(2 * count));
    for (int i = 0; i < count; ++i) {
      ty[std::size_t(2 * i)] = 0.01 * i;
      ty[std::size_t(2 * i + 1)] = std::exp(0.3 * 0.01 * i + 0.1);
    }
    auto curve = std::make_shared<mh::ExpCurveDeviceModel>(ty.data());
    // CurveFittingModel is a BaseModel (tst/curve_fitting.cpp:81-98: no Jacobian): forward differences
    exercise<double, mh::CostFunctionNumerical<double, 2, 1>>("ExpCurveDeviceModel numeric", 2, 1, [&] {
      return std::make_unique<mh::CostFunctionNumerical<double, 2, 1>>(curve, count);
    });
    // ... and asked for its Jacobian it answers as BaseModel::f_df does (model.h:66-70), on the device too
    try {
      mh::CostFunctionAnalytical<double, 2, 1> analytic(curve, count);
      double x[2] = {0.1, 0.1}, H[4], b[2];
      bool threw = false;
      try {
        analytic.linearize(x, H, b);
      } catch (const moptimizer::Exception &e) {
        threw = std::strstr(e.what(), "Non implemented non-jacobian model function") != nullptr;
      }
      expectTrue("a model without a Jacobian linearized analytically throws BaseModel's exception", threw);
    } catch (const moptimizer::Exception &) {
      // no device: counted above
    }
    auto powell = std::make_shared<mh::PowellDeviceModel>();
    exercise<double, mh::CostFunctionNumerical<double, 4, 4>>("PowellDeviceModel numeric", 4, 4, [&] {
      return std::make_unique<mh::CostFunctionNumerical<double, 4, 4>>(powell, 1);
    });
    // sharded over a device list: the group form of the same class
    static std::vector<double> src(std::size_t(3 * count), 0.5), tgt(std::size_t(3 * count), 0.25);
    auto model = std::make_shared<mh::Point2PointDeviceModel<double>>(src.data(), tgt.data(), std::size_t(count));
    exercise<double, mh::CostFunctionAnalyticalHip<double>>("CostFunctionAnalyticalHip over {0}", 6, 3, [&] {
      return std::make_unique<mh::CostFunctionAnalyticalHip<double>>(model, 6, 3, count, std::vector<int>{0});
    });
  }

#ifdef BINDING_OVER_HOST_API
  std::printf("declarations: host_api.hpp (this repository's mirror)\n");
#else
  std::printf("declarations: the reference's <moptimizer/cost_function.h>\n");
#endif
  std::printf("%d checks, %d failures; %d costs built on a device, %d refused for want of one\n", g_checks,
              g_fail, g_built, g_no_device);
  if (g_built && g_no_device) {
    std::printf("FAIL some costs found a device and others did not\n");
    return 1;
  }
  return g_fail ? 1 : 0;
}
