// Drop-in check, C++ side, for the reference's remaining known-answer tests: the same programs
// as tst/curve_fitting.cpp:101-147, tst/powell.cpp:62-136, tst/simple_model.cpp:28-82,
// tst/loss_function.cpp:45-60, tst/covariance.cpp:26-63, tst/multiple_objectives.cpp:102-132 and
// tst/differentiation.cpp:47-77,134-161 and tst/state_model.cpp:83-112, with `moptimizer::hip::` cost classes and device models
// in place of the CPU ones and the LM loop unchanged.  Expected values and tolerances are the
// reference's.  (curve_data.inc is the reference tests' data table; oracle/ is only its holder.)
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "moptimizer_amd/cost_function_hip.hpp"
#include "moptimizer_caller/levenberg_marquadt.hpp"

#include "curve_data.inc"

namespace mh = moptimizer::hip;
using moptimizer::LevenbergMarquadtDynamic;

static int g_fail = 0, g_checks = 0;
static void expectNear(const char *what, double got, double want, double tol) {
  ++g_checks;
  const bool ok = std::fabs(got - want) <= tol && !std::isnan(got);
  if (!ok) ++g_fail;
  std::printf("%s %-60s got % .10g want % .10g tol %.1e\n", ok ? "PASS" : "FAIL", what, got, want, tol);
}

static void curveFitting() {
  for (int ic = 0; ic < 2; ++ic) {
    LevenbergMarquadtDynamic<double> optimizer(2);
    mh::CostFunctionNumerical<double, 2, 1> cost(std::make_shared<mh::ExpCurveDeviceModel>(kCurveData),
                                                kNumObservations);
    if (ic == 1) optimizer.setMaximumIterations(50);
    optimizer.addCost(&cost);
    double x0[2] = {ic == 0 ? 0.0 : 1.20, ic == 0 ? 0.0 : 2.0};
    optimizer.minimize(x0);
    expectNear(ic == 0 ? "CurveFitting.InitialCondition1 x[0]" : "CurveFitting.InitialCondition2 x[0]",
               x0[0], 0.291861, ic == 0 ? 5e-5 : 1e-4);
    expectNear(ic == 0 ? "CurveFitting.InitialCondition1 x[1]" : "CurveFitting.InitialCondition2 x[1]",
               x0[1], 0.131439, ic == 0 ? 5e-5 : 1e-4);
  }
}

static void powell() {
  for (int variant = 0; variant < 3; ++variant) {
    double x0[] = {3, -1, 0, 4};
    LevenbergMarquadtDynamic<double> optimizer(4);
    optimizer.setMaximumIterations(25);
    std::unique_ptr<moptimizer::CostFunctionBase<double>> cost;
    if (variant == 0)
      cost.reset(new mh::CostFunctionNumerical<double, 4, 4>(std::make_shared<mh::PowellDeviceModel>(), 1));
    else
      cost.reset(new mh::CostFunctionNumericalDynamic<double>(std::make_shared<mh::PowellDeviceModel>(), 4, 4, 1));
    if (variant == 2) {
      auto covariance = std::make_shared<moptimizer::covariance::Matrix<double>>();
      covariance->resize(4, 4);
      covariance->setIdentity();
      *covariance *= 0.01;
      cost->setCovariance(covariance);
    }
    optimizer.addCost(cost.get());
    optimizer.minimize(x0);
    static const char *names[] = {"PowellFunction.InitialCondition0", "PowellFunction.IC0Dynamic",
                                  "PowellFunction.IC0DynamicCovariance"};
    for (int i = 0; i < 4; ++i) {
      char label[96];
      std::snprintf(label, sizeof label, "%s x[%d]", names[variant], i);
      expectNear(label, x0[i], 0.0, 5e-5);
    }
  }
}

static void simpleModelFloat() {
  float x_data[7] = {0.038, 0.194, 0.425, 0.626, 1.253, 2.5, 3.70};
  float y_data[7] = {0.05, 0.127, 0.094, 0.2122, 0.2729, 0.2665, 0.3317};
  const float starts[2][2] = {{0.9f, 0.2f}, {1.9f, 1.5f}};
  for (int with_loss = 0; with_loss < 2; ++with_loss)
    for (int s = 0; s < 2; ++s)
      for (int dyn_cost = 0; dyn_cost < 2; ++dyn_cost) {
        float x0[2] = {starts[s][0], starts[s][1]};
        LevenbergMarquadtDynamic<float> optimizer(2);
        std::unique_ptr<moptimizer::CostFunctionBase<float>> cost;
        auto model = std::make_shared<mh::RationalDeviceModel<float>>(x_data, y_data);
        if (dyn_cost)
          cost.reset(new mh::CostFunctionNumericalDynamic<float>(model, 2, 1, 7));
        else
          cost.reset(new mh::CostFunctionNumerical<float, 2, 1>(model, 7));
        if (with_loss)
          cost->setLossFunction(std::make_shared<moptimizer::loss::GemmanMCClure<float>>(100.0f));
        optimizer.addCost(cost.get());
        optimizer.minimize(x0);
        char label[96];
        std::snprintf(label, sizeof label, "SimpleModel(float%s%s) start%d x[0]", with_loss ? ",GM100" : "",
                      dyn_cost ? ",dyn" : "", s);
        expectNear(label, x0[0], 0.362, 0.01);
        std::snprintf(label, sizeof label, "SimpleModel(float%s%s) start%d x[1]", with_loss ? ",GM100" : "",
                      dyn_cost ? ",dyn" : "", s);
        expectNear(label, x0[1], 0.556, 0.01);
      }
}

static void multipleObjectives() {
  LevenbergMarquadtDynamic<double> multi(2), single(2);
  double x_multi[] = {0.0, 0.0}, x_single[] = {0.0, 0.0};
  mh::CostFunctionNumerical<double, 2, 1> whole(std::make_shared<mh::ExpCurveDeviceModel>(kCurveData), 67);
  mh::CostFunctionNumerical<double, 2, 1> first(std::make_shared<mh::ExpCurveDeviceModel>(kCurveData), 30);
  mh::CostFunctionNumerical<double, 2, 1> rest(std::make_shared<mh::ExpCurveDeviceModel>(&kCurveData[60]), 37);
  single.addCost(&whole);
  multi.addCost(&first);
  multi.addCost(&rest);
  multi.minimize(x_multi);
  single.minimize(x_single);
  expectNear("MultipleObjectives.SplitCost multi==single x[0]", x_multi[0], x_single[0], 1e-8);
  expectNear("MultipleObjectives.SplitCost multi==single x[1]", x_multi[1], x_single[1], 1e-8);
  expectNear("MultipleObjectives.SplitCost x[0]", x_multi[0], 0.291861, 5e-5);
  expectNear("MultipleObjectives.SplitCost x[1]", x_multi[1], 0.131439, 5e-5);
  // the same program with the two costs linked (the first one asked at an x queues the other's
  // sweep too): the loop is unchanged and so is every number
  mh::linkCosts<double>({&first, &rest});
  LevenbergMarquadtDynamic<double> linked(2);
  double x_linked[] = {0.0, 0.0};
  linked.addCost(&first);
  linked.addCost(&rest);
  linked.minimize(x_linked);
  expectNear("MultipleObjectives.SplitCost linked==unlinked x[0]", x_linked[0], x_multi[0], 0.0);
  expectNear("MultipleObjectives.SplitCost linked==unlinked x[1]", x_linked[1], x_multi[1], 0.0);
  mh::linkCosts<double>({});
}

template <typename S>
static void differentiationSimple(const char *tag) {
  S x_data[] = {S(0.038), S(0.194), S(0.425), S(0.626), S(1.253), S(2.5), S(3.70), S(5), S(0)};
  S y_data[] = {S(0.05), S(0.127), S(0.094), S(0.2122), S(0.2729), S(0.2665), S(0.3317), S(0.2), S(0)};
  auto model = std::make_shared<mh::RationalDeviceModel<S>>(x_data, y_data);
  mh::CostFunctionAnalytical<S, 2, 1> cost_ana(model, 9);
  mh::CostFunctionNumerical<S, 2, 1> cost_num(model, 9);
  S H[4], Hn[4], b[2];
  S x0[2] = {S(0.9), S(0.2)};
  char label[96];
  std::snprintf(label, sizeof label, "Differentiation.SimpleModel<%s> cost", tag);
  expectNear(label, cost_ana.computeCost(x0), cost_num.computeCost(x0), 1e-4);
  cost_ana.linearize(x0, H, b);
  cost_num.linearize(x0, Hn, b);
  for (int i = 0; i < 4; ++i) {
    std::snprintf(label, sizeof label, "Differentiation.SimpleModel<%s> H(%d)", tag, i);
    expectNear(label, H[i], Hn[i], 5e-3);
  }
}

static void differentiationPowell() {
  auto model = std::make_shared<mh::PowellDeviceModel>();
  mh::CostFunctionAnalytical<double, 4, 4> cost_ana(model, 1);
  mh::CostFunctionNumerical<double, 4, 4> cost_num(model, 1);
  double H[16], Hn[16], r[4];
  double x0[4] = {3, -1, 0, 4};
  expectNear("Differentiation.PowellModel cost", cost_ana.computeCost(x0), cost_num.computeCost(x0), 1e-4);
  cost_ana.linearize(x0, H, r);
  cost_num.linearize(x0, Hn, r);
  for (int i = 0; i < 16; ++i) {
    char label[96];
    std::snprintf(label, sizeof label, "Differentiation.PowellModel H(%d)", i);
    expectNear(label, H[i], Hn[i], 1e-4);
  }
}

static void covarianceScaling() {
  float x_data[7] = {0.038, 0.194, 0.425, 0.626, 1.253, 2.5, 3.70};
  float y_data[7] = {0.05, 0.127, 0.094, 0.2122, 0.2729, 0.2665, 0.3317};
  mh::CostFunctionNumericalDynamic<float> cost(std::make_shared<mh::RationalDeviceModel<float>>(x_data, y_data),
                                               2, 1, 7);
  float x0[2] = {1.9f, 1.5f};
  float H[4], b[2], Hc[4], bc[2];
  cost.linearize(x0, H, b);
  for (int pass = 0; pass < 2; ++pass) {
    const float cov_val = pass == 0 ? 1.0f : 0.5f;
    auto covariance = std::make_shared<moptimizer::covariance::Matrix<float>>();
    covariance->resize(1, 1);
    (*covariance)(0, 0) = cov_val;
    cost.setCovariance(covariance);
    cost.linearize(x0, Hc, bc);
    char label[96];
    for (int i = 0; i < 4; ++i) {
      std::snprintf(label, sizeof label, "testCovariance(%.1f) H(%d)", cov_val, i);
      expectNear(label, Hc[i], H[i] * cov_val, 1e-5);
    }
    for (int i = 0; i < 2; ++i) {
      std::snprintf(label, sizeof label, "testCovariance(%.1f) b(%d)", cov_val, i);
      expectNear(label, bc[i], b[i] * cov_val, 1e-5);
    }
  }
}

// tst/curve_fitting.cpp:81-147 once more, with the model written by the user (source text) rather
// than picked from the built-in ones: same LM program, same known answers; plus the user's
// Jacobian against forward differences as tst/differentiation.cpp does for its models.
static void userDefinedModel() {
  std::vector<double> t(kNumObservations), y(kNumObservations);
  for (int i = 0; i < kNumObservations; ++i) {
    t[i] = kCurveData[2 * i];
    y[i] = kCurveData[2 * i + 1];
  }
  auto model = std::make_shared<mh::JitDeviceModel<double>>(
      2, 1, "r[0] = d[1] - exp(x[0] * d[0] + x[1]);",
      "const S e = exp(x[0] * d[0] + x[1]); J[0] = -d[0] * e; J[1] = -e;",
      std::vector<const double *>{t.data(), y.data()});
  for (int analytic = 0; analytic < 2; ++analytic) {
    LevenbergMarquadtDynamic<double> optimizer(2);
    std::unique_ptr<moptimizer::CostFunctionBase<double>> cost;
    if (analytic)
      cost.reset(new mh::CostFunctionAnalytical<double, 2, 1>(model, kNumObservations));
    else
      cost.reset(new mh::CostFunctionNumerical<double, 2, 1>(model, kNumObservations));
    optimizer.addCost(cost.get());
    double x0[2] = {0.0, 0.0};
    optimizer.minimize(x0);
    expectNear(analytic ? "UserModel(jit) analytic x[0]" : "UserModel(jit) numeric x[0]", x0[0], 0.291861, 5e-5);
    expectNear(analytic ? "UserModel(jit) analytic x[1]" : "UserModel(jit) numeric x[1]", x0[1], 0.131439, 5e-5);
  }
  mh::CostFunctionAnalytical<double, 2, 1> cost_ana(model, kNumObservations);
  mh::CostFunctionNumerical<double, 2, 1> cost_num(model, kNumObservations);
  double H[4], Hn[4], b[2], x0[2] = {0.29, 0.13};
  cost_ana.linearize(x0, H, b);
  cost_num.linearize(x0, Hn, b);
  for (int i = 0; i < 4; ++i) {
    char label[96];
    std::snprintf(label, sizeof label, "UserModel(jit) analytic vs numeric H(%d)", i);
    expectNear(label, H[i], Hn[i], 1e-4 * std::fabs(Hn[i]));
  }
  bool threw = false;
  try {
    mh::CostFunctionNumerical<double, 2, 1> bad(
        std::make_shared<mh::JitDeviceModel<double>>(2, 1, "r[0] = undefined_symbol;", "",
                                                     std::vector<const double *>{t.data(), y.data()}),
        kNumObservations);
  } catch (const moptimizer::Exception &) {
    threw = true;
  }
  expectNear("UserModel(jit) source error surfaces as moptimizer::Exception", threw ? 1 : 0, 1, 0);
}

// tst/state_model.cpp:83-112: StateModel (n = m = 15, one residual block: x (-) x_init with the
// rotation part through Exp / Log) under CostFunctionNumericalDynamic(model, 15, 15, 1) and
// LevenbergMarquadtDynamic<double> lm(15).  The reference's test asserts nothing; the residual
// vanishes at x_init and nowhere else, so that is what the solve must return.
static void stateModel() {
  const char *residual = R"SRC(
  auto Exp = [](const S *w, S (&R)[9]) {
    const S t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? S(1) : S(0);
    if (t > S(10) * S(2.220446049250313e-16)) {
      const S a[3] = {w[0] / t, w[1] / t, w[2] / t};
      const S K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
      const S s = sin(t), c1 = S(1) - cos(t);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          S kk = 0;
          for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
          R[i * 3 + j] = ((i == j ? S(1) : S(0)) + s * K[i * 3 + j]) + c1 * kk;
        }
    }
  };
  S R0[9], R[9], rel[9];
  Exp(d, R0);
  Exp(x, R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      S v = 0;
      for (int k = 0; k < 3; ++k) v += R0[k * 3 + i] * R[k * 3 + j];
      rel[i * 3 + j] = v;
    }
  const S trace = rel[0] + rel[4] + rel[8];
  const S theta = (trace > S(3.0 - 1e-6)) ? S(0) : acos(S(0.5) * (trace - S(1)));
  const S K[3] = {rel[7] - rel[5], rel[2] - rel[6], rel[3] - rel[1]};
  const S k = (fabs(theta) < S(0.001)) ? S(0.5) : S(0.5) * theta / sin(theta);
  for (int i = 0; i < 3; ++i) r[i] = k * K[i];
  for (int i = 0; i < 12; ++i) r[3 + i] = x[3 + i] - d[3 + i];
)SRC";
  double x_init[15] = {0.6, 0.8, 0.3, -0.4, 0.11, -0.9};  // :88
  double x[15] = {0.1, 0.2, 0.3, 0.4, 0.5, 0.6};           // :89
  std::vector<const double *> planes;
  for (int p = 0; p < 15; ++p) planes.push_back(&x_init[p]);
  auto model = std::make_shared<mh::JitDeviceModel<double>>(15, 15, residual, "", planes);
  LevenbergMarquadtDynamic<double> lm(15);                                // :99
  mh::CostFunctionNumericalDynamic<double> cost(model, 15, 15, 1);        // :101
  lm.addCost(&cost);                                                      // :108
  lm.minimize(x);                                                         // :109
  for (int i = 0; i < 15; ++i) {
    char label[64];
    std::snprintf(label, sizeof label, "StateModel.Optimize x[%d]", i);
    expectNear(label, x[i], x_init[i], 1e-7);
  }
}

int main() {
  try {
    stateModel();
    curveFitting();
    userDefinedModel();
    powell();
    simpleModelFloat();
    multipleObjectives();
    differentiationSimple<float>("float");
    differentiationSimple<double>("double");
    differentiationPowell();
    covarianceScaling();
    // a model without Jacobian asked for one: the reference throws from f_df (model.h:66-70)
    bool threw = false;
    try {
      mh::CostFunctionAnalytical<double, 2, 1> bad(std::make_shared<mh::ExpCurveDeviceModel>(kCurveData), 67);
      double H[4], b[2], x[2] = {0, 0};
      bad.linearize(x, H, b);
    } catch (const moptimizer::Exception &) {
      threw = true;
    }
    expectNear("analytic linearize of a Jacobian-less model throws", threw ? 1 : 0, 1, 0);
  } catch (const std::exception &e) {
    std::printf("FAIL exception: %s\n", e.what());
    return 1;
  }
  std::printf("SUMMARY %d checks, failures=%d\n", g_checks, g_fail);
  return g_fail == 0 ? 0 : 1;
}
