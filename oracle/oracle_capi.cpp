// ORACLE — TEST INFRASTRUCTURE ONLY (see cost_computation.hpp).
//
// C entry points over the CPU restatement so that tests/ (ctypes), __graft_entry__.smoke() and
// bench.py's cpu_baseline leg can run the reference algorithm on the same inputs as the HIP
// path.  Every call goes through the restated cost classes (cpu_costs.hpp) and, for the
// minimize calls, the product's LM loop (the caller of the path, restated from
// /root/reference/src/levenberg_marquadt_dyn.cpp:34-119).
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "cpu_costs.hpp"
#include "moptimizer_caller/levenberg_marquadt.hpp"
#include "test_models.hpp"

namespace {

enum CostClass { kAnalyticDyn = 0, kNumericDyn = 1, kAnalyticStatic = 2, kNumericStatic = 3 };

template <class S>
typename moptimizer::loss::ILossFunction<S>::Ptr makeLoss(int kind, double param) {
  if (kind == 1) return std::make_shared<moptimizer::loss::GemmanMCClure<S>>(S(param));
  return std::make_shared<moptimizer::loss::NoLoss<S>>();
}

template <class S>
std::unique_ptr<moptimizer::CostFunctionBase<S>> makeCost(
    int cost_class, typename moptimizer::IBaseModel<S>::Ptr model, int n_params, int n_out,
    int count, const S *cov, int loss_kind, double loss_param) {
  std::unique_ptr<moptimizer::CostFunctionBase<S>> cost;
  switch (cost_class) {
    case kAnalyticDyn:
      cost.reset(new oracle::CostFunctionAnalyticalDynamic<S>(model, n_params, n_out, count));
      break;
    case kNumericDyn:
      cost.reset(new oracle::CostFunctionNumericalDynamic<S>(model, n_params, n_out, count));
      break;
    case kAnalyticStatic:
      // the static twins are instantiated for the (n, m) pairs the reference's tests use
      if (n_params == 6 && n_out == 3)
        cost.reset(new oracle::CostFunctionAnalytical<S, 6, 3>(model, count));
      else
        cost.reset(new oracle::CostFunctionAnalyticalDynamic<S>(model, n_params, n_out, count));
      break;
    default:
      if (n_params == 6 && n_out == 3)
        cost.reset(new oracle::CostFunctionNumerical<S, 6, 3>(model, count));
      else if (n_params == 6 && n_out == 2)
        cost.reset(new oracle::CostFunctionNumerical<S, 6, 2>(model, count));
      else
        cost.reset(new oracle::CostFunctionNumericalDynamic<S>(model, n_params, n_out, count));
      break;
  }
  if (cov) {
    auto m = std::make_shared<moptimizer::covariance::Matrix<S>>();
    m->resize(n_out, n_out);
    std::memcpy(m->data(), cov, sizeof(S) * n_out * n_out);
    cost->setCovariance(m);
  }
  cost->setLossFunction(makeLoss<S>(loss_kind, loss_param));
  return cost;
}

inline oracle::P2PJacobianLayout layoutOf(int layout) {
  switch (layout) {
    case 1: return oracle::P2PJacobianLayout::kAsWrittenInTst;
    case 2: return oracle::P2PJacobianLayout::kLeftPerturbation;
    case 3: return oracle::P2PJacobianLayout::kRightPerturbation;
    default: return oracle::P2PJacobianLayout::kRowMajor;
  }
}

template <class S>
int p2pLinearize(int cost_class, int layout, const S *src, const S *tgt, int n, const S *x,
                 const S *cov, int loss_kind, double loss_param, S *H, S *b, S *cost_out) {
  auto model = std::make_shared<oracle::Point2Point<S>>(
      src, tgt,
      layoutOf(layout));
  auto cost = makeCost<S>(cost_class, model, 6, 3, n, cov, loss_kind, loss_param);
  *cost_out = cost->linearize(x, H, b);
  return 0;
}

// All-cores variant for the timed CPU baseline: contiguous index ranges, one restated
// single-threaded sweep per worker, partials added in worker order.
template <class S>
int p2pLinearizeThreads(int cost_class, int layout, const S *src, const S *tgt, long long n,
                        const S *x, const S *cov, int loss_kind, double loss_param, int threads,
                        S *H, S *b, S *cost_out) {
  if (threads < 1) threads = 1;
  std::vector<std::vector<S>> parts(threads, std::vector<S>(43, S(0)));
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t]() {
      const long long lo = n * t / threads, hi = n * (t + 1) / threads;
      if (hi <= lo) return;
      S *p = parts[t].data();
      p2pLinearize<S>(cost_class, layout, src + 3 * lo, tgt + 3 * lo, int(hi - lo), x, cov,
                      loss_kind, loss_param, p, p + 36, p + 42);
    });
  }
  for (auto &th : pool) th.join();
  for (int k = 0; k < 36; ++k) H[k] = 0;
  for (int k = 0; k < 6; ++k) b[k] = 0;
  *cost_out = 0;
  for (int t = 0; t < threads; ++t) {
    for (int k = 0; k < 36; ++k) H[k] += parts[t][k];
    for (int k = 0; k < 6; ++k) b[k] += parts[t][36 + k];
    *cost_out += parts[t][42];
  }
  return 0;
}

template <class S>
int p2pMinimize(int cost_class, int layout, const S *src, const S *tgt, int n, S *x, int max_iter,
                int lm_iter, const S *cov, int loss_kind, double loss_param, int *status,
                int *iterations) {
  auto model = std::make_shared<oracle::Point2Point<S>>(
      src, tgt,
      layoutOf(layout & 3));
  auto cost = makeCost<S>(cost_class, model, 6, 3, n, cov, loss_kind, loss_param);
  moptimizer::LevenbergMarquadtDynamic<S> lm(6);
  lm.setMaximumIterations(max_iter);
  if (lm_iter > 0) lm.setLevenbergMarquadtIterations(lm_iter);
  // bits 2, 3 of `layout`: x (+) delta on SE(3), composed on the left / on the right
  lm.setManifoldUpdate((layout & 4) ? 1 : ((layout & 8) ? 2 : 0));
  lm.addCost(cost.get());
  *status = int(lm.minimize(x));
  *iterations = int(lm.getExecutedIterations());
  return 0;
}

}  // namespace

extern "C" {

// scalar_bytes: 4 = float, 8 = double.  cost_class: 0 analytic-dynamic, 1 numeric-dynamic,
// 2 analytic-static<6,3>, 3 numeric-static<6,3>.  layout: 0 row-major Jacobian, 1 as written in
// tst/point2point.cpp.  cov: m*m column-major or NULL (identity).  loss_kind: 0 none,
// 1 Geman-McClure(loss_param).  H: 36 column-major, b: 6, cost: 1, all in the call's scalar.
int oracle_p2p_linearize(int scalar_bytes, int cost_class, int layout, const void *src,
                         const void *tgt, int n, const void *x, const void *cov, int loss_kind,
                         double loss_param, void *H, void *b, void *cost) {
  try {
    if (scalar_bytes == 8)
      return p2pLinearize<double>(cost_class, layout, (const double *)src, (const double *)tgt, n,
                                  (const double *)x, (const double *)cov, loss_kind, loss_param,
                                  (double *)H, (double *)b, (double *)cost);
    if (scalar_bytes == 4)
      return p2pLinearize<float>(cost_class, layout, (const float *)src, (const float *)tgt, n,
                                 (const float *)x, (const float *)cov, loss_kind, loss_param,
                                 (float *)H, (float *)b, (float *)cost);
  } catch (...) {
    return -2;
  }
  return -1;
}

int oracle_p2p_linearize_threads(int scalar_bytes, int cost_class, int layout, const void *src,
                                 const void *tgt, long long n, const void *x, const void *cov,
                                 int loss_kind, double loss_param, int threads, void *H, void *b,
                                 void *cost) {
  try {
    if (scalar_bytes == 8)
      return p2pLinearizeThreads<double>(cost_class, layout, (const double *)src,
                                         (const double *)tgt, n, (const double *)x,
                                         (const double *)cov, loss_kind, loss_param, threads,
                                         (double *)H, (double *)b, (double *)cost);
    if (scalar_bytes == 4)
      return p2pLinearizeThreads<float>(cost_class, layout, (const float *)src,
                                        (const float *)tgt, n, (const float *)x,
                                        (const float *)cov, loss_kind, loss_param, threads,
                                        (float *)H, (float *)b, (float *)cost);
  } catch (...) {
    return -2;
  }
  return -1;
}

// Sum of squared residuals.  threads == 0: the sequential definition (linearization.h:36-47);
// threads > 0: the parallel sweep (:49-63) with that many workers.
int oracle_p2p_cost(int scalar_bytes, const void *src, const void *tgt, int n, const void *x,
                    int threads, void *cost) {
  try {
    if (scalar_bytes == 8) {
      auto model = std::make_shared<oracle::Point2Point<double>>((const double *)src,
                                                                 (const double *)tgt);
      oracle::CostComputation<double> cc(6, 3);
      *(double *)cost = threads > 0
                            ? cc.parallelComputeCost((const double *)x, model, n, threads)
                            : cc.computeCost((const double *)x, model, n);
      return 0;
    }
    if (scalar_bytes == 4) {
      auto model =
          std::make_shared<oracle::Point2Point<float>>((const float *)src, (const float *)tgt);
      oracle::CostComputation<float> cc(6, 3);
      *(float *)cost = threads > 0 ? cc.parallelComputeCost((const float *)x, model, n, threads)
                                   : cc.computeCost((const float *)x, model, n);
      return 0;
    }
  } catch (...) {
    return -2;
  }
  return -1;
}

// LM over one point2point cost; x is updated in place.  status = moptimizer::OptimizationStatus.
int oracle_p2p_minimize(int scalar_bytes, int cost_class, int layout, const void *src,
                        const void *tgt, int n, void *x, int max_iter, int lm_iter,
                        const void *cov, int loss_kind, double loss_param, int *status,
                        int *iterations) {
  try {
    if (scalar_bytes == 8)
      return p2pMinimize<double>(cost_class, layout, (const double *)src, (const double *)tgt, n,
                                 (double *)x, max_iter, lm_iter, (const double *)cov, loss_kind,
                                 loss_param, status, iterations);
    if (scalar_bytes == 4)
      return p2pMinimize<float>(cost_class, layout, (const float *)src, (const float *)tgt, n,
                                (float *)x, max_iter, lm_iter, (const float *)cov, loss_kind,
                                loss_param, status, iterations);
  } catch (...) {
    return -2;
  }
  return -1;
}

// Reprojection (camera calibration) cost, double only, forward differences only.
int oracle_camera_linearize(const double *points_xyzw, const int32_t *pixels_uv, int n,
                            const double *x, const double *cov, int loss_kind, double loss_param,
                            double *H, double *b, double *cost) {
  try {
    auto model = std::make_shared<oracle::CameraModel>(points_xyzw, pixels_uv, (size_t)n);
    auto c = makeCost<double>(kNumericStatic, model, 6, 2, n, cov, loss_kind, loss_param);
    *cost = c->linearize(x, H, b);
    return 0;
  } catch (...) {
    return -2;
  }
}

int oracle_camera_cost(const double *points_xyzw, const int32_t *pixels_uv, int n,
                       const double *x, double *cost) {
  try {
    auto model = std::make_shared<oracle::CameraModel>(points_xyzw, pixels_uv, (size_t)n);
    oracle::CostComputation<double> cc(6, 2);
    *cost = cc.computeCost(x, model, n);
    return 0;
  } catch (...) {
    return -2;
  }
}

// LM over `num_costs` reprojection costs that split the element list at the given counts
// (multi-objective form, levenberg_marquadt_dyn.cpp:48-60), each with the same loss.
int oracle_camera_minimize(const double *points_xyzw, const int32_t *pixels_uv,
                           const int *counts, int num_costs, double *x, int max_iter,
                           int loss_kind, double loss_param, int *status, int *iterations) {
  try {
    std::vector<std::unique_ptr<moptimizer::CostFunctionBase<double>>> costs;
    moptimizer::LevenbergMarquadtDynamic<double> lm(6);
    lm.setMaximumIterations(max_iter);
    size_t offset = 0;
    for (int k = 0; k < num_costs; ++k) {
      auto model = std::make_shared<oracle::CameraModel>(points_xyzw + 4 * offset,
                                                         pixels_uv + 2 * offset, (size_t)counts[k]);
      costs.push_back(
          makeCost<double>(kNumericStatic, model, 6, 2, counts[k], nullptr, loss_kind, loss_param));
      lm.addCost(costs.back().get());
      offset += counts[k];
    }
    *status = int(lm.minimize(x));
    *iterations = int(lm.getExecutedIterations());
    return 0;
  } catch (...) {
    return -2;
  }
}

// The reference tests' small parametric models (model_kind: 1 exp curve, 2 rational, 3 Powell; 4 / 5:
// the rational / exp curve model whose f / f_df return false for an observation with a NaN y),
// scalar_bytes 4 or 8 (exp curve / Powell: 8 only), numeric = 0 analytic cost class, 1 numeric.
// t, y: `count` scalars each.  H: n*n column-major, b: n.
int oracle_scalar_linearize(int scalar_bytes, int model_kind, int numeric, const void *t,
                            const void *y, int count, const void *x, const void *cov,
                            int loss_kind, double loss_param, void *H, void *b, void *cost) {
  try {
    const int cc = numeric ? kNumericDyn : kAnalyticDyn;
    if (model_kind == 2 && scalar_bytes == 4) {
      auto model = std::make_shared<oracle::RationalModel<float>>((const float *)t, (const float *)y);
      auto c = makeCost<float>(cc, model, 2, 1, count, (const float *)cov, loss_kind, loss_param);
      *(float *)cost = c->linearize((const float *)x, (float *)H, (float *)b);
      return 0;
    }
    if (model_kind == 4 && scalar_bytes == 4) {
      auto model = std::make_shared<oracle::SkippingRationalModel<float>>((const float *)t, (const float *)y);
      auto c = makeCost<float>(cc, model, 2, 1, count, (const float *)cov, loss_kind, loss_param);
      *(float *)cost = c->linearize((const float *)x, (float *)H, (float *)b);
      return 0;
    }
    if (scalar_bytes != 8) return -1;
    std::unique_ptr<moptimizer::CostFunctionBase<double>> c;
    std::vector<double> interleaved;
    if (model_kind == 1 || model_kind == 5) {
      interleaved.resize(2 * size_t(count));
      for (int i = 0; i < count; ++i) {
        interleaved[2 * i] = ((const double *)t)[i];
        interleaved[2 * i + 1] = ((const double *)y)[i];
      }
      if (model_kind == 1)
        c = makeCost<double>(cc, std::make_shared<oracle::CurveFittingModel>(interleaved.data()), 2, 1,
                             count, (const double *)cov, loss_kind, loss_param);
      else
        c = makeCost<double>(cc, std::make_shared<oracle::SkippingCurveFittingModel>(interleaved.data()),
                             2, 1, count, (const double *)cov, loss_kind, loss_param);
    } else if (model_kind == 2) {
      c = makeCost<double>(cc, std::make_shared<oracle::RationalModel<double>>((const double *)t, (const double *)y),
                           2, 1, count, (const double *)cov, loss_kind, loss_param);
    } else if (model_kind == 4) {
      c = makeCost<double>(cc, std::make_shared<oracle::SkippingRationalModel<double>>((const double *)t, (const double *)y),
                           2, 1, count, (const double *)cov, loss_kind, loss_param);
    } else if (model_kind == 3) {
      c = makeCost<double>(cc, std::make_shared<oracle::PowellModel>(), 4, 4, count,
                           (const double *)cov, loss_kind, loss_param);
    } else {
      return -1;
    }
    *(double *)cost = c->linearize((const double *)x, (double *)H, (double *)b);
    return 0;
  } catch (...) {
    return -2;
  }
}

// The transforms the restatement derives from x: column-major 4x4 for x and, when n_plus = 6,
// for each forward-difference point x + h_j e_j (linearization.h:78-92), plus the steps h.
int oracle_se3_from_x(const double *x, double *T16, double *T16_plus /* 6*16 or NULL */,
                      double *h /* 6 or NULL */) {
  oracle::so3::convert6DOFParameterToMatrix<double>(x, T16);
  if (T16_plus && h) {
    const double min_step = std::sqrt(std::numeric_limits<double>::epsilon());
    for (int j = 0; j < 6; ++j) {
      double xp[6];
      for (int k = 0; k < 6; ++k) xp[k] = x[k];
      oracle::forwardStep<double>(x[j], min_step, &h[j], &xp[j]);
      oracle::so3::convert6DOFParameterToMatrix<double>(xp, T16_plus + 16 * j);
    }
  }
  return 0;
}

// tst/state_model.cpp:83-112: StateModel(x_init) under CostFunctionNumericalDynamic(model, 15, 15, 1).
// cov: 15 x 15 column-major or NULL.  H: 225 column-major, b: 15.
int oracle_state_linearize(const double *x_init, const double *x, const double *cov, int loss_kind,
                           double loss_param, double *H, double *b, double *cost) {
  try {
    auto c = makeCost<double>(kNumericDyn, std::make_shared<oracle::StateModel>(x_init), 15, 15, 1, cov,
                              loss_kind, loss_param);
    *cost = c->linearize(x, H, b);
    return 0;
  } catch (...) {
    return -2;
  }
}

int oracle_state_cost(const double *x_init, const double *x, double *cost) {
  try {
    auto c = makeCost<double>(kNumericDyn, std::make_shared<oracle::StateModel>(x_init), 15, 15, 1,
                              nullptr, 0, 0.0);
    *cost = c->computeCost(x);
    return 0;
  } catch (...) {
    return -2;
  }
}

// LevenbergMarquadtDynamic<double> lm(15); lm.addCost(&cost); lm.minimize(x)  (:99, :108-109)
int oracle_state_minimize(const double *x_init, double *x, int max_iter, int *status, int *iterations) {
  try {
    auto c = makeCost<double>(kNumericDyn, std::make_shared<oracle::StateModel>(x_init), 15, 15, 1,
                              nullptr, 0, 0.0);
    moptimizer::LevenbergMarquadtDynamic<double> lm(15);
    lm.setMaximumIterations(max_iter);
    lm.addCost(c.get());
    *status = int(lm.minimize(x));
    *iterations = int(lm.getExecutedIterations());
    return 0;
  } catch (...) {
    return -2;
  }
}

int oracle_hardware_threads(void) { return int(std::thread::hardware_concurrency()); }

// compiler and flags this checker was built with (oracle/Makefile), for bench.py's cpu_baseline
#ifndef ORACLE_BUILD_FLAGS
#define ORACLE_BUILD_FLAGS "unknown"
#endif
const char *oracle_build_flags(void) { return "g++ " __VERSION__ " " ORACLE_BUILD_FLAGS; }

}  // extern "C"
