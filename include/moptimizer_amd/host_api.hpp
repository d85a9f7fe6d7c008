// Host-side plug-in surface of the linearization path, Eigen-free.
//
// These are the types the reference's optimizer talks to when it drives one linearization
// sweep; the HIP-backed cost classes (cost_function_hip.hpp) implement the same virtuals, so
// an LM loop written against the reference headers drives them without edits.
//
//   reference type                               file:line in /root/reference
//   ------------------------------------------   ------------------------------------------
//   moptimizer::Exception                        include/moptimizer/exception.h:7-19
//   moptimizer::OptimizationStatus               include/moptimizer/types.h:6-12
//   moptimizer::IBaseModel / BaseModel /
//     BaseModelJacobian                          include/moptimizer/model.h:11-104
//   moptimizer::loss::ILossFunction / NoLoss     include/moptimizer/loss_function/loss_function.h:7-23
//   moptimizer::loss::GemmanMCClure              include/moptimizer/loss_function/geman_mcclure.h:6-19
//   moptimizer::covariance::Matrix / MatrixPtr   include/moptimizer/covariance/covariance.h:10-13
//   moptimizer::CostFunctionBase                 include/moptimizer/cost_function.h:15-59
//
// Contracts kept: Jacobians are row-major m x n (model.h:35-42, tst/differentiation.cpp:31,102);
// hessian buffers are n x n column-major and fully overwritten; linearize() returns the
// unweighted sum of squared residuals (linearization.h:152,157).
#pragma once

#include <exception>
#include <memory>
#include <string>

#include "moptimizer_amd/dense.hpp"

namespace moptimizer {

class Exception : public std::exception {
 public:
  explicit Exception(const char *what_arg) : text_(what_arg) {}
  explicit Exception(const std::string &what_arg) : text_(what_arg) {}
  ~Exception() noexcept override = default;
  const char *what() const noexcept override { return text_.c_str(); }

 protected:
  std::string text_;
};

enum OptimizationStatus {
  CONVERGED,
  MAXIMUM_ITERATIONS_REACHED,
  SMALL_DELTA,
  NUMERIC_ERROR,
  FATAL_ERROR,
};

// ---------------------------------------------------------------------------------------
// Models: one residual block (m outputs) per index, optionally with its m x n Jacobian.
// ---------------------------------------------------------------------------------------
template <typename Scalar>
class IBaseModel {
 public:
  using Ptr = std::shared_ptr<IBaseModel>;
  using ConstPtr = std::shared_ptr<const IBaseModel>;
  virtual ~IBaseModel() = default;

  // once per parameter vector (e.g. x -> SE(3) matrix)
  virtual void setup(const Scalar *x) = 0;
  // once per outer optimizer iteration (e.g. correspondence search)
  virtual void update(const Scalar *x) = 0;
  // residual block `index` at x; false = skip this index
  virtual bool f(const Scalar *x, Scalar *f_x, unsigned int index) const = 0;
  // residual block and row-major Jacobian; false = skip this index
  virtual bool f_df(const Scalar *x, Scalar *f_x, Scalar *jacobian, unsigned int index) const = 0;
  virtual Ptr clone() const = 0;
};

// Residual-only model; asking it for a Jacobian is an error.
template <typename Scalar, class Derived>
class BaseModel : public IBaseModel<Scalar> {
 public:
  using Ptr = std::shared_ptr<Derived>;
  using ConstPtr = std::shared_ptr<const Derived>;

  void setup(const Scalar *) override {}
  void update(const Scalar *) override {}
  bool f(const Scalar *x, Scalar *f_x, unsigned int index) const override = 0;
  bool f_df(const Scalar *, Scalar *, Scalar *, unsigned int) const final {
    throw Exception("Non implemented non-jacobian model function `f_df` being used.");
  }
  std::shared_ptr<IBaseModel<Scalar>> clone() const override {
    return std::make_shared<Derived>(*static_cast<const Derived *>(this));
  }
};

// Model that supplies its Jacobian; f() alone is optional.
template <typename Scalar, class Derived>
class BaseModelJacobian : public IBaseModel<Scalar> {
 public:
  using Ptr = std::shared_ptr<Derived>;
  using ConstPtr = std::shared_ptr<const Derived>;

  void setup(const Scalar *) override {}
  void update(const Scalar *) override {}
  bool f(const Scalar *, Scalar *, unsigned int) const override {
    throw Exception("Non implemented jacobian model function `f` being used.");
  }
  bool f_df(const Scalar *x, Scalar *f_x, Scalar *jacobian,
            unsigned int index) const override = 0;
  std::shared_ptr<IBaseModel<Scalar>> clone() const override {
    return std::make_shared<Derived>(*static_cast<const Derived *>(this));
  }
};

// ---------------------------------------------------------------------------------------
// Robust-loss weights w(s), s = squared residual norm.
// ---------------------------------------------------------------------------------------
namespace loss {

template <typename T>
class ILossFunction {
 public:
  using Ptr = std::shared_ptr<ILossFunction>;
  using ConstPtr = std::shared_ptr<const ILossFunction>;
  virtual ~ILossFunction() = default;
  virtual T weight(T errorSquaredNorm) = 0;
};

template <typename T>
class NoLoss : public ILossFunction<T> {
 public:
  T weight(T) override { return T(1); }
};

// w(s) = t^2 / (s + t)^2
template <typename T>
class GemmanMCClure : public ILossFunction<T> {
 public:
  using Ptr = std::shared_ptr<GemmanMCClure>;
  explicit GemmanMCClure(T threshold) : threshold_(threshold) {}
  T weight(T errorSquaredNorm) override {
    const T den = errorSquaredNorm + threshold_;
    return (threshold_ * threshold_) / (den * den);
  }
  T threshold() const { return threshold_; }

 private:
  T threshold_;
};

}  // namespace loss

// ---------------------------------------------------------------------------------------
// Output-space weighting matrix (m x m, column-major).
// ---------------------------------------------------------------------------------------
namespace covariance {
template <class Scalar>
using Matrix = dense::Matrix<Scalar>;
template <class Scalar>
using MatrixPtr = std::shared_ptr<Matrix<Scalar>>;
}  // namespace covariance

// ---------------------------------------------------------------------------------------
// What the optimizer sees of a cost: update / computeCost / linearize.
// ---------------------------------------------------------------------------------------
template <class Scalar = double>
class CostFunctionBase {
 public:
  using Model = IBaseModel<Scalar>;
  using ModelPtr = typename Model::Ptr;
  using ModelConstPtr = typename Model::ConstPtr;
  using LossFunctionPtr = typename loss::ILossFunction<Scalar>::Ptr;

  CostFunctionBase(ModelPtr model, int num_residuals)
      : num_residuals_(num_residuals),
        model_(std::move(model)),
        loss_function_(std::make_shared<loss::NoLoss<Scalar>>()),
        covariance_(std::make_shared<covariance::Matrix<Scalar>>()) {}
  CostFunctionBase() = delete;
  CostFunctionBase(const CostFunctionBase &) = delete;
  CostFunctionBase &operator=(const CostFunctionBase &) = delete;
  virtual ~CostFunctionBase() = default;

  void setLossFunction(LossFunctionPtr loss_function) { loss_function_ = std::move(loss_function); }
  void setCovariance(covariance::MatrixPtr<Scalar> covariance) {
    covariance_ = std::move(covariance);
  }

  // start of every outer iteration
  virtual void update(const Scalar *x) {
    if (model_) model_->update(x);
  }
  // sum over indices of r^T r at x
  virtual Scalar computeCost(const Scalar *x) = 0;
  // H = sum w J^T S J (n x n col-major), b = sum w J^T S r; returns sum r^T r
  virtual Scalar linearize(const Scalar *x, Scalar *hessian, Scalar *b) = 0;

 protected:
  int num_residuals_;
  ModelPtr model_;
  LossFunctionPtr loss_function_;
  covariance::MatrixPtr<Scalar> covariance_;
};

}  // namespace moptimizer
