// ORACLE — TEST INFRASTRUCTURE ONLY (see cost_computation.hpp).
//
// The reference's four CPU cost classes, restated over oracle::CostComputation:
//   CostFunctionAnalyticalDynamic  /root/reference/include/moptimizer/cost_function_analytical_dyn.h:11-32,
//                                  src/cost_function_analytical_dyn.cpp:7-33
//   CostFunctionNumericalDynamic   include/moptimizer/cost_function_numerical_dyn.h:12-34,
//                                  src/cost_function_numerical_dyn.cpp:7-33
//   CostFunctionAnalytical<S,n,m>  include/moptimizer/cost_function_analytical.h:15-46
//   CostFunctionNumerical<S,n,m>   include/moptimizer/cost_function_numerical.h:15-46
// Each call builds a fresh CostComputation, as the originals do; the covariance defaults to the
// m x m identity; computeCost goes through the parallel sweep.  The compile-time-dimension twins
// perform the same arithmetic (only Eigen's storage differs in the reference), so they forward
// to the run-time-dimension computation.
#pragma once

#include "cost_computation.hpp"

namespace oracle {

template <class Scalar = double>
class CostFunctionAnalyticalDynamic : public moptimizer::CostFunctionBase<Scalar> {
 public:
  using Base = moptimizer::CostFunctionBase<Scalar>;
  using typename Base::ModelPtr;

  CostFunctionAnalyticalDynamic(ModelPtr model, int num_parameters, int num_outputs,
                                int num_residuals)
      : Base(model, num_residuals), num_parameters_(num_parameters), num_outputs_(num_outputs) {
    this->covariance_->resize(num_outputs_, num_outputs_);
    this->covariance_->setIdentity();
  }

  Scalar computeCost(const Scalar *x) override {
    CostComputation<Scalar> compute(num_parameters_, num_outputs_);
    return compute.parallelComputeCost(x, this->model_, this->num_residuals_);
  }
  Scalar linearize(const Scalar *x, Scalar *hessian, Scalar *b) override {
    CostComputation<Scalar> compute(num_parameters_, num_outputs_);
    return compute.computeHessian(x, this->covariance_->data(), this->loss_function_, hessian, b,
                                  this->model_, this->num_residuals_);
  }

 protected:
  int num_parameters_;
  int num_outputs_;
};

template <class Scalar = double>
class CostFunctionNumericalDynamic : public moptimizer::CostFunctionBase<Scalar> {
 public:
  using Base = moptimizer::CostFunctionBase<Scalar>;
  using typename Base::ModelPtr;

  CostFunctionNumericalDynamic(ModelPtr model, int num_parameters, int num_outputs,
                               int num_residuals)
      : Base(model, num_residuals), num_parameters_(num_parameters), num_outputs_(num_outputs) {
    this->covariance_->resize(num_outputs_, num_outputs_);
    this->covariance_->setIdentity();
  }

  Scalar computeCost(const Scalar *x) override {
    CostComputation<Scalar> compute(num_parameters_, num_outputs_);
    return compute.parallelComputeCost(x, this->model_, this->num_residuals_);
  }
  Scalar linearize(const Scalar *x, Scalar *hessian, Scalar *b) override {
    CostComputation<Scalar> compute(num_parameters_, num_outputs_);
    return compute.computeHessianNumerical(x, this->covariance_->data(), this->loss_function_,
                                           hessian, b, this->model_, this->num_residuals_);
  }

 protected:
  int num_parameters_;
  int num_outputs_;
};

template <class Scalar = double, int model_parameter_dim = 1, int model_output_dim = 1>
class CostFunctionAnalytical : public CostFunctionAnalyticalDynamic<Scalar> {
 public:
  using typename CostFunctionAnalyticalDynamic<Scalar>::ModelPtr;
  CostFunctionAnalytical(ModelPtr model, int num_residuals)
      : CostFunctionAnalyticalDynamic<Scalar>(model, model_parameter_dim, model_output_dim,
                                              num_residuals) {}
};

template <class Scalar = double, int model_parameter_dim = 1, int model_output_dim = 1>
class CostFunctionNumerical : public CostFunctionNumericalDynamic<Scalar> {
 public:
  using typename CostFunctionNumericalDynamic<Scalar>::ModelPtr;
  CostFunctionNumerical(ModelPtr model, int num_residuals)
      : CostFunctionNumericalDynamic<Scalar>(model, model_parameter_dim, model_output_dim,
                                             num_residuals) {}
};

}  // namespace oracle
