// LevenbergMarquadtDynamic with the iteration resident on the GPU.
//
// Same surface as the reference's optimizer (include/moptimizer/optimizer.h:12-89,
// include/moptimizer/levenberg_marquadt_dyn.h:8-50): addCost / clearCosts / setMaximumIterations /
// setLevenbergMarquadtIterations / minimize / getExecutedIterations, same OptimizationStatus
// values, same arithmetic (src/levenberg_marquadt_dyn.cpp:34-119) — but minimize() hands the whole
// loop to mopt_lm_minimize: the damped solve, the trial point, the gain ratio and the stopping
// tests run in a one-workgroup kernel behind every sweep, sweeps are queued ahead, and the host
// sees only the final x.  It accepts the HIP cost classes of cost_function_hip.hpp (a host cost
// has nothing to run on the device and is refused).  Where the optimizer must stay the reference's
// own class, the boundary CostFunctionBase::linearize / computeCost is unchanged and that class
// keeps working; this one is the additive fast path.
#pragma once

#include <stdexcept>
#include <vector>

#include "moptimizer_amd/cost_function_hip.hpp"

namespace moptimizer {
namespace hip {

template <class Scalar = double>
class LevenbergMarquadtDevice {
 public:
  using CostFunctionType = CostFunctionBase<Scalar>;

  explicit LevenbergMarquadtDevice(int num_parameters) : num_parameters_(num_parameters) {}
  LevenbergMarquadtDevice(const LevenbergMarquadtDevice &) = delete;
  LevenbergMarquadtDevice &operator=(const LevenbergMarquadtDevice &) = delete;

  void setMaximumIterations(int max_iterations) {  // optimizer.h:33-37
    if (max_iterations < 0)
      throw std::invalid_argument("Optimization::max_iterations cannot be less than 0.");
    maximum_iterations_ = static_cast<unsigned int>(max_iterations);
  }
  unsigned int getMaximumIterations() const { return maximum_iterations_; }
  unsigned int getExecutedIterations() const { return executed_iterations_; }
  unsigned int getLevenbergMarquadtIterations() const { return lm_max_iterations_; }
  void setLevenbergMarquadtIterations(int max_iterations) { lm_max_iterations_ = max_iterations; }
  // xi = x0 (+) delta on SE(3) (mopt_se3_plus) instead of the reference's xi = x0 + delta
  // (levenberg_marquadt_dyn.cpp:82-83, "TODO Manifold operation"); 6-parameter poses, to be used
  // with CostFunctionAnalyticalLeftHip costs.
  void setManifoldUpdate(bool on) { manifold_update_ = on ? 1 : 0; }
  // The same composed on the right — R <- R Exp(delta_w), t <- t + delta_t, the form of the
  // reference's own sketches (tst/manifold.cpp:47, tst/state_model.cpp:28-34) — to be used with
  // CostFunctionAnalyticalRightHip costs.
  void setRightManifoldUpdate(bool on) { manifold_update_ = on ? 2 : 0; }

  // Non-owning, as Optimizer::addCost (optimizer.h:58).
  void addCost(CostFunctionType *cost) {
    auto *access = dynamic_cast<DeviceCostAccess *>(cost);
    if (!access || !access->deviceCost())
      throw Exception(
          "LevenbergMarquadtDevice drives single-device HIP costs (moptimizer::hip::CostFunction*); "
          "use LevenbergMarquadtDynamic for host costs and device groups");
    costs_.push_back(cost);
    access_.push_back(access);
  }
  void clearCosts(bool delete_costs = false) {  // optimizer.h:62-68
    if (delete_costs)
      for (auto *c : costs_) delete c;
    costs_.clear();
    access_.clear();
  }

  OptimizationStatus step(Scalar *) { return OptimizationStatus::NUMERIC_ERROR; }  // as the reference's stub

  OptimizationStatus minimize(Scalar *x0) {
    if (costs_.empty()) throw std::runtime_error("No cost function added!");  // optimizer.h:48-54
    std::vector<mopt_cost *> handles;
    std::vector<int> modes;
    for (std::size_t k = 0; k < costs_.size(); ++k) {
      costs_[k]->update(x0);  // the model's per-iteration hook: empty for every device model here
      access_[k]->syncDeviceState();
      handles.push_back(access_[k]->deviceCost());
      modes.push_back(access_[k]->deviceJacobianMode());
    }
    mopt_lm_options options;
    options.max_iterations = int(maximum_iterations_);
    options.lm_max_iterations = int(lm_max_iterations_);
    options.manifold = manifold_update_;
    options.window = 0;
    mopt_lm_report report;
    (void)num_parameters_;
    throwOnError(mopt_lm_minimize(handles.data(), int(handles.size()), modes.data(), x0, &options,
                                  &report),
                 "mopt_lm_minimize");
    executed_iterations_ = static_cast<unsigned int>(report.iterations);
    sweeps_ = report.sweeps;
    final_cost_ = report.cost;
    return static_cast<OptimizationStatus>(report.status);
  }

  long long sweeps() const { return sweeps_; }   // points evaluated by the last minimize()
  double finalCost() const { return final_cost_; }

 private:
  int num_parameters_;
  unsigned int maximum_iterations_ = 15;  // optimizer.h:19
  unsigned int lm_max_iterations_ = 3;    // levenberg_marquadt_dyn.cpp:9
  unsigned int executed_iterations_ = 0;
  int manifold_update_ = 0;  // mopt_lm_options.manifold
  long long sweeps_ = 0;
  double final_cost_ = 0.0;
  std::vector<CostFunctionType *> costs_;
  std::vector<DeviceCostAccess *> access_;
};

}  // namespace hip
}  // namespace moptimizer
