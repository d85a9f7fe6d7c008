// HIP-backed cost functions: moptimizer::CostFunctionBase<Scalar> implementations whose
// linearize() / computeCost() sweeps run on an MI355X through the C ABI of
// libmoptimizer_hip.so.  They are what a user adds to LevenbergMarquadtDynamic in place of
// CostFunctionAnalyticalDynamic / CostFunctionNumericalDynamic
// (/root/reference/include/moptimizer/cost_function_analytical_dyn.h:11-32,
//  cost_function_numerical_dyn.h:12-34): same constructor shape (model, n, m, N), same virtuals,
// same buffer conventions, so the LM loop (src/levenberg_marquadt_dyn.cpp:48-60,86) is unchanged.
//
// The reference evaluates user models through per-index virtual calls; device code cannot call
// into host objects, so the models of the path are *device models*: host-side descriptors
// (Point2PointDeviceModel, ReprojectionDeviceModel) naming a kernel family and the user-owned
// arrays.  A cost constructed with any other IBaseModel throws — there is no CPU fallback.
//
// The header only needs the CostFunctionBase / IBaseModel / covariance declarations; in the
// reference tree include <moptimizer/cost_function.h> first and define
// MOPTIMIZER_AMD_USE_REFERENCE_HEADERS.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#ifndef MOPTIMIZER_AMD_USE_REFERENCE_HEADERS
#include "moptimizer_amd/host_api.hpp"
#endif
#include "moptimizer_hip.h"

namespace moptimizer {
namespace hip {

// Geman-McClure threshold of a loss object: through its accessor when the class has one (this
// repository's host_api.hpp), otherwise recovered from one probe of weight() — the reference class
// keeps the threshold private, and w(1) = t^2 / (1 + t)^2 gives t = sqrt(w) / (1 - sqrt(w)).
template <class Loss>
auto gemanMcClureThreshold(Loss *gm, int) -> decltype(double(gm->threshold())) {
  return double(gm->threshold());
}
template <class Loss>
double gemanMcClureThreshold(Loss *gm, long) {
  const double r = std::sqrt(double(gm->weight(1)));
  return r / (1.0 - r);
}

inline void throwOnError(int rc, const char *where) {
  if (rc != MOPT_OK)
    throw moptimizer::Exception(std::string(where) + " failed (" + std::to_string(rc) +
                                "): " + mopt_last_error());
}

// ---- device models --------------------------------------------------------------------------
// Common base: a model whose f / f_df live on the GPU.  The per-index host virtuals exist only to
// satisfy IBaseModel and refuse to run.
template <typename Scalar>
class DeviceModel : public IBaseModel<Scalar> {
 public:
  void setup(const Scalar *) override {}
  void update(const Scalar *) override {}
  bool f(const Scalar *, Scalar *, unsigned int) const override {
    throw Exception("device model: f() is evaluated on the GPU, not per index on the host");
  }
  bool f_df(const Scalar *, Scalar *, Scalar *, unsigned int) const override {
    throw Exception("device model: f_df() is evaluated on the GPU, not per index on the host");
  }
  // creates the mopt_cost for this model on `device`
  virtual mopt_cost *createDeviceCost(int device, int num_residuals) const = 0;
  // creates the sharded form over several devices of this node; models whose data
  // do not shard by index range keep the default
  virtual mopt_group *createDeviceGroup(const std::vector<int> &, int) const {
    throw Exception("this device model has no multi-device form");
  }
  virtual int numOutputs() const = 0;
  virtual int numParameters() const { return 6; }
  // The cost hands its device handle to the model so that model->update(x) can act on it.
  virtual void attach(mopt_cost *) {}
};

// The Point2Point model of tst/point2point.cpp:24-84: residual R(x) p + t(x) - q over
// index-aligned clouds given as packed xyz triples (e.g. std::vector<Eigen::Vector3d>::data()).
template <typename Scalar>
class Point2PointDeviceModel : public DeviceModel<Scalar> {
 public:
  using Ptr = std::shared_ptr<Point2PointDeviceModel>;
  Point2PointDeviceModel(const Scalar *src_xyz, const Scalar *tgt_xyz, std::size_t count)
      : src_(src_xyz), tgt_(tgt_xyz), count_(count) {}
  typename IBaseModel<Scalar>::Ptr clone() const override {
    return std::make_shared<Point2PointDeviceModel>(*this);
  }
  mopt_cost *createDeviceCost(int device, int num_residuals) const override {
    if (num_residuals < 0 || std::size_t(num_residuals) > count_)
      throw Exception("Point2PointDeviceModel: num_residuals exceeds the cloud size");
    mopt_cost *h = nullptr;
    throwOnError(mopt_point2point_create(&h, device, int(sizeof(Scalar)), src_, tgt_,
                                         num_residuals, MOPT_INPUT_HOST),
                 "mopt_point2point_create");
    return h;
  }
  mopt_group *createDeviceGroup(const std::vector<int> &devices, int num_residuals) const override {
    if (num_residuals < 0 || std::size_t(num_residuals) > count_)
      throw Exception("Point2PointDeviceModel: num_residuals exceeds the cloud size");
    mopt_group *g = nullptr;
    throwOnError(mopt_group_point2point_create(&g, devices.data(), int(devices.size()),
                                               int(sizeof(Scalar)), src_, tgt_, num_residuals),
                 "mopt_group_point2point_create");
    return g;
  }
  int numOutputs() const override { return 3; }

 private:
  const Scalar *src_;
  const Scalar *tgt_;
  std::size_t count_;
};

// Point-to-point ICP with correspondence search: update(x) — which the optimizer calls through
// cost->update(x) at the top of every outer iteration (levenberg_marquadt_dyn.cpp:54) — re-matches
// every source point to its nearest target under the current pose, on the GPU.
template <typename Scalar>
class IcpDeviceModel : public DeviceModel<Scalar> {
 public:
  using Ptr = std::shared_ptr<IcpDeviceModel>;
  IcpDeviceModel(const Scalar *src_xyz, std::size_t num_src, const Scalar *tgt_xyz,
                 std::size_t num_tgt, double max_distance)
      : src_(src_xyz), tgt_(tgt_xyz), num_src_(num_src), num_tgt_(num_tgt),
        max_distance_(max_distance) {}
  typename IBaseModel<Scalar>::Ptr clone() const override {
    return std::make_shared<IcpDeviceModel>(*this);
  }
  mopt_cost *createDeviceCost(int device, int num_residuals) const override {
    if (num_residuals < 0 || std::size_t(num_residuals) > num_src_)
      throw Exception("IcpDeviceModel: num_residuals exceeds the source cloud size");
    mopt_cost *h = nullptr;
    throwOnError(mopt_icp_create(&h, device, int(sizeof(Scalar)), src_, num_residuals, tgt_,
                                 std::int64_t(num_tgt_), max_distance_),
                 "mopt_icp_create");
    return h;
  }
  int numOutputs() const override { return 3; }
  void attach(mopt_cost *handle) override { handle_ = handle; }
  void update(const Scalar *x) override {
    if (handle_) throwOnError(mopt_icp_update(handle_, x, nullptr), "mopt_icp_update");
  }

 private:
  const Scalar *src_;
  const Scalar *tgt_;
  std::size_t num_src_, num_tgt_;
  double max_distance_;
  mopt_cost *handle_ = nullptr;
};

// The CameraModel of tst/camera_calibration.cpp:12-57 (fp64, no Jacobian).
class ReprojectionDeviceModel : public DeviceModel<double> {
 public:
  using Ptr = std::shared_ptr<ReprojectionDeviceModel>;
  ReprojectionDeviceModel(const double *points_xyzw, const std::int32_t *pixels_uv,
                          std::size_t count)
      : points_(points_xyzw), pixels_(pixels_uv), count_(count) {
    if (count == 0) throw std::runtime_error("Empty point list");
  }
  IBaseModel<double>::Ptr clone() const override {
    return std::make_shared<ReprojectionDeviceModel>(*this);
  }
  mopt_cost *createDeviceCost(int device, int num_residuals) const override {
    if (num_residuals <= 0 || std::size_t(num_residuals) > count_)
      throw Exception("ReprojectionDeviceModel: bad num_residuals");
    mopt_cost *h = nullptr;
    throwOnError(mopt_reprojection_create(&h, device, points_, pixels_, num_residuals, nullptr,
                                          nullptr, MOPT_INPUT_HOST),
                 "mopt_reprojection_create");
    return h;
  }
  int numOutputs() const override { return 2; }

 private:
  const double *points_;
  const std::int32_t *pixels_;
  std::size_t count_;
};

// The small parametric models of the reference's other tests (n != 6), as device models.
// y - exp(x0 t + x1) over interleaved (t, y) pairs — CurveFittingModel, tst/curve_fitting.cpp:81-98.
class ExpCurveDeviceModel : public DeviceModel<double> {
 public:
  using Ptr = std::shared_ptr<ExpCurveDeviceModel>;
  explicit ExpCurveDeviceModel(const double *interleaved_ty) : data_(interleaved_ty) {}
  IBaseModel<double>::Ptr clone() const override {
    return std::make_shared<ExpCurveDeviceModel>(*this);
  }
  mopt_cost *createDeviceCost(int device, int num_residuals) const override {
    mopt_cost *h = nullptr;
    throwOnError(mopt_scalar_model_create(&h, device, 8, MOPT_MODEL_EXP_CURVE, data_, data_ + 1, 2,
                                          num_residuals),
                 "mopt_scalar_model_create");
    return h;
  }
  int numOutputs() const override { return 1; }
  int numParameters() const override { return 2; }

 private:
  const double *data_;
};

// y - x0 t / (x1 + t) over separate t and y arrays — `Model`, tst/test_models.h:7-20.
template <typename Scalar>
class RationalDeviceModel : public DeviceModel<Scalar> {
 public:
  using Ptr = std::shared_ptr<RationalDeviceModel>;
  RationalDeviceModel(const Scalar *t, const Scalar *y) : t_(t), y_(y) {}
  typename IBaseModel<Scalar>::Ptr clone() const override {
    return std::make_shared<RationalDeviceModel>(*this);
  }
  mopt_cost *createDeviceCost(int device, int num_residuals) const override {
    mopt_cost *h = nullptr;
    throwOnError(mopt_scalar_model_create(&h, device, int(sizeof(Scalar)), MOPT_MODEL_RATIONAL, t_,
                                          y_, 1, num_residuals),
                 "mopt_scalar_model_create");
    return h;
  }
  int numOutputs() const override { return 1; }
  int numParameters() const override { return 2; }

 private:
  const Scalar *t_;
  const Scalar *y_;
};

// Powell's singular function — PowellModel, tst/powell.cpp:21-60 (one residual block of 4).
class PowellDeviceModel : public DeviceModel<double> {
 public:
  using Ptr = std::shared_ptr<PowellDeviceModel>;
  IBaseModel<double>::Ptr clone() const override {
    return std::make_shared<PowellDeviceModel>(*this);
  }
  mopt_cost *createDeviceCost(int device, int num_residuals) const override {
    mopt_cost *h = nullptr;
    throwOnError(mopt_scalar_model_create(&h, device, 8, MOPT_MODEL_POWELL, nullptr, nullptr, 1,
                                          num_residuals),
                 "mopt_scalar_model_create");
    return h;
  }
  int numOutputs() const override { return 4; }
  int numParameters() const override { return 4; }
};

/// A model written by the user: the device counterpart of deriving from BaseModel /
/// BaseModelJacobian (model.h:29-47).  The bodies are the statements of
///   void setup(const Scalar *x, Scalar *a)                                    // optional: num_aux per-x values
///   void f    (const Scalar *x, const Scalar *a, const Scalar *d, Scalar *r)  // r: num_outputs values
///   void f_df (const Scalar *x, const Scalar *a, const Scalar *d, Scalar *J)  // J: num_outputs x n, row-major
/// in HIP C++ (`S` names the scalar type); d holds the element's value from each data plane and
/// `a` what setup computed for this parameter vector (IBaseModel::setup, model.h:19-22).
/// Compiled for the GPU when the cost function is constructed; a model without a Jacobian body
/// behaves as BaseModel does (f_df throws).
template <typename Scalar>
class JitDeviceModel : public DeviceModel<Scalar> {
 public:
  using Ptr = std::shared_ptr<JitDeviceModel>;
  JitDeviceModel(int num_parameters, int num_outputs, std::string residual_body,
                 std::string jacobian_body, std::vector<const Scalar *> planes, int num_aux = 0,
                 std::string setup_body = std::string())
      : n_(num_parameters), m_(num_outputs), aux_(num_aux), residual_(std::move(residual_body)),
        jacobian_(std::move(jacobian_body)), setup_(std::move(setup_body)),
        planes_(std::move(planes)) {}
  typename IBaseModel<Scalar>::Ptr clone() const override {
    return std::make_shared<JitDeviceModel>(*this);
  }
  mopt_cost *createDeviceCost(int device, int num_residuals) const override {
    // the planes are separate user arrays: gather them into one staging block
    std::vector<Scalar> staged(planes_.size() * std::size_t(num_residuals));
    for (std::size_t p = 0; p < planes_.size(); ++p)
      std::copy(planes_[p], planes_[p] + num_residuals, staged.begin() + p * num_residuals);
    mopt_cost *h = nullptr;
    throwOnError(mopt_jit_model_create(&h, device, int(sizeof(Scalar)), n_, m_, int(planes_.size()),
                                       aux_, setup_.empty() ? nullptr : setup_.c_str(),
                                       residual_.c_str(), jacobian_.empty() ? nullptr : jacobian_.c_str(),
                                       staged.data(), num_residuals, num_residuals, MOPT_INPUT_HOST),
                 "mopt_jit_model_create");
    return h;
  }
  int numOutputs() const override { return m_; }
  int numParameters() const override { return n_; }

 private:
  int n_, m_, aux_;
  std::string residual_, jacobian_, setup_;
  std::vector<const Scalar *> planes_;
};

// ---- cost functions ---------------------------------------------------------------------------
// What the device-resident LM loop (levenberg_marquadt_device.hpp) needs to know of a cost it is
// handed as a CostFunctionBase*: the device object behind it and which Jacobian its class stands for.
class DeviceCostAccess {
 public:
  virtual ~DeviceCostAccess() = default;
  virtual mopt_cost *deviceCost() const = 0;  // nullptr for a cost sharded over a device group
  virtual int deviceJacobianMode() const = 0;
  virtual void syncDeviceState() = 0;  // forward the current loss / covariance to the device
};

// The costs an optimizer is going to hold (`optimizer.addCost(&a); optimizer.addCost(&b);`): its loop
// asks them one after the other at the same x (levenberg_marquadt_dyn.cpp:52-59, :86).  Linked, the
// first one asked queues the others' sweeps as well (mopt_costs_link): one launch path per evaluated
// point instead of one per cost.  The loop itself is unchanged.  Costs that are not HIP costs, or
// are sharded over a device group, are left out.
template <class Scalar>
inline void linkCosts(std::initializer_list<CostFunctionBase<Scalar> *> costs) {
  std::vector<mopt_cost *> handles;
  for (CostFunctionBase<Scalar> *c : costs) {
    auto *access = dynamic_cast<DeviceCostAccess *>(c);
    if (access && access->deviceCost()) handles.push_back(access->deviceCost());
  }
  throwOnError(mopt_costs_link(handles.data(), int(handles.size())), "mopt_costs_link");
}

// Shared implementation; JacobianMode selects what linearize() means.
template <class Scalar, int JacobianMode>
class CostFunctionHip : public CostFunctionBase<Scalar>, public DeviceCostAccess {
 public:
  using Base = CostFunctionBase<Scalar>;
  using typename Base::ModelPtr;

  CostFunctionHip(ModelPtr model, int num_parameters, int num_outputs, int num_residuals,
                  int device = 0)
      : Base(model, num_residuals), num_parameters_(num_parameters), num_outputs_(num_outputs) {
    auto *dm = dynamic_cast<DeviceModel<Scalar> *>(model.get());
    if (!dm)
      throw Exception(
          "CostFunctionHip needs a device model (Point2PointDeviceModel / "
          "ReprojectionDeviceModel); arbitrary host IBaseModel objects cannot run on the GPU");
    if (num_parameters != dm->numParameters() || num_outputs != dm->numOutputs())
      throw Exception("CostFunctionHip: (num_parameters, num_outputs) do not match the device model");
    handle_ = dm->createDeviceCost(device, num_residuals);
    dm->attach(handle_);
    this->covariance_->resize(num_outputs_, num_outputs_);
    this->covariance_->setIdentity();
  }
  // The same cost sharded over several GPUs of the node (contiguous index ranges, one RCCL
  // all-reduce of n*n + n + 1 doubles per sweep); the single-process LM loop sees one cost.
  CostFunctionHip(ModelPtr model, int num_parameters, int num_outputs, int num_residuals,
                  const std::vector<int> &devices)
      : Base(model, num_residuals), num_parameters_(num_parameters), num_outputs_(num_outputs) {
    auto *dm = dynamic_cast<DeviceModel<Scalar> *>(model.get());
    if (!dm) throw Exception("CostFunctionHip needs a device model");
    if (num_parameters != dm->numParameters() || num_outputs != dm->numOutputs())
      throw Exception("CostFunctionHip: (num_parameters, num_outputs) do not match the device model");
    if (devices.empty()) throw Exception("CostFunctionHip: empty device list");
    group_ = dm->createDeviceGroup(devices, num_residuals);
    this->covariance_->resize(num_outputs_, num_outputs_);
    this->covariance_->setIdentity();
  }
  ~CostFunctionHip() override {
    if (handle_) mopt_cost_destroy(handle_);
    if (group_) mopt_group_destroy(group_);
  }

  Scalar computeCost(const Scalar *x) override {
    Scalar sum = 0;
    if (group_)
      throwOnError(mopt_group_compute(group_, x, &sum), "mopt_group_compute");
    else
      throwOnError(mopt_cost_compute(handle_, x, &sum), "mopt_cost_compute");
    return sum;
  }

  Scalar linearize(const Scalar *x, Scalar *hessian, Scalar *b) override {
    pushState();
    Scalar sum = 0;
    if (group_)
      throwOnError(mopt_group_linearize(group_, JacobianMode, x, hessian, b, &sum),
                   "mopt_group_linearize");
    else
      throwOnError(mopt_cost_linearize(handle_, JacobianMode, x, hessian, b, &sum),
                   "mopt_cost_linearize");
    return sum;
  }

  mopt_cost *handle() const { return handle_; }
  mopt_group *group() const { return group_; }
  mopt_cost *deviceCost() const override { return handle_; }
  int deviceJacobianMode() const override { return JacobianMode; }
  void syncDeviceState() override { pushState(); }

 protected:
  // setLossFunction / setCovariance are non-virtual setters on the base (cost_function.h:37-40),
  // so the current loss and covariance are forwarded at the start of every sweep.
  void pushState() {
    throwOnError(group_ ? mopt_group_set_covariance(group_, this->covariance_->data())
                        : mopt_cost_set_covariance(handle_, this->covariance_->data()),
                 "set_covariance");
    auto *gm = dynamic_cast<loss::GemmanMCClure<Scalar> *>(this->loss_function_.get());
    int kind = MOPT_LOSS_NONE;
    double parameter = 0.0;
    if (gm) {
      kind = MOPT_LOSS_GEMAN_MCCLURE;
      parameter = gemanMcClureThreshold(gm, 0);
    } else if (!dynamic_cast<loss::NoLoss<Scalar> *>(this->loss_function_.get())) {
      throw Exception("CostFunctionHip: only NoLoss and GemmanMCClure have device kernels");
    }
    throwOnError(group_ ? mopt_group_set_loss(group_, kind, parameter)
                        : mopt_cost_set_loss(handle_, kind, parameter),
                 "set_loss");
  }

  int num_parameters_;
  int num_outputs_;
  mopt_cost *handle_ = nullptr;
  mopt_group *group_ = nullptr;
};

// Drop-in for CostFunctionAnalyticalDynamic: model-supplied, API-conformant row-major Jacobian.
template <class Scalar = double>
using CostFunctionAnalyticalHip = CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC>;
// Bit-faithful to tst/point2point.cpp's Point2Point::f_df as written.
template <class Scalar = double>
using CostFunctionAnalyticalTstLayoutHip = CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC_TST_LAYOUT>;
// Point2point with the Jacobian of a left perturbation of the pose, [I | -skew(R p + t)]: the cost
// to use with an SE(3) manifold update (LevenbergMarquadtDevice::setManifoldUpdate; the reference
// leaves that update as a TODO, src/levenberg_marquadt_dyn.cpp:82-83).
template <class Scalar = double>
using CostFunctionAnalyticalLeftHip = CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC_LEFT>;
// The same for a right perturbation of the rotation, [I | -R skew(p)] (the composition of
// tst/manifold.cpp:47 / tst/state_model.cpp:28-34; LevenbergMarquadtDevice::setRightManifoldUpdate).
template <class Scalar = double>
using CostFunctionAnalyticalRightHip = CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC_RIGHT>;
// Drop-in for CostFunctionNumericalDynamic: forward differences.
template <class Scalar = double>
using CostFunctionNumericalHip = CostFunctionHip<Scalar, MOPT_JAC_NUMERIC>;

// The reference's four class names, one namespace down: switching a cost to the GPU is
// `moptimizer::CostFunctionNumerical<double, 6, 3>` -> `moptimizer::hip::CostFunctionNumerical<...>`
// with the same constructor arguments (plus an optional device index).
//   cost_function_analytical_dyn.h:17-18 / cost_function_numerical_dyn.h:19-20: (model, n, m, N)
//   cost_function_analytical.h:21-25 / cost_function_numerical.h:24-28:          (model, N)
template <class Scalar = double>
using CostFunctionAnalyticalDynamic = CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC>;
template <class Scalar = double>
using CostFunctionNumericalDynamic = CostFunctionHip<Scalar, MOPT_JAC_NUMERIC>;

template <class Scalar = double, int model_parameter_dim = 1, int model_output_dim = 1>
class CostFunctionAnalytical : public CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC> {
 public:
  using typename CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC>::ModelPtr;
  CostFunctionAnalytical(ModelPtr model, int num_residuals, int device = 0)
      : CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC>(model, model_parameter_dim, model_output_dim,
                                                   num_residuals, device) {}
  CostFunctionAnalytical(ModelPtr model, int num_residuals, const std::vector<int> &devices)
      : CostFunctionHip<Scalar, MOPT_JAC_ANALYTIC>(model, model_parameter_dim, model_output_dim,
                                                   num_residuals, devices) {}
};

template <class Scalar = double, int model_parameter_dim = 1, int model_output_dim = 1>
class CostFunctionNumerical : public CostFunctionHip<Scalar, MOPT_JAC_NUMERIC> {
 public:
  using typename CostFunctionHip<Scalar, MOPT_JAC_NUMERIC>::ModelPtr;
  CostFunctionNumerical(ModelPtr model, int num_residuals, int device = 0)
      : CostFunctionHip<Scalar, MOPT_JAC_NUMERIC>(model, model_parameter_dim, model_output_dim,
                                                  num_residuals, device) {}
  CostFunctionNumerical(ModelPtr model, int num_residuals, const std::vector<int> &devices)
      : CostFunctionHip<Scalar, MOPT_JAC_NUMERIC>(model, model_parameter_dim, model_output_dim,
                                                  num_residuals, devices) {}
};

}  // namespace hip
}  // namespace moptimizer
