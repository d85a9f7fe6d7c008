// Run-time compiled device models (hipRTC): internal interface between c_abi.cpp and jit_model.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "sweep.hpp"

namespace mopt {

struct JitVariant {
  hipModule_t module = nullptr;
  hipFunction_t sweep = nullptr;
  hipFunction_t sweep_resident = nullptr;  // arguments from HBM, early exit (device-resident LM)
  hipFunction_t sweep_resident_set = nullptr;  // several costs over this model in one launch
};

// One user model: its assembled source and the sweeps compiled from it so far.
struct JitKernel {
  std::string source;
  int scalar_bytes = 8, n_params = 0, n_outputs = 0, n_planes = 0, n_aux = 0;
  bool has_jacobian = false;
  bool wide = false;  // n > 8 or m > 4: the column-per-lane sweep (JitWideArgs, rows of n*n + n + 1)
  JitVariant variants[7];  // [0]: cost only; [1 + (mode - 1) * 3 + cov_mode]: mode 1 / 2 x CovMode
};

// Kernel argument block; the device-side declaration in jit_model.cpp has the same members in the
// same order (both sides use natural alignment).
template <typename S>
struct JitArgs {
  const S *data;
  long long count;
  long long stride;
  int loss_kind;
  int pad_[3];
  S loss_param;
  S x[8];
  S h[8];
  S cov[16];
  double *partials;
};

// The wide sweep's argument block (n <= 16, m <= 16).
template <typename S>
struct JitWideArgs {
  const S *data;
  long long count;
  long long stride;
  int loss_kind;
  int pad_[3];
  S loss_param;
  S x[kMaxWideParams];
  S h[kMaxWideParams];
  S cov[kMaxWideOutputs * kMaxWideOutputs];  // row-major M x M
  double *partials;
};
constexpr int kJitWideElementsPerBlock = 16;  // 256 threads = 16 elements x 16 column lanes

// Validates the shape, assembles the source and compiles the sweeps that exercise every user body.
bool jitCreate(int scalar_bytes, int n_params, int n_outputs, int n_planes, int n_aux,
               const char *setup_body, const char *residual_body, const char *jacobian_body,
               JitKernel &out);
// mode: 0 cost only, 1 supplied Jacobian, 2 forward differences; cov_mode: CovMode (identity /
// symmetric rows are the upper triangle, general rows the full matrix).  Compiled on first use;
// nullptr (and jitLastError) when that fails.
const JitVariant *jitVariant(JitKernel &k, int mode, int cov_mode);
void jitRelease(JitKernel &k);
hipError_t jitLaunch(const JitVariant &v, const void *args, size_t args_bytes, int grid,
                     hipStream_t stream);
// the resident entry point of the same module: (const JitArgs *d_args, const LmControl *control)
hipError_t jitLaunchResident(const JitVariant &v, const void *d_args, const LmControl *control,
                             int grid, hipStream_t stream);
// several costs created from the same source (same shape): one launch of the first one's module
hipError_t jitLaunchResidentSet(const JitVariant &v, const ResidentSweepSet &set,
                                const LmControl *control, hipStream_t stream);
const char *jitLastError();

}  // namespace mopt
