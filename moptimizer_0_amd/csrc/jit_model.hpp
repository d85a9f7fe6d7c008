// Run-time compiled device models (hipRTC): internal interface between c_abi.cpp and jit_model.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include "sweep.hpp"

namespace mopt {

struct JitKernel {
  hipModule_t module = nullptr;
  hipFunction_t sweep = nullptr;
  bool has_jacobian = false;
};

// Kernel argument block; the device-side declaration in jit_model.cpp has the same members in the
// same order (both sides use natural alignment).
template <typename S>
struct JitArgs {
  const S *data;
  long long count;
  long long stride;
  int loss_kind;
  int numeric;
  int cost_only;
  int pad_;
  S loss_param;
  S x[8];
  S h[8];
  S cov[16];
  double *partials;
};

bool jitCompile(int scalar_bytes, int n_params, int n_outputs, int n_planes,
                const char *residual_body, const char *jacobian_body, JitKernel &out);
void jitRelease(JitKernel &k);
hipError_t jitLaunch(const JitKernel &k, const void *args, size_t args_bytes, int grid,
                     hipStream_t stream);
const char *jitLastError();

}  // namespace mopt
