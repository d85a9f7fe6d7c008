// User-defined device models: the GPU counterpart of subclassing moptimizer::IBaseModel.
//
// In the reference a new model is a C++ class whose f() (and optionally f_df()) the sweep calls
// per index through a virtual (include/moptimizer/model.h:11-47).  Device code cannot call host
// virtuals, so here a model is the *text* of those two function bodies; hipRTC compiles it for
// gfx950 into the same per-element sweep the built-in models use (residual, forward-difference
// or supplied Jacobian, loss weight, w J^T S J / w J^T S r / r^T r accumulation, one partial row
// per workgroup).  The workgroup rows are finished by the library's finalizeDenseKernel.
#include "jit_model.hpp"

#include <hip/hiprtc.h>

#include <cstdio>
#include <string>
#include <vector>

namespace mopt {
namespace {

// Compiled at run time as  kJitPrologue + the two user functions + kJitSweep;  MOPT_S / MOPT_N /
// MOPT_M / MOPT_D arrive as -D options.
const char *const kJitPrologue = R"JIT(
#define kBlock 256
typedef MOPT_S S;
#define N MOPT_N
#define M MOPT_M
#define D MOPT_D
#define NACC (N * N + N + 1)

struct JitArgs {
  const S *data;       // planes: data[p * stride + i]
  long long count;
  long long stride;
  int loss_kind;       // 0 none, 1 Geman-McClure
  int numeric;         // 1: forward differences, 0: the supplied Jacobian
  int cost_only;
  int pad_;
  S loss_param;
  S x[8];
  S h[8];
  S cov[16];           // row-major M x M
  double *partials;    // [grid][NACC] (or [grid] when cost_only)
};

)JIT";

const char *const kJitSweep = R"JIT(
__device__ inline double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

extern "C" __global__ __launch_bounds__(kBlock) void mopt_jit_sweep(const JitArgs A) {
  __shared__ double lds[kBlock / 64][NACC];
  double acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < A.count;
       i += (long long)gridDim.x * kBlock) {
    S d[D > 0 ? D : 1];
    for (int p = 0; p < D; ++p) d[p] = A.data[p * A.stride + i];
    S r[M];
    user_residual(A.x, d, r);
    S rr = 0;
    for (int a = 0; a < M; ++a) rr += r[a] * r[a];
    if (A.cost_only) {
      acc[0] += (double)rr;
      continue;
    }
    S J[M * N];  // row-major, as IBaseModel::f_df fills it
    if (A.numeric) {
      for (int j = 0; j < N; ++j) {
        S xp[N];
        for (int k = 0; k < N; ++k) xp[k] = A.x[k];
        xp[j] += A.h[j];
        S rp[M];
        user_residual(xp, d, rp);
        for (int a = 0; a < M; ++a) J[a * N + j] = (rp[a] - r[a]) / A.h[j];
      }
    } else {
      user_jacobian(A.x, d, J);
    }
    S w = 1;
    if (A.loss_kind == 1) {
      const S den = rr + A.loss_param;
      w = (A.loss_param * A.loss_param) / (den * den);
    }
    S SJ[M * N], Sr[M];
    for (int a = 0; a < M; ++a) {
      for (int j = 0; j < N; ++j) {
        S v = 0;
        for (int c = 0; c < M; ++c) v += A.cov[a * M + c] * J[c * N + j];
        SJ[a * N + j] = v;
      }
      S v = 0;
      for (int c = 0; c < M; ++c) v += A.cov[a * M + c] * r[c];
      Sr[a] = v;
    }
    for (int j = 0; j < N; ++j)
      for (int i2 = 0; i2 < N; ++i2) {
        S v = 0;
        for (int a = 0; a < M; ++a) v += (w * J[a * N + i2]) * SJ[a * N + j];
        acc[j * N + i2] += (double)v;
      }
    for (int i2 = 0; i2 < N; ++i2) {
      S v = 0;
      for (int a = 0; a < M; ++a) v += (w * J[a * N + i2]) * Sr[a];
      acc[N * N + i2] += (double)v;
    }
    acc[N * N + N] += (double)rr;
  }
  const int nacc = A.cost_only ? 1 : NACC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < nacc; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) lds[wave][k] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < nacc) {
    double v = 0.0;
    for (int w2 = 0; w2 < kBlock / 64; ++w2) v += lds[w2][threadIdx.x];
    A.partials[(size_t)blockIdx.x * nacc + threadIdx.x] = v;
  }
}
)JIT";

std::string &jitError() {
  static thread_local std::string e;
  return e;
}

}  // namespace

const char *jitLastError() { return jitError().c_str(); }

bool jitCompile(int scalar_bytes, int n_params, int n_outputs, int n_planes,
                const char *residual_body, const char *jacobian_body, JitKernel &out) {
  if (!residual_body || n_params < 1 || n_params > kMaxParams || n_outputs < 1 || n_outputs > 4 ||
      n_planes < 0 || n_planes > 16) {
    jitError() = "bad model shape (1 <= n <= 8, 1 <= m <= 4, planes <= 16) or no residual source";
    return false;
  }
  std::string source(kJitPrologue);
  source += "__device__ inline void user_residual(const S *x, const S *d, S *r) {\n";
  source += residual_body;
  source += "\n}\n__device__ inline void user_jacobian(const S *x, const S *d, S *J) {\n";
  source += jacobian_body ? jacobian_body : "";
  source += "\n}\n";
  source += kJitSweep;
  const std::string defs[] = {
      std::string("-DMOPT_S=") + (scalar_bytes == 8 ? "double" : "float"),
      "-DMOPT_N=" + std::to_string(n_params),
      "-DMOPT_M=" + std::to_string(n_outputs),
      "-DMOPT_D=" + std::to_string(n_planes),
      "--offload-arch=gfx950",
      "-O3",
      "-std=c++17",
  };
  std::vector<const char *> opts;
  for (const auto &d : defs) opts.push_back(d.c_str());

  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, source.c_str(), "mopt_jit_model.hip", 0, nullptr, nullptr) !=
      HIPRTC_SUCCESS) {
    jitError() = "hiprtcCreateProgram failed";
    return false;
  }
  const hiprtcResult rc = hiprtcCompileProgram(prog, int(opts.size()), opts.data());
  if (rc != HIPRTC_SUCCESS) {
    size_t log_size = 0;
    hiprtcGetProgramLogSize(prog, &log_size);
    std::string log(log_size, '\0');
    if (log_size) hiprtcGetProgramLog(prog, &log[0]);
    jitError() = std::string("model source did not compile: ") + hiprtcGetErrorString(rc) + "\n" + log;
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t code_size = 0;
  hiprtcGetCodeSize(prog, &code_size);
  std::vector<char> code(code_size);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  if (hipModuleLoadData(&out.module, code.data()) != hipSuccess) {
    jitError() = "hipModuleLoadData failed for the compiled model";
    return false;
  }
  if (hipModuleGetFunction(&out.sweep, out.module, "mopt_jit_sweep") != hipSuccess) {
    jitError() = "compiled model has no mopt_jit_sweep";
    (void)hipModuleUnload(out.module);
    out.module = nullptr;
    return false;
  }
  out.has_jacobian = jacobian_body && *jacobian_body;
  return true;
}

void jitRelease(JitKernel &k) {
  if (k.module) (void)hipModuleUnload(k.module);
  k.module = nullptr;
  k.sweep = nullptr;
}

hipError_t jitLaunch(const JitKernel &k, const void *args, size_t args_bytes, int grid,
                     hipStream_t stream) {
  size_t size = args_bytes;
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void *>(args),
                    HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(k.sweep, unsigned(grid), 1, 1, kBlockThreads, 1, 1, 0, stream, nullptr,
                               config);
}

}  // namespace mopt
