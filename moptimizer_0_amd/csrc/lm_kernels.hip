// The Levenberg-Marquardt iteration of /root/reference/src/levenberg_marquadt_dyn.cpp:34-119 as a
// one-workgroup gfx950 kernel that runs between sweeps (mopt_lm_minimize; structures in sweep.hpp).
//
// One launch per evaluated point.  It adds the costs' H | b | sum_sq (already finalized, and summed
// over the ranks of a sharded cost), takes the LM decision the reference takes on the host —
// damping, pivoted LDL^T solve of (H + lambda diag H) delta = -b, gain ratio rho, accept / reject,
// the stopping tests — and leaves the per-x constants of the next sweep in HBM: SE(3) transforms at
// x and at the forward-difference points x + h_j e_j (so3.cpp:7-19,43-57; linearization.h:78-92),
// the affine Jacobian basis of the moments sweep, the projection matrices of the reprojection
// model.  Every point is evaluated with the linearization sweep, which also yields sum r^T r, so an
// accepted trial point needs no second sweep (the speculation of mopt_cost_compute, made structural).
//
// Serial logic runs on lane 0 in the reference's order of operations with multiply-add fusion off;
// the seven transforms of a cost are formed by seven lanes.
#include "jit_model.hpp"
#include "lm_device.hpp"

namespace mopt {
namespace {


template <typename S>
__global__ __launch_bounds__(128) void lmStepKernel(const LmProblem P, int init,
                                                    const LmStart<S> start) {
  if (!init && P.control->done) return;
  lmStepBody<S>(P, init != 0, start, nullptr, -1, false, LmStateWords());
}

template <typename Args>
__global__ void storeArgsKernel(const Args value, Args *dst) {
  // one lane copies; the struct is a few hundred bytes and this runs once per changed cost state
  if (threadIdx.x == 0) *dst = value;
}

}  // namespace

template <typename S>
hipError_t launchLmStep(const LmProblem &problem, bool init, const S *x0, hipStream_t stream) {
  LmStart<S> start;
  for (int i = 0; i < kMaxWideParams; ++i) start.x[i] = (init && x0 && i < problem.n) ? x0[i] : S(0);
  hipLaunchKernelGGL((lmStepKernel<S>), dim3(1), dim3(128), 0, stream, problem, init ? 1 : 0, start);
  return hipGetLastError();
}
template hipError_t launchLmStep<float>(const LmProblem &, bool, const float *, hipStream_t);
template hipError_t launchLmStep<double>(const LmProblem &, bool, const double *, hipStream_t);

template <typename Args>
hipError_t launchStoreArgs(const Args &value, Args *d_dst, hipStream_t stream) {
  hipLaunchKernelGGL((storeArgsKernel<Args>), dim3(1), dim3(64), 0, stream, value, d_dst);
  return hipGetLastError();
}
template hipError_t launchStoreArgs<P2PSweepArgs<float>>(const P2PSweepArgs<float> &,
                                                         P2PSweepArgs<float> *, hipStream_t);
template hipError_t launchStoreArgs<P2PSweepArgs<double>>(const P2PSweepArgs<double> &,
                                                          P2PSweepArgs<double> *, hipStream_t);
template hipError_t launchStoreArgs<ReprojSweepArgs>(const ReprojSweepArgs &, ReprojSweepArgs *,
                                                     hipStream_t);
template hipError_t launchStoreArgs<ScalarSweepArgs<float>>(const ScalarSweepArgs<float> &,
                                                            ScalarSweepArgs<float> *, hipStream_t);
template hipError_t launchStoreArgs<ScalarSweepArgs<double>>(const ScalarSweepArgs<double> &,
                                                             ScalarSweepArgs<double> *,
                                                             hipStream_t);
template hipError_t launchStoreArgs<AffineBasis>(const AffineBasis &, AffineBasis *, hipStream_t);
template hipError_t launchStoreArgs<JitArgs<float>>(const JitArgs<float> &, JitArgs<float> *,
                                                    hipStream_t);
template hipError_t launchStoreArgs<JitArgs<double>>(const JitArgs<double> &, JitArgs<double> *,
                                                     hipStream_t);
template hipError_t launchStoreArgs<JitWideArgs<float>>(const JitWideArgs<float> &,
                                                        JitWideArgs<float> *, hipStream_t);
template hipError_t launchStoreArgs<JitWideArgs<double>>(const JitWideArgs<double> &,
                                                         JitWideArgs<double> *, hipStream_t);

}  // namespace mopt
