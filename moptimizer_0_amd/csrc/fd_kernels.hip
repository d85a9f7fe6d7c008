// gfx950 kernels of the forward-difference linearization sweep evaluated as the reference does
// (linearization.h:65-124 with the point-to-point model of tst/point2point.cpp:32-51): seven
// residuals and eighteen quotients per correspondence.  A translation unit of its own: compiled
// without the SLP vectorizer (it pairs the fp32 points of a pack into v_pk_* instructions at the
// price of ~150 register moves and 90 more VGPRs per wave).
#include "sweep_device.hpp"
#include "fd_device.hpp"

#include <cstdlib>

namespace mopt {
namespace {

// ---- point-to-point, forward differences as the reference evaluates them ------------------------
// The literal kernel above would carry 7 transforms x 12 scalars as kernel arguments; in fp64 that
// is 168 SGPRs of constants against ~100 available, so the compiler spilled them and the sweep ran
// at 28-44 % of the HBM roof.  This form of the same arithmetic
//   * keeps the constants in LDS and re-reads what a step needs (same-address reads: one broadcast
//     each; a compiler barrier in front keeps them from being hoisted into ~114 VGPRs);
//   * uses what the reference's perturbed models have in common: x + h e_j for a translation
//     parameter leaves R untouched, so r+_a - r_a is EXACTLY zero for a != j and the j-th entry
//     only needs the shared partial sum (R p)_j — three columns cost 4 operations each instead of
//     a transformed point each: J = [diag(d) | A];
//   * spells the transformed point as fma(T2, p2, fma(T1, p1, T0 p0)) + t, the contraction of the
//     reference's 4x4 * [p;1] product (tst/point2point.cpp:42-45 under -march=native), for every
//     one of the seven residuals alike, so that r+ - r carries the same rounding as on the CPU;
//   * takes the V correspondences of a 16-byte pack one after the other behind scheduling barriers
//     (interleaved by the scheduler they need > 300 registers), with compile-time pack indices.
// Every J entry, the loss weight and every product are still formed per point (linearization.h:
// 101-117).  COV: identity (w J^T J has three structural zeros and nine one-term entries), symmetric
// (rows of kAccSym) or general (rows of kAccFull) covariance — tst/covariance.cpp:45-63,
// tst/powell.cpp:107-136 are forward differences under a covariance.  Only the bits of J matter for
// parity with the CPU path (the quotient amplifies them by eps / h); the products that follow are
// associated for the fewest instructions: S A once, w folded into S A, d and r.
// (the per-point arithmetic — PointPair, the accumulate helpers, p2pForwardDiffRow / p2pForwardDiffBody — is in
// fd_device.hpp: the one-launch solve of small problems in sweep_kernels.hip evaluates with it too)

template <typename S, bool STREAMING, int COV, int HOME>
__global__ __launch_bounds__(kBlockThreads) void p2pForwardDiffKernel(const S *tiles, int num_tiles,
                                                                      const P2PSweepArgs<S> A) {
  p2pForwardDiffBody<S, STREAMING, COV, HOME>(tiles, num_tiles, A, blockIdx.x, gridDim.x, A);
}

// the rotation entries' home as in launchForwardDiff
template <typename S, int COV>
constexpr int kFdHomeFor = sizeof(S) == 8 ? (COV != kCovGeneral ? int(kFdRotationRegisters)
                                                                : int(kFdRotationMixedPlus))
                                          : int(kFdRotationLds);

template <typename S, bool STREAMING, int COV>
__global__ __launch_bounds__(kBlockThreads) void p2pForwardDiffResidentArgsKernel(
    const P2PSweepArgs<S> *__restrict__ d_args, const LmControl *__restrict__ control) {
  if (control->done) return;
  const P2PSweepArgs<S> A = *d_args;
  p2pForwardDiffBody<S, STREAMING, COV, kFdHomeFor<S, COV>>(A.tiles, A.num_tiles, A, blockIdx.x,
                                                            gridDim.x, *d_args);
}

// The sweep of a cost whose forward differences are taken by moments or literally as each evaluated
// point asks (sweep.hpp kLmGateMoments; lm_device.hpp writes the gates): ONE launch per point, the choice
// read from the control block — uniform over the grid.  The literal form runs over the whole grid (two
// workgroups per CU: it is VALU-heavy); the moments form over its first `moments_grid` workgroups (one
// per CU, rows of kAccMoments), the others leave at once.  Either way the arithmetic is the body the
// single-purpose kernels run.
template <typename S, bool STREAMING, int COV>
__global__ __launch_bounds__(kBlockThreads) void p2pForwardDiffEitherResidentKernel(
    const S *tiles, int num_tiles, const P2PSweepArgs<S> *__restrict__ d_args,
    const LmControl *__restrict__ control, int moments_grid) {
  const int gate = control[kLmGateMoments].done;  // (the one word these kernels read of the control block)
  if (gate == kLmGateStopped) return;
  const bool literal = gate == kLmGateLiteralForm;
  if (!literal && int(blockIdx.x) >= moments_grid) return;
  const P2PSweepArgs<S> A = *d_args;
  if (literal) {
    p2pForwardDiffBody<S, STREAMING, COV, kFdHomeFor<S, COV>>(tiles, num_tiles, A, blockIdx.x, gridDim.x,
                                                              *d_args);
  } else {
    double acc[kAccMoments];
#pragma unroll
    for (int k = 0; k < kAccMoments; ++k) acc[k] = 0.0;
    sweepTiles<S, STREAMING>(
        tiles, num_tiles,
        [&](const Pack<S>(&cur)[6], long long first) { momentsOfPack<S>(acc, cur, first, A); },
        int(blockIdx.x), moments_grid);
    blockReduceStore<kAccMoments>(acc, A.partials + size_t(blockIdx.x) * kAccMoments);
  }
}

// several forward-difference costs of one problem in one launch (workgroups [first_block[k],
// first_block[k + 1]) sweep cost k)
template <typename S, bool STREAMING, int COV>
__global__ __launch_bounds__(kBlockThreads) void p2pForwardDiffResidentSetKernel(
    const ResidentSweepSet set, const LmControl *__restrict__ control) {
  if (control->done) return;
  const int k = costOfBlock(set);
  const P2PSweepArgs<S> *d_args = static_cast<const P2PSweepArgs<S> *>(set.args[k]);
  const P2PSweepArgs<S> A = *d_args;
  p2pForwardDiffBody<S, STREAMING, COV, kFdHomeFor<S, COV>>(
      A.tiles, A.num_tiles, A, int(blockIdx.x) - set.first_block[k],
      set.first_block[k + 1] - set.first_block[k], *d_args);
}

}  // namespace

template <typename S>
hipError_t launchForwardDiff(const P2PSweepArgs<S> &args, int cov_mode, int grid,
                             const LaunchSite &site) {
  // fp64: identity / symmetric covariance leave room for the 54 VGPRs (206 / 220 in all: two waves
  // per SIMD either way); the general form (43 accumulators) for 48 of them (256: three entries of
  // the third perturbed rotation re-read from LDS; all from LDS 106 us at 10 M, two rotations in
  // registers 100.5-101.3, this 99.5); fp32 (whose LDS reads are half the size) re-reads everything
  // from LDS.  MOPT_FD_ROTATION_HOME=0|1|2|3 overrides (tuning).
  static const int forced = [] {
    const char *e = getenv("MOPT_FD_ROTATION_HOME");
    return e ? atoi(e) : -1;
  }();
  const int by_form = sizeof(S) == 8 ? (cov_mode != kCovGeneral ? int(kFdRotationRegisters)
                                                                : int(kFdRotationMixedPlus))
                                     : int(kFdRotationLds);
  const int home = forced >= 0 ? forced : by_form;
#define MOPT_LAUNCH_FD_HOME(COV, HOME)                                                        \
  (site.streaming ? launchTiled(p2pForwardDiffKernel<S, true, COV, HOME>, grid, site, args)      \
                  : launchTiled(p2pForwardDiffKernel<S, false, COV, HOME>, grid, site, args))
#define MOPT_LAUNCH_FD(COV)                                                                      \
  (home == kFdRotationRegisters ? MOPT_LAUNCH_FD_HOME(COV, kFdRotationRegisters)                 \
   : home == kFdRotationMixed   ? MOPT_LAUNCH_FD_HOME(COV, kFdRotationMixed)                     \
   : home == kFdRotationMixedPlus ? MOPT_LAUNCH_FD_HOME(COV, kFdRotationMixedPlus)               \
                                : MOPT_LAUNCH_FD_HOME(COV, kFdRotationLds))
  switch (cov_mode) {
    case kCovIdentity:
      return MOPT_LAUNCH_FD(kCovIdentity);
    case kCovSymmetric:
      return MOPT_LAUNCH_FD(kCovSymmetric);
    default:
      return MOPT_LAUNCH_FD(kCovGeneral);
  }
#undef MOPT_LAUNCH_FD
#undef MOPT_LAUNCH_FD_HOME
}
template hipError_t launchForwardDiff<float>(const P2PSweepArgs<float> &, int, int, const LaunchSite &);
template hipError_t launchForwardDiff<double>(const P2PSweepArgs<double> &, int, int,
                                              const LaunchSite &);

template <typename S>
hipError_t launchForwardDiffResident(const P2PSweepArgs<S> *d_args, const LmControl *control,
                                     int cov_mode, int grid, const LaunchSite &site) {
  // tiles / num_tiles are not in reach of this signature: read through the argument block
  const dim3 g(grid), b(kBlockThreads);
#define MOPT_LAUNCH_FD_RESIDENT(COV)                                                              \
  if (site.streaming)                                                                             \
    hipLaunchKernelGGL((p2pForwardDiffResidentArgsKernel<S, true, COV>), g, b, 0, site.stream,    \
                       d_args, control);                                                          \
  else                                                                                            \
    hipLaunchKernelGGL((p2pForwardDiffResidentArgsKernel<S, false, COV>), g, b, 0, site.stream,   \
                       d_args, control)
  switch (cov_mode) {
    case kCovIdentity:
      MOPT_LAUNCH_FD_RESIDENT(kCovIdentity);
      break;
    case kCovSymmetric:
      MOPT_LAUNCH_FD_RESIDENT(kCovSymmetric);
      break;
    default:
      MOPT_LAUNCH_FD_RESIDENT(kCovGeneral);
      break;
  }
#undef MOPT_LAUNCH_FD_RESIDENT
  return hipGetLastError();
}
template <typename S>
hipError_t launchForwardDiffEitherResident(const S *tiles, int num_tiles, const P2PSweepArgs<S> *d_args,
                                           const LmControl *control, int cov_mode, int grid,
                                           int moments_grid, const LaunchSite &site) {
  const dim3 g(grid), b(kBlockThreads);
#define MOPT_LAUNCH_FD_EITHER(COV)                                                                \
  if (site.streaming)                                                                             \
    hipLaunchKernelGGL((p2pForwardDiffEitherResidentKernel<S, true, COV>), g, b, 0, site.stream,  \
                       tiles, num_tiles, d_args, control, moments_grid);                          \
  else                                                                                            \
    hipLaunchKernelGGL((p2pForwardDiffEitherResidentKernel<S, false, COV>), g, b, 0, site.stream, \
                       tiles, num_tiles, d_args, control, moments_grid)
  switch (cov_mode) {
    case kCovIdentity:
      MOPT_LAUNCH_FD_EITHER(kCovIdentity);
      break;
    case kCovSymmetric:
      MOPT_LAUNCH_FD_EITHER(kCovSymmetric);
      break;
    default:
      MOPT_LAUNCH_FD_EITHER(kCovGeneral);
      break;
  }
#undef MOPT_LAUNCH_FD_EITHER
  return hipGetLastError();
}
template hipError_t launchForwardDiffEitherResident<float>(const float *, int, const P2PSweepArgs<float> *,
                                                           const LmControl *, int, int, int,
                                                           const LaunchSite &);
template hipError_t launchForwardDiffEitherResident<double>(const double *, int,
                                                            const P2PSweepArgs<double> *, const LmControl *,
                                                            int, int, int, const LaunchSite &);

template <typename S>
hipError_t launchForwardDiffResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                        int cov_mode, const LaunchSite &site) {
  const dim3 g(set.first_block[set.num_costs]), b(kBlockThreads);
#define MOPT_LAUNCH_FD_SET(COV)                                                                   \
  if (site.streaming)                                                                             \
    hipLaunchKernelGGL((p2pForwardDiffResidentSetKernel<S, true, COV>), g, b, 0, site.stream, set, \
                       control);                                                                  \
  else                                                                                            \
    hipLaunchKernelGGL((p2pForwardDiffResidentSetKernel<S, false, COV>), g, b, 0, site.stream,    \
                       set, control)
  switch (cov_mode) {
    case kCovIdentity:
      MOPT_LAUNCH_FD_SET(kCovIdentity);
      break;
    case kCovSymmetric:
      MOPT_LAUNCH_FD_SET(kCovSymmetric);
      break;
    default:
      MOPT_LAUNCH_FD_SET(kCovGeneral);
      break;
  }
#undef MOPT_LAUNCH_FD_SET
  return hipGetLastError();
}
template hipError_t launchForwardDiffResidentSet<float>(const ResidentSweepSet &, const LmControl *,
                                                        int, const LaunchSite &);
template hipError_t launchForwardDiffResidentSet<double>(const ResidentSweepSet &,
                                                         const LmControl *, int,
                                                         const LaunchSite &);

template hipError_t launchForwardDiffResident<float>(const P2PSweepArgs<float> *, const LmControl *,
                                                     int, int, const LaunchSite &);
template hipError_t launchForwardDiffResident<double>(const P2PSweepArgs<double> *,
                                                      const LmControl *, int, int,
                                                      const LaunchSite &);

}  // namespace mopt
