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
#include "sweep.hpp"

#include <limits>

namespace mopt {
namespace {

template <typename S>
struct LmState {
  S x0[kMaxParams];
  S xi[kMaxParams];
  S delta[kMaxParams];
  S H[kMaxParams * kMaxParams];  // column-major n x n, at x0
  S b[kMaxParams];
  S y0;
  S lambda;
  S nu;
  int k;            // trial points tried in this outer iteration
  int it;           // executed outer iterations
  int awaiting_x0;  // the sweep in flight is the linearization at x0, not a trial
  int status;       // LmStatus
  int trials;
  unsigned long long steps;  // step-kernel runs: the host's progress word
};

template <typename S>
struct LmStart {
  S x[kMaxParams];
};

__device__ __forceinline__ void sinCosOf(double t, double *s, double *c) { sincos(t, s, c); }
__device__ __forceinline__ void sinCosOf(float t, float *s, float *c) { sincosf(t, s, c); }

// x = (t, w) -> row-major 3x4 [Exp(w) | t], the arithmetic of include/moptimizer_amd/so3.hpp
// (so3::convert6DOFParameterToMatrix + so3::Exp, src/so3.cpp:7-19,43-57).
template <typename S>
__device__ void rigidFrom6DOF(const S *x, S (&T)[12]) {
#pragma clang fp contract(off)
  const S *w = x + 3;
  S R[9];
  const S theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (theta > S(10) * std::numeric_limits<S>::epsilon()) {
    const S ax = w[0] / theta, ay = w[1] / theta, az = w[2] / theta;
    S s, c;
    sinCosOf(theta, &s, &c);
    const S c1 = S(1) - c;
    const S xx = ax * ax, yy = ay * ay, zz = az * az;
    R[0] = S(1) + c1 * (-(yy + zz));
    R[1] = s * (-az) + c1 * (ax * ay);
    R[2] = s * (ay) + c1 * (ax * az);
    R[3] = s * (az) + c1 * (ax * ay);
    R[4] = S(1) + c1 * (-(xx + zz));
    R[5] = s * (-ax) + c1 * (ay * az);
    R[6] = s * (-ay) + c1 * (ax * az);
    R[7] = s * (ax) + c1 * (ay * az);
    R[8] = S(1) + c1 * (-(xx + yy));
  } else {
    R[0] = 1; R[1] = 0; R[2] = 0;
    R[3] = 0; R[4] = 1; R[5] = 0;
    R[6] = 0; R[7] = 0; R[8] = 1;
  }
  for (int i = 0; i < 3; ++i) {
    T[i * 4 + 0] = R[i * 3 + 0];
    T[i * 4 + 1] = R[i * 3 + 1];
    T[i * 4 + 2] = R[i * 3 + 2];
    T[i * 4 + 3] = x[i];
  }
}

// Forward-difference step of linearization.h:85-89.
template <typename S>
__device__ __forceinline__ S forwardStep(S xj) {
#pragma clang fp contract(off)
  const S min_step = sqrt(std::numeric_limits<S>::epsilon());
  S h = min_step * fabs(xj);
  if (h == S(0)) h = min_step;
  return h;
}

// delta = (H + lambda diag H)^{-1} (-b) by LDL^T with diagonal pivoting: the factorisation the
// reference asks Eigen for (levenberg_marquadt_dyn.cpp:78-80), operation for operation the host
// statement of tests/support/moptimizer_caller/ldlt.hpp (vanishing pivots give a zero component).
// Runs on one lane; its work arrays live in LDS (indexed dynamically: as private arrays they would
// go to scratch memory, a global-memory round trip per dependent access).
template <typename S>
struct SolveScratch {
  S m[kMaxParams][kMaxParams];
  S scaled[kMaxParams], y[kMaxParams];
  int perm[kMaxParams];
};

template <typename S>
__device__ void solveDamped(const S *H, const S *b, S lambda, int n, S *delta,
                            SolveScratch<S> &w) {
#pragma clang fp contract(off)
  auto &m = w.m;
  auto &perm = w.perm;
  auto &scaled = w.scaled;
  auto &y = w.y;
  for (int c = 0; c < n; ++c)
    for (int r = 0; r < n; ++r) m[r][c] = H[c * n + r];
  for (int i = 0; i < n; ++i) {
    m[i][i] += lambda * H[i * n + i];
    perm[i] = i;
  }
  for (int k = 0; k < n; ++k) {
    int piv = k;
    S best = fabs(m[k][k]);
    for (int i = k + 1; i < n; ++i) {
      const S v = fabs(m[i][i]);
      if (v > best) {
        best = v;
        piv = i;
      }
    }
    if (piv != k) {  // exchange rows / columns k < piv of the symmetric matrix in the lower triangle
      const int a = k, bb = piv;
      for (int j = 0; j < a; ++j) { const S t = m[a][j]; m[a][j] = m[bb][j]; m[bb][j] = t; }
      for (int i = bb + 1; i < n; ++i) { const S t = m[i][a]; m[i][a] = m[i][bb]; m[i][bb] = t; }
      for (int i = a + 1; i < bb; ++i) { const S t = m[i][a]; m[i][a] = m[bb][i]; m[bb][i] = t; }
      { const S t = m[a][a]; m[a][a] = m[bb][bb]; m[bb][bb] = t; }
      { const int t = perm[a]; perm[a] = perm[bb]; perm[bb] = t; }
    }
    S dk = m[k][k];
    for (int j = 0; j < k; ++j) {
      scaled[j] = m[k][j] * m[j][j];
      dk -= m[k][j] * scaled[j];
    }
    m[k][k] = dk;
    for (int i = k + 1; i < n; ++i) {
      S v = m[i][k];
      for (int j = 0; j < k; ++j) v -= m[i][j] * scaled[j];
      m[i][k] = v;
    }
    if (fabs(dk) > S(0))
      for (int i = k + 1; i < n; ++i) m[i][k] /= dk;
  }
  for (int i = 0; i < n; ++i) y[i] = -b[perm[i]];
  for (int i = 0; i < n; ++i) {
    S v = y[i];
    for (int j = 0; j < i; ++j) v -= m[i][j] * y[j];
    y[i] = v;
  }
  const S tiny = std::numeric_limits<S>::min();
  for (int i = 0; i < n; ++i) {
    const S d = m[i][i];
    y[i] = (fabs(d) > tiny) ? y[i] / d : S(0);
  }
  for (int i = n - 1; i >= 0; --i) {
    S v = y[i];
    for (int j = i + 1; j < n; ++j) v -= m[j][i] * y[j];
    y[i] = v;
  }
  for (int i = 0; i < n; ++i) delta[perm[i]] = y[i];
}

// (K T) C, row-major 3x4: the matrix products of tst/camera_calibration.cpp:37 in the association
// of the host statement (c_abi.cpp projectionFor).
__device__ void projectionFor(const LmCostDesc &d, const double (&T)[12], double (&M)[12]) {
#pragma clang fp contract(off)
  double T4[16];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 4; ++k) T4[r * 4 + k] = T[r * 4 + k];
  T4[12] = T4[13] = T4[14] = 0.0;
  T4[15] = 1.0;
  double KT[12];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 4; ++k) {
      double v = 0.0;
      for (int q = 0; q < 4; ++q) v += d.camera[r * 4 + q] * T4[q * 4 + k];
      KT[r * 4 + k] = v;
    }
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 4; ++k) {
      double v = 0.0;
      for (int q = 0; q < 4; ++q) v += KT[r * 4 + q] * d.frame[q * 4 + k];
      M[r * 4 + k] = v;
    }
}

// The per-x constants of every cost's next sweep, at point x (LDS).  Lane j (0..6) forms the
// transform at x (j = 0) or at x + h_j e_j.
template <typename S>
__device__ void writeSweepConstants(const LmProblem &P, const S *x) {
  __shared__ S Tj[1 + kNumParams][12];
  __shared__ S inv_h[kNumParams];
  const int tid = threadIdx.x;
  for (int ci = 0; ci < P.num_costs; ++ci) {
    const LmCostDesc &d = P.cost[ci];
    const bool numeric = d.jac_mode == kJacNumeric;
    if (d.model == kLmPoint2Point || d.model == kLmReprojection) {
      if (tid <= kNumParams) {
        S xp[kNumParams];
        for (int k = 0; k < kNumParams; ++k) xp[k] = x[k];
        if (tid > 0) {
          if (numeric) {
            const S h = forwardStep<S>(x[tid - 1]);
            xp[tid - 1] = x[tid - 1] + h;
            inv_h[tid - 1] = S(1) / h;
          } else {
            inv_h[tid - 1] = S(0);
          }
        }
        rigidFrom6DOF<S>(xp, Tj[tid]);
      }
      __syncthreads();
      if (d.model == kLmPoint2Point) {
        P2PSweepArgs<S> *a = static_cast<P2PSweepArgs<S> *>(d.args);
        if (tid < (1 + kNumParams) * 12) a->T[tid / 12][tid % 12] = Tj[tid / 12][tid % 12];
        if (tid < kNumParams) a->inv_h[tid] = inv_h[tid];
        if (d.moments && numeric && tid < 18) {
          // column j of J is ((R_j - R) p + (t_j - t)) / h_j  (c_abi.cpp fillBasis)
          const int r = tid / 6, j = tid % 6;
          d.basis->J[0][r * 6 + j] = double((Tj[1 + j][r * 4 + 3] - Tj[0][r * 4 + 3]) * inv_h[j]);
          for (int k = 0; k < 3; ++k)
            d.basis->J[1 + k][r * 6 + j] =
                double((Tj[1 + j][r * 4 + k] - Tj[0][r * 4 + k]) * inv_h[j]);
        }
      } else {
        if constexpr (sizeof(S) == 8) {
          ReprojSweepArgs *a = static_cast<ReprojSweepArgs *>(d.args);
          if (tid <= kNumParams) {
            double M[12];
            projectionFor(d, Tj[tid], M);
            for (int k = 0; k < 12; ++k) a->M[tid][k] = M[k];
            if (tid > 0) a->inv_h[tid - 1] = inv_h[tid - 1];
          }
        }
      }
      __syncthreads();
    } else {  // scalar models: the sweep differentiates by itself, it needs x and the steps
      ScalarSweepArgs<S> *a = static_cast<ScalarSweepArgs<S> *>(d.args);
      if (tid < kMaxParams) {
        const S xv = tid < P.n ? x[tid] : S(0);
        a->x[tid] = xv;
        a->h[tid] = forwardStep<S>(xv);
      }
    }
  }
}

template <typename S>
__device__ __forceinline__ bool isCostSmall(S y) {  // optimizer.h:26-29
  return fabs(y) < S(8) * std::numeric_limits<S>::epsilon();
}

__device__ __forceinline__ void storeReport(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p),
                     static_cast<unsigned long long>(__double_as_longlong(v)), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename S>
__global__ __launch_bounds__(128) void lmStepKernel(const LmProblem P, int init,
                                                    const LmStart<S> start) {
  LmControl *ctl = P.control;
  LmState<S> &st = *static_cast<LmState<S> *>(P.state);
  if (!init && ctl->done) return;
  __shared__ S next_x[kMaxParams];
  __shared__ int propose;
  __shared__ SolveScratch<S> solve_scratch;
  __shared__ S Hs[kMaxParams * kMaxParams], bs[kMaxParams];
  const int n = P.n;
  const int nn = n * n;

  if (threadIdx.x == 0) {
#pragma clang fp contract(off)
    propose = 0;
    auto finish = [&](int status) {
      st.status = status;
      ctl->done = 1;
    };
    // (H + lambda D) delta = -b ; xi = x0 + delta  (levenberg_marquadt_dyn.cpp:78-83)
    auto proposeTrial = [&]() {
      solveDamped<S>(st.H, st.b, st.lambda, n, st.delta, solve_scratch);
      for (int i = 0; i < n; ++i) {
        st.xi[i] = st.x0[i] + st.delta[i];
        next_x[i] = st.xi[i];
      }
      propose = 1;
    };
    // top of an outer iteration once H, b, y0 at x0 are known (:62-70)
    auto beginOuter = [&]() {
      if (isCostSmall<S>(st.y0)) return finish(kLmConverged);
      if (st.lambda < S(0)) {
        S max_diag = 0;
        for (int i = 0; i < n; ++i) max_diag = fmax(max_diag, fabs(st.H[i * n + i]));
        st.lambda = S(1e-9) * max_diag;
      }
      st.nu = S(2);
      st.k = 0;
      proposeTrial();
    };

    if (init) {
      for (int i = 0; i < kMaxParams; ++i) {
        st.x0[i] = i < n ? start.x[i] : S(0);
        st.xi[i] = st.x0[i];
        st.delta[i] = S(0);
        next_x[i] = st.x0[i];
      }
      st.y0 = S(0);
      st.lambda = S(-1);  // prepare(): :16-17
      st.nu = S(2);
      st.k = 0;
      st.it = 0;
      st.awaiting_x0 = 1;
      st.status = kLmRunning;
      st.trials = 0;
      st.steps = 0;
      ctl->done = 0;
      ctl->trial = 0;
      ctl->pad[0] = 0;
      propose = 1;
    } else {
      // sums over the costs, accumulated in Scalar in cost order (:48-60, :86)
      S ys = S(0);
      for (int q = 0; q < nn; ++q) Hs[q] = S(0);
      for (int q = 0; q < n; ++q) bs[q] = S(0);
      for (int ci = 0; ci < P.num_costs; ++ci) {
        const double *res = P.cost[ci].result;
        for (int q = 0; q < nn; ++q) Hs[q] += S(res[q]);
        for (int q = 0; q < n; ++q) bs[q] += S(res[nn + q]);
        ys += S(res[nn + n]);
      }
      st.trials += 1;
      ctl->trial = st.trials;
      auto adopt = [&]() {
        for (int q = 0; q < nn; ++q) st.H[q] = Hs[q];
        for (int q = 0; q < n; ++q) st.b[q] = bs[q];
        st.y0 = ys;
      };
      if (st.awaiting_x0) {
        st.awaiting_x0 = 0;
        adopt();
        beginOuter();
      } else if (ys != ys) {
        finish(kLmNumericError);  // :88-91
      } else {
        S predicted = S(0);
        for (int i = 0; i < n; ++i) predicted += st.delta[i] * (st.lambda * st.delta[i] - st.b[i]);
        const S rho = (st.y0 - ys) / predicted;  // :93
        if (rho < S(0)) {
          S max_delta = S(0);
          for (int i = 0; i < n; ++i) max_delta = fmax(max_delta, fabs(st.delta[i]));
          if (max_delta < sqrt(std::numeric_limits<S>::epsilon())) {  // delta.h:10-16
            finish(isCostSmall<S>(ys) ? kLmConverged : kLmSmallDelta);
          } else {
            st.lambda = st.nu * st.lambda;  // :108-109
            st.nu = S(2) * st.nu;
            st.k += 1;
            if (st.k < P.lm_max_iterations) {
              proposeTrial();
            } else {
              // inner loop exhausted: the next outer iteration linearizes at the same x0, which
              // gives the H, b, y0 already held
              st.it += 1;
              if (st.it >= P.max_iterations) finish(kLmMaxIterations);
              else beginOuter();
            }
          }
        } else {
          for (int i = 0; i < n; ++i) st.x0[i] = st.xi[i];  // :112
          const double t = 2.0 * double(rho) - 1.0;
          const double shrink = fmax(1.0 / 3.0, 1.0 - t * t * t);  // :113
          st.lambda = S(double(st.lambda) * shrink);
          st.it += 1;
          adopt();  // the trial sweep WAS the linearization at the new x0
          if (st.it >= P.max_iterations) finish(kLmMaxIterations);
          else beginOuter();
        }
      }
    }
    st.steps += 1;
  }
  __syncthreads();
  if (propose) writeSweepConstants<S>(P, next_x);

  // progress for the host: payload write-through into the buffer of this run's parity (the host
  // may still be reading the previous run's), drained, then the progress word
  if (threadIdx.x == 0 && P.report) {
    double *rep = reinterpret_cast<double *>(P.report + (st.steps & 1));
    for (int i = 0; i < kMaxParams; ++i) storeReport(rep + i, double(st.x0[i]));
    storeReport(rep + 8, double(st.y0));
    storeReport(rep + 9, double(st.lambda));
    storeReport(rep + 10, double(st.status));
    storeReport(rep + 11, double(st.it));
    storeReport(rep + 12, double(st.trials));
    storeReport(rep + 13, double(ctl->pad[0]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&P.report->flag, st.steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
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
  for (int i = 0; i < kMaxParams; ++i) start.x[i] = (init && x0 && i < problem.n) ? x0[i] : S(0);
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

}  // namespace mopt
