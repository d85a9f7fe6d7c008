// Device code of the device-resident Levenberg-Marquardt step (see lm_kernels.hip for what it does
// and which reference lines it follows).  Included by lm_kernels.hip (the stand-alone step / init
// kernel) and by sweep_kernels.hip, whose finalize kernels run the step in the same launch when
// they belong to the last cost of a problem.  Everything here is in an anonymous namespace of the
// including translation unit.
#pragma once

#include "sweep.hpp"

#include <limits>

namespace mopt {
namespace {

// NMAX: kMaxParams (8) for the models of the path, kMaxWideParams (16) for wide run-time compiled
// models (tst/state_model.cpp: n = 15) — the step is instantiated for both and a problem uses the
// one its n asks for, so that the 6-parameter problems do not carry 16-wide loops.
template <typename S, int NMAX>
struct LmState {
  S x0[NMAX];
  S xi[NMAX];
  S delta[NMAX];
  S H[NMAX * NMAX];  // column-major n x n, at x0
  S b[NMAX];
  S y0;
  S lambda;
  S nu;
  int k;            // trial points tried in this outer iteration
  int it;           // executed outer iterations
  int awaiting_x0;  // the sweep in flight is the linearization at x0, not a trial
  int status;       // LmStatus
  int trials;
  int literal_points;  // of the evaluated points, how many took the literal forward-difference sweep
                       // (LmProblem::fd_per_iterate; reported in LmReport::pad[0])
  unsigned long long steps;  // step-kernel runs: the host's progress word
};

// two words per thread of a stored LmState, requested ahead of use (lmPrefetchState)
struct LmStateWords {
  unsigned int lo, hi;
};

template <typename S>
struct LmStart {
  S x[kMaxWideParams];
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

// x (+) delta on SE(3): include/moptimizer_amd/so3.hpp se3Plus (Exp / Log of src/so3.cpp:43-57,
// :96-105), the manifold form of `xi_ = x0_map_ + delta_` (levenberg_marquadt_dyn.cpp:82-83).
template <typename S>
__device__ void se3Plus(const S *x, const S *delta, S *out) {
#pragma clang fp contract(off)
  S TR[12], TD[12];
  rigidFrom6DOF<S>(x, TR);
  rigidFrom6DOF<S>(delta, TD);  // rotation part Exp(delta_w); its translation column is delta_t
  S RR[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      RR[i * 3 + j] = (TD[i * 4 + 0] * TR[0 * 4 + j] + TD[i * 4 + 1] * TR[1 * 4 + j]) +
                      TD[i * 4 + 2] * TR[2 * 4 + j];
  for (int i = 0; i < 3; ++i)
    out[i] = ((TD[i * 4 + 0] * x[0] + TD[i * 4 + 1] * x[1]) + TD[i * 4 + 2] * x[2]) + delta[i];
  const S trace = RR[0] + RR[4] + RR[8];
  const S theta = (trace > S(3.0) - S(1e-6)) ? S(0) : acos(S(0.5) * (trace - S(1)));
  const S K[3] = {RR[7] - RR[5], RR[2] - RR[6], RR[3] - RR[1]};
  const S k = (fabs(theta) < S(0.001)) ? S(0.5) : S(0.5) * theta / sin(theta);
  for (int i = 0; i < 3; ++i) out[3 + i] = k * K[i];
}

// the same composed on the right (so3.hpp se3PlusRight): R' = Exp(x_w) Exp(delta_w), t' = x_t + delta_t
template <typename S>
__device__ void se3PlusRight(const S *x, const S *delta, S *out) {
#pragma clang fp contract(off)
  S TR[12], TD[12];
  rigidFrom6DOF<S>(x, TR);
  rigidFrom6DOF<S>(delta, TD);
  S RR[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      RR[i * 3 + j] = (TR[i * 4 + 0] * TD[0 * 4 + j] + TR[i * 4 + 1] * TD[1 * 4 + j]) +
                      TR[i * 4 + 2] * TD[2 * 4 + j];
  for (int i = 0; i < 3; ++i) out[i] = x[i] + delta[i];
  const S trace = RR[0] + RR[4] + RR[8];
  const S theta = (trace > S(3.0) - S(1e-6)) ? S(0) : acos(S(0.5) * (trace - S(1)));
  const S K[3] = {RR[7] - RR[5], RR[2] - RR[6], RR[3] - RR[1]};
  const S k = (fabs(theta) < S(0.001)) ? S(0.5) : S(0.5) * theta / sin(theta);
  for (int i = 0; i < 3; ++i) out[3 + i] = k * K[i];
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
template <typename S, int NMAX>
struct SolveScratch {
  S m[NMAX][NMAX];
  S scaled[NMAX], y[NMAX], delta[NMAX];
  int perm[NMAX];
};

// The solution is left in w.delta (LDS: it is written through the permutation, a dynamic index).
template <typename S, int NMAX>
__device__ void solveDamped(const S *H, const S *b, S lambda, int n, SolveScratch<S, NMAX> &w) {
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
  for (int i = 0; i < n; ++i) w.delta[perm[i]] = y[i];
}

// The same factorisation for a compile-time n, fully unrolled so that the matrix lives in registers
// (every index is a constant after unrolling; the run-time pivot choice becomes a chain of
// predicated swaps).  One lane runs it: ~n^3 / 3 dependent fp64 operations instead of as many LDS
// round trips.  Same operations in the same order as solveDamped.
template <typename S, int N>
__device__ void solveDampedFixed(const S *H, const S *b, S lambda, S *delta) {
#pragma clang fp contract(off)
  S m[N][N];
  S scaled[N], y[N];
  int perm[N];
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int r = 0; r < N; ++r) m[r][c] = H[c * N + r];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    m[i][i] += lambda * H[i * N + i];
    perm[i] = i;
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    int piv = k;
    S best = fabs(m[k][k]);
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const S v = fabs(m[i][i]);
      if (v > best) {
        best = v;
        piv = i;
      }
    }
#pragma unroll
    for (int p = k + 1; p < N; ++p) {
      if (piv == p) {
#pragma unroll
        for (int j = 0; j < k; ++j) { const S t = m[k][j]; m[k][j] = m[p][j]; m[p][j] = t; }
#pragma unroll
        for (int i = p + 1; i < N; ++i) { const S t = m[i][k]; m[i][k] = m[i][p]; m[i][p] = t; }
#pragma unroll
        for (int i = k + 1; i < p; ++i) { const S t = m[i][k]; m[i][k] = m[p][i]; m[p][i] = t; }
        { const S t = m[k][k]; m[k][k] = m[p][p]; m[p][p] = t; }
        { const int t = perm[k]; perm[k] = perm[p]; perm[p] = t; }
      }
    }
    S dk = m[k][k];
#pragma unroll
    for (int j = 0; j < k; ++j) {
      scaled[j] = m[k][j] * m[j][j];
      dk -= m[k][j] * scaled[j];
    }
    m[k][k] = dk;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      S v = m[i][k];
#pragma unroll
      for (int j = 0; j < k; ++j) v -= m[i][j] * scaled[j];
      m[i][k] = v;
    }
    if (fabs(dk) > S(0)) {
#pragma unroll
      for (int i = k + 1; i < N; ++i) m[i][k] /= dk;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    S v = S(0);
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (perm[i] == j) v = -b[j];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    S v = y[i];
#pragma unroll
    for (int j = 0; j < i; ++j) v -= m[i][j] * y[j];
    y[i] = v;
  }
  const S tiny = std::numeric_limits<S>::min();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const S d = m[i][i];
    y[i] = (fabs(d) > tiny) ? y[i] / d : S(0);
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    S v = y[i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) v -= m[j][i] * y[j];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (perm[i] == j) delta[j] = y[i];
}

// 1 / d to about an ulp: the hardware estimate and two Newton steps (5 instructions; an IEEE fp64
// division is ~25, and the solve has n of them on its critical path).
__device__ __forceinline__ double fastReciprocal(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
  return r;
}
__device__ __forceinline__ float fastReciprocal(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
  return r;
}
__device__ __forceinline__ double fusedMulAdd(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fusedMulAdd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// The same system when H + lambda diag H is safely positive definite — which it is at every step of
// a well-posed problem: LDL^T in the given order (no pivot search, no predicated row / column
// exchanges: those are most of the ~800 dependent instructions solveDampedFixed<6> issues on its one
// lane), multiply-adds fused, reciprocals instead of divisions: ~130 instructions.  A factorisation
// without pivoting is backward stable for a positive definite matrix, so delta agrees with the
// pivoted solve to eps * cond(H) — the same pose, not the same bits (no restatement reproduces the
// reference's Eigen::LDLT bit for bit either).  Returns false, leaving delta untouched, when a pivot
// is not clearly positive (a parameter the data do not constrain — the as-written test Jacobian has
// an all-zero row and column — or an indefinite matrix): the caller then takes the pivoted path,
// whose handling of vanishing pivots is the host statement's.
template <typename S, int N>
__device__ bool solveDampedPositive(const S *H, const S *b, S lambda, S *delta) {
  S L[N][N];  // strictly lower part
  S d[N], inv_d[N], y[N];
  bool positive = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const S akk = fusedMulAdd(lambda, H[k * N + k], H[k * N + k]);
    S u[N];  // u[j] = L(k, j) d_j
    S dk = akk;
#pragma unroll
    for (int j = 0; j < k; ++j) {
      u[j] = L[k][j] * d[j];
      dk = fusedMulAdd(-L[k][j], u[j], dk);
    }
    // a pivot that has lost ten digits against its diagonal entry is where pivoting starts to matter
    // — in fp32, with seven digits in all, one that is within 64 roundings of having cancelled away
    // (1e-10 sits a thousand times below the fp32 epsilon: nothing would ever fall back)
    constexpr S kPivotFloor = sizeof(S) == 8 ? S(1e-10) : S(64) * std::numeric_limits<S>::epsilon();
    positive = positive && (dk > kPivotFloor * akk);
    d[k] = dk;
    inv_d[k] = fastReciprocal(dk);
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      S v = H[k * N + i];  // lower triangle, as the pivoted statement reads it
#pragma unroll
      for (int j = 0; j < k; ++j) v = fusedMulAdd(-L[i][j], u[j], v);
      L[i][k] = v * inv_d[k];
    }
  }
  if (!positive) return false;  // (also NaN anywhere in H)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    S v = -b[i];
#pragma unroll
    for (int j = 0; j < i; ++j) v = fusedMulAdd(-L[i][j], y[j], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] *= inv_d[i];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    S v = y[i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) v = fusedMulAdd(-L[j][i], y[j], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) delta[i] = y[i];
  return true;
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
// args_local / basis_local: where a kernel that sweeps in the same launch (p2pSolveSmallKernel: the
// whole minimisation in one workgroup) keeps its one point2point cost's constants — LDS — instead of
// the cost's blocks in HBM.
template <typename S>
__device__ __forceinline__ void writeSweepConstants(const LmProblem &P, const S *x,
                                                    void *args_local = nullptr,
                                                    AffineBasis *basis_local = nullptr) {
  // The transforms at x and at x + h_j e_j are the same for every SE(3) cost of the problem: formed
  // once (seven lanes, one sincos each), before the loop over the costs — a cost that does not
  // differentiate numerically takes the transform at x in all seven places and steps of zero, as when
  // each cost formed its own.
  __shared__ S Tstep[1 + kNumParams][12];
  __shared__ S inv_step[kNumParams];
  const int tid = threadIdx.x;
  bool any_pose = false;
  for (int ci = 0; ci < P.num_costs; ++ci)
    any_pose = any_pose || P.cost[ci].model == kLmPoint2Point || P.cost[ci].model == kLmReprojection;
  if (any_pose) {
    if (tid <= kNumParams) {
      S xp[kNumParams];
      for (int k = 0; k < kNumParams; ++k) xp[k] = x[k];
      if (tid > 0) {
        const S h = forwardStep<S>(x[tid - 1]);
        xp[tid - 1] = x[tid - 1] + h;
        inv_step[tid - 1] = S(1) / h;
      }
      rigidFrom6DOF<S>(xp, Tstep[tid]);
    }
    __syncthreads();
  }
  for (int ci = 0; ci < P.num_costs; ++ci) {
    const LmCostDesc &d = P.cost[ci];
    const bool numeric = d.jac_mode == kJacNumeric;
    if (d.model == kLmPoint2Point || d.model == kLmReprojection) {
      // this cost's view: Tj[j] = the transform at x + h_j e_j (numeric) or at x; inv_h likewise
      auto Tj = [&](int j) -> const S(&)[12] { return Tstep[numeric ? j : 0]; };
      auto inv_h_of = [&](int j) { return numeric ? inv_step[j] : S(0); };
      if (d.model == kLmPoint2Point) {
        P2PSweepArgs<S> *a = static_cast<P2PSweepArgs<S> *>(args_local ? args_local : d.args);
        AffineBasis *const basis = basis_local ? basis_local : d.basis;
        if (tid < (1 + kNumParams) * 12) a->T[tid / 12][tid % 12] = Tj(tid / 12)[tid % 12];
        if (tid < kNumParams) a->inv_h[tid] = inv_h_of(tid);
        if (d.moments && d.jac_mode == kJacAnalyticLeft && tid < 18) {
          // J(p) = [I | -skew(R p + t)] = J0 + sum_k p_k J_k (c_abi.cpp fillBasis)
          const int r = tid / 6, j = tid % 6;
          auto pattern = [&](const S (&w)[3]) -> S {  // entry (r, j) of [I | -skew(w)]
            if (j < 3) return r == j ? S(1) : S(0);
            const int c = j - 3;
            if (r == c) return S(0);
            // -skew(w) = [[0, w2, -w1], [-w2, 0, w0], [w1, -w0, 0]]
            const int other = 3 - r - c;
            const bool positive = (r == 0 && c == 1) || (r == 1 && c == 2) || (r == 2 && c == 0);
            return positive ? w[other] : -w[other];
          };
          const S w0[3] = {Tj(0)[3], Tj(0)[7], Tj(0)[11]};
          const S base = pattern(w0);
          basis->J[0][r * 6 + j] = double(base);
          for (int k = 0; k < 3; ++k) {
            const S wk[3] = {w0[0] + Tj(0)[0 * 4 + k], w0[1] + Tj(0)[1 * 4 + k],
                             w0[2] + Tj(0)[2 * 4 + k]};
            basis->J[1 + k][r * 6 + j] = double(pattern(wk)) - double(base);
          }
        }
        if (d.moments && d.jac_mode == kJacAnalyticRight && tid < 18) {
          // J(p) = [I | -R skew(p)]: J0 = [I | 0], J_k = [0 | -R skew(e_k)] (c_abi.cpp fillBasis).
          // skew(e_k)(m, c) = -eps(m, c, k): column c != k has its one entry in row m = 3 - c - k.
          const int r = tid / 6, j = tid % 6;
          basis->J[0][r * 6 + j] = (j < 3 && r == j) ? 1.0 : 0.0;
          for (int k = 0; k < 3; ++k) {
            double v = 0.0;
            if (j >= 3) {
              const int c = j - 3;
              if (c != k) {
                const int m = 3 - c - k;
                const bool even = (m == 0 && c == 1 && k == 2) || (m == 1 && c == 2 && k == 0) ||
                                  (m == 2 && c == 0 && k == 1);
                const double skew_mc = even ? -1.0 : 1.0;
                v = -double(Tj(0)[r * 4 + m]) * skew_mc;
              }
            }
            basis->J[1 + k][r * 6 + j] = v;
          }
        }
        if (d.moments && numeric && tid < 18) {
          // column j of J is ((R_j - R) p + (t_j - t)) / h_j  (c_abi.cpp fillBasis)
          const int r = tid / 6, j = tid % 6;
          basis->J[0][r * 6 + j] = double((Tj(1 + j)[r * 4 + 3] - Tj(0)[r * 4 + 3]) * inv_h_of(j));
          for (int k = 0; k < 3; ++k)
            basis->J[1 + k][r * 6 + j] =
                double((Tj(1 + j)[r * 4 + k] - Tj(0)[r * 4 + k]) * inv_h_of(j));
        }
      } else {
        if constexpr (sizeof(S) == 8) {
          ReprojSweepArgs *a = static_cast<ReprojSweepArgs *>(d.args);
          if (tid <= kNumParams) {
            double M[12];
            projectionFor(d, Tj(tid), M);
            for (int k = 0; k < 12; ++k) a->M[tid][k] = M[k];
            if (tid > 0) a->inv_h[tid - 1] = inv_h_of(tid - 1);
          }
        }
      }
      __syncthreads();
    } else {  // scalar / run-time compiled models: the sweep differentiates (and runs the user's
              // setup) by itself, it needs x and the steps; d.x_offset locates them in the block
      S *xs = reinterpret_cast<S *>(static_cast<char *>(d.args) + d.x_offset);
      const int slots = d.x_slots;  // x[slots] | h[slots]: 8, or 16 in a wide model's block
      if (tid < slots) {
        const S xv = tid < P.n ? x[tid] : S(0);
        xs[tid] = xv;
        xs[slots + tid] = forwardStep<S>(xv);  // h[] follows x[] in every argument layout
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


// One run of the LM step by the calling workgroup (any size >= 128 threads; all threads must call).
//   init        start a minimisation from start.x
//   own_result  H | b | sum_sq of cost `own_index` in LDS when the caller has just finalized it in
//               this same launch (its copy in HBM may not be visible to this workgroup yet);
//               nullptr when every cost's result comes from HBM
//   state_words words `threadIdx.x` (+ blockDim.x) of the stored state, loaded by the caller ahead
//               of its own work so that the round trip overlaps it (ignored when `prefetched` is
//               false)
// (forced inline: as a real call the by-value kernel arguments it takes by reference — 1.2 KB of
// LmProblem — are first copied to scratch by every lane of the 1024-thread workgroup: 60 us)
// Returns whether the loop has stopped (the same for every thread).
// STATE_STAYS: the caller runs every step of the minimisation in this one launch — the state then
// lives in this function's LDS from one run to the next and never travels to HBM and back.
// FIXED_N: the parameter count when the caller knows it at compile time (0: P.n) — the solve's
// dispatch on n and every `i < n` predicate then fold away.
// CHOOSE_FD (with STATE_STAYS: the one-launch solve that holds both forward-difference forms): the step
// also decides which form the next point takes and says so in *choice_out (LDS).
template <typename S, int NMAX, bool STATE_STAYS = false, int FIXED_N = 0, bool CHOOSE_FD = false>
__device__ __forceinline__ bool lmStepBodyFor(const LmProblem &P, bool init, const LmStart<S> &start,
                              const double *own_result, int own_index, bool prefetched,
                              LmStateWords state_words, void *args_local = nullptr,
                              AffineBasis *basis_local = nullptr, int *choice_out = nullptr) {
#ifdef MOPT_LM_TIMING
  __shared__ unsigned long long tick[8];
#define MOPT_TICK(i) if (threadIdx.x == 0) tick[i] = wall_clock64()
#else
#define MOPT_TICK(i)
#endif
  MOPT_TICK(0);
  LmControl *ctl = P.control;
  // The loop's state lives in HBM between launches; this run works on a copy in LDS (loaded and
  // written back by all lanes: one memory round trip each way instead of one per access).
  __shared__ LmState<S, NMAX> st;
  __shared__ S sums[NMAX * NMAX + NMAX + 1];  // H | b | sum_sq over the costs
  __shared__ S next_x[NMAX];
  __shared__ int propose, finished, literal_next;
  __shared__ SolveScratch<S, NMAX> solve_scratch;
  LmState<S, NMAX> *stored = static_cast<LmState<S, NMAX> *>(P.state);
  const int n = FIXED_N > 0 ? FIXED_N : P.n;
  const int nn = n * n;
  const int tid = threadIdx.x;
  constexpr int kStateWords = int(sizeof(LmState<S, NMAX>) / sizeof(unsigned int));
  if (!init) {
    if constexpr (STATE_STAYS) {
    } else if (prefetched) {
      if (tid < kStateWords) reinterpret_cast<unsigned int *>(&st)[tid] = state_words.lo;
      if (tid + int(blockDim.x) < kStateWords)
        reinterpret_cast<unsigned int *>(&st)[tid + blockDim.x] = state_words.hi;
    } else {
      for (int i = tid; i < kStateWords; i += blockDim.x)
        reinterpret_cast<unsigned int *>(&st)[i] = reinterpret_cast<const unsigned int *>(stored)[i];
    }
    // sums over the costs, accumulated in Scalar in cost order (:48-60, :86)
    if (tid < nn + n + 1) {
      S v = S(0);
      if (P.merged && own_result) {
        v = S(own_result[tid]);  // already the sum over the costs (one reduction over all rows)
      } else {
        for (int ci = 0; ci < P.num_costs; ++ci)
          v += S((own_result && ci == own_index) ? own_result[tid] : P.cost[ci].result[tid]);
      }
      sums[tid] = v;
    }
  }
  __syncthreads();
  MOPT_TICK(1);

  __shared__ int adopt_sums, rematch_next;
  if (tid == 0) {
#pragma clang fp contract(off)
    // The decision runs on one lane, so what it costs is a chain of dependent operations: the
    // small vectors are pulled into registers with constant indices (independent LDS reads, one
    // wait), loops run to the compile-time bound under `i < n`, and the 43-value adoption of the
    // sums as the new H | b is left to all lanes afterwards — the solve reads whichever copy holds
    // the current H, b.
    propose = 0;
    finished = 0;
    adopt_sums = 0;
    rematch_next = init ? P.rematch : 0;
    // (16-wide problems keep these in LDS: 4 x 16 fp64 values in registers next to the solve would
    // not fit the 128 VGPRs a 1024-thread workgroup leaves a lane, and scratch costs every launch)
    constexpr bool kInRegisters = NMAX <= kMaxParams;
    S x0_r[kInRegisters ? NMAX : 1], xi_r[kInRegisters ? NMAX : 1], delta_r[kInRegisters ? NMAX : 1],
        bcur_r[kInRegisters ? NMAX : 1];
    __shared__ S vectors_l[kInRegisters ? 1 : 4 * NMAX];
    S *const x0 = kInRegisters ? x0_r : vectors_l;
    S *const xi = kInRegisters ? xi_r : vectors_l + NMAX;
    S *const delta = kInRegisters ? delta_r : vectors_l + 2 * NMAX;
    S *const bcur = kInRegisters ? bcur_r : vectors_l + 3 * NMAX;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      x0[i] = init ? (i < n ? start.x[i] : S(0)) : st.x0[i];
      xi[i] = init ? x0[i] : st.xi[i];
      delta[i] = init ? S(0) : st.delta[i];
      bcur[i] = init ? S(0) : st.b[i];
    }
    S y0 = init ? S(0) : st.y0;
    S lambda = init ? S(-1) : st.lambda;  // prepare(): :16-17
    S nu = init ? S(2) : st.nu;
    int k = init ? 0 : st.k;
    int it = init ? 0 : st.it;
    int status = kLmRunning;
    const S *Hcur = st.H;  // where the current H | b live: the stored state, or the fresh sums
    const S *bsrc = st.b;

    auto finish = [&](int code) {
      status = code;
      finished = 1;
    };
    // (H + lambda D) delta = -b ; xi = x0 + delta  (levenberg_marquadt_dyn.cpp:78-83)
    auto proposeTrial = [&]() {
      MOPT_TICK(5);
      // The unpivoted solve in registers where every pivot is clearly positive; where one is not
      // (a singular or indefinite H: the as-written test Jacobian's zero row, a cost with no valid
      // residual) the pivoted statement of the host runs from LDS — solveDamped, run-time n, the same
      // operations in the same order as the register form of rounds 2-4 (solveDampedFixed) but cold:
      // that form's predicated exchanges kept ~80 lane masks alive across the hot path (118 spilled
      // SGPRs in lmStepKernel<double>, 52 in the fused finalize-and-step kernels).
      bool solved = false;
      if constexpr (kInRegisters) {  // (the 16-wide instantiation never sees these n)
        switch (n) {
          case 6:
            solved = solveDampedPositive<S, 6>(Hcur, bsrc, lambda, delta);
            break;
          case 4:
            solved = solveDampedPositive<S, 4>(Hcur, bsrc, lambda, delta);
            break;
          case 2:
            solved = solveDampedPositive<S, 2>(Hcur, bsrc, lambda, delta);
            break;
          default:
            break;
        }
      }
      if (!solved) {
        solveDamped<S, NMAX>(Hcur, bsrc, lambda, n, solve_scratch);
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
          if (i < n) delta[i] = solve_scratch.delta[i];
      }
      MOPT_TICK(6);
      if (kInRegisters && P.manifold && n == kNumParams) {
        S plus[kNumParams];
        if (P.manifold == 2)
          se3PlusRight<S>(x0, delta, plus);
        else
          se3Plus<S>(x0, delta, plus);
#pragma unroll
        for (int i = 0; i < kNumParams; ++i) xi[i] = plus[i];
      } else {
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
          if (i < n) xi[i] = x0[i] + delta[i];
      }
#pragma unroll
      for (int i = 0; i < NMAX; ++i) next_x[i] = xi[i];
      propose = 1;
    };
    // top of an outer iteration once H, b, y0 at x0 are known (:62-70)
    auto beginOuter = [&]() {
      if (isCostSmall<S>(y0)) return finish(kLmConverged);
      if (lambda < S(0)) {
        S max_diag = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
          if (i < n) max_diag = fmax(max_diag, fabs(Hcur[i * n + i]));
        lambda = S(1e-9) * max_diag;
      }
      nu = S(2);
      k = 0;
      proposeTrial();
    };

    if (init) {
      st.awaiting_x0 = 1;
      st.trials = 0;
      st.steps = 0;
#pragma unroll
      for (int i = 0; i < NMAX; ++i) next_x[i] = x0[i];
      propose = 1;
    } else {
      const S ys = sums[nn + n];
      st.trials += 1;
      auto adopt = [&]() {  // the sums become H | b | y0 (copied into the state by all lanes below)
        adopt_sums = 1;
        Hcur = sums;
        bsrc = sums + nn;
        y0 = ys;
      };
      if (st.awaiting_x0) {
        st.awaiting_x0 = 0;
        adopt();
        beginOuter();
      } else if (ys != ys) {
        finish(kLmNumericError);  // :88-91
      } else {
        S predicted = S(0);
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
          if (i < n) predicted += delta[i] * (lambda * delta[i] - bcur[i]);
        const S rho = (y0 - ys) / predicted;  // :93
        if (rho < S(0)) {
          S max_delta = S(0);
#pragma unroll
          for (int i = 0; i < NMAX; ++i)
            if (i < n) max_delta = fmax(max_delta, fabs(delta[i]));
          if (max_delta < sqrt(std::numeric_limits<S>::epsilon())) {  // delta.h:10-16
            finish(isCostSmall<S>(ys) ? kLmConverged : kLmSmallDelta);
          } else {
            lambda = nu * lambda;  // :108-109
            nu = S(2) * nu;
            k += 1;
            if (k < P.lm_max_iterations) {
              proposeTrial();
            } else {
              // inner loop exhausted: the next outer iteration linearizes at the same x0, which
              // gives the H, b, y0 already held
              it += 1;
              if (it >= P.max_iterations) finish(kLmMaxIterations);
              else beginOuter();
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < NMAX; ++i)
            if (i < n) x0[i] = xi[i];  // :112
          const double t = 2.0 * double(rho) - 1.0;
          const double shrink = fmax(1.0 / 3.0, 1.0 - t * t * t);  // :113
          lambda = S(double(lambda) * shrink);
          it += 1;
          if (it >= P.max_iterations) {
            adopt();
            finish(kLmMaxIterations);
          } else if (P.rematch) {
            // a cost re-searches its correspondences at the top of the next outer iteration
            // (cost->update(x0), :54): the sums of this sweep belong to the old ones — search, then
            // linearize again at the new x0
            y0 = ys;
            st.awaiting_x0 = 1;
            rematch_next = 1;
#pragma unroll
            for (int i = 0; i < NMAX; ++i) next_x[i] = x0[i];
            propose = 1;
          } else {
            adopt();  // the trial sweep WAS the linearization at the new x0
            beginOuter();
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      st.x0[i] = x0[i];
      st.xi[i] = xi[i];
      st.delta[i] = delta[i];
    }
    st.y0 = y0;
    st.lambda = lambda;
    st.nu = nu;
    st.k = k;
    st.it = it;
    st.status = status;
    st.steps += 1;
    // Which forward-difference sweep the next point takes (costs of LmProblem::fd_per_iterate): the rule
    // of the blocking call (c_abi.cpp hasSmallForwardStep) at the very x that sweep is evaluated at.
    // (Compiled out of the one-launch solve that holds the moments only: lm.cpp keeps such problems out of
    // it, and the code cost that kernel a hundred spilled scalar registers.)
    literal_next = 0;
    if constexpr (!STATE_STAYS || CHOOSE_FD) {
      if (P.fd_per_iterate) {
        if (init) st.literal_points = 0;
        if (propose)
          for (int i = 0; i < kNumParams; ++i) {
            const double a = fabs(double(next_x[i]));
            if (a > 0.0 && a < 0.08) literal_next = 1;
          }
        st.literal_points += literal_next;
      }
      if constexpr (CHOOSE_FD) *choice_out = literal_next;
    }
  }
  __syncthreads();
  if (init) {
    for (int q = tid; q < NMAX * NMAX; q += blockDim.x) st.H[q] = S(0);
    if (tid < NMAX) st.b[tid] = S(0);
    __syncthreads();
  } else if (adopt_sums) {
    if (tid < nn) st.H[tid] = sums[tid];
    else if (tid < nn + n) st.b[tid - nn] = sums[tid];
    __syncthreads();
  }
  MOPT_TICK(2);
  if constexpr (!STATE_STAYS)
    for (int i = tid; i < kStateWords; i += blockDim.x)
      reinterpret_cast<unsigned int *>(stored)[i] = reinterpret_cast<const unsigned int *>(&st)[i];
  if (tid == 0) {
    if (init) ctl->pad[0] = 0;
    ctl->trial = st.trials;
    ctl->pad[1] = rematch_next;
    ctl->done = finished;
    if constexpr (!STATE_STAYS) {
      if (P.fd_per_iterate)  // which forward-difference form the next point's kernels run (sweep.hpp)
        ctl[kLmGateMoments].done = finished ? kLmGateStopped : literal_next;
    }
  }
  MOPT_TICK(3);
  if (propose) writeSweepConstants<S>(P, next_x, args_local, basis_local);
  MOPT_TICK(4);

  // For the host: the progress word after every run (one write-through store: the host only needs
  // it to keep its queue a few points ahead), the whole report once, when the loop has stopped —
  // payload write-through, drained, then the word with its low bit set.
  if (tid == 0 && P.report) {
    if (finished) {
      LmReport *rep = P.report;
      for (int i = 0; i < NMAX; ++i) storeReport(&rep->x[i], double(st.x0[i]));
      storeReport(&rep->cost, double(st.y0));
      storeReport(&rep->lambda, double(st.lambda));
      storeReport(&rep->status, double(st.status));
      storeReport(&rep->iterations, double(st.it));
      storeReport(&rep->trials, double(st.trials));
      storeReport(&rep->peer_status, double(ctl->pad[0]));
      if constexpr (!STATE_STAYS || CHOOSE_FD)
        storeReport(&rep->pad[0], P.fd_per_iterate ? double(st.literal_points) : 0.0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef MOPT_LM_TIMING
    // 10 ns ticks since entry: state+sums loaded | decision taken | state stored | constants written
    if (!init && !finished)
      printf("lm step %llu: load %llu decide %llu (before solve %llu, solve %llu) store %llu constants %llu (x10 ns)\n",
             st.steps, tick[1] - tick[0], tick[2] - tick[1], tick[5] - tick[1], tick[6] - tick[5],
             tick[3] - tick[2], tick[4] - tick[3]);
#endif
    __hip_atomic_store(&P.report->flag, st.steps * 2 + (finished ? 1ull : 0ull), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
#undef MOPT_TICK
  return finished != 0;  // (written before the barrier that precedes the state's write-back)
}

// Words `threadIdx.x` and `threadIdx.x + blockDim.x` of the stored state (LmStateWords), for
// lmStepBody's `state_words` (the largest state, 16 fp64 parameters, is 654 words: a workgroup of
// >= 327 threads).
template <typename S>
__device__ __forceinline__ LmStateWords lmPrefetchState(const LmProblem &P) {
  const int words = P.n <= kMaxParams ? int(sizeof(LmState<S, kMaxParams>) / sizeof(unsigned int))
                                      : int(sizeof(LmState<S, kMaxWideParams>) / sizeof(unsigned int));
  // unconditional loads from clamped indices: under a branch they would be waited for on the spot
  const unsigned int *state = reinterpret_cast<const unsigned int *>(P.state);
  const int lo = int(threadIdx.x), hi = int(threadIdx.x + blockDim.x);
  LmStateWords w;
  w.lo = state[lo < words ? lo : 0];
  w.hi = state[hi < words ? hi : 0];
  return w;
}

// The step for this problem's parameter count (see LmState).
template <typename S, int FIXED_N = 0>
__device__ __forceinline__ bool lmStepBody(const LmProblem &P, bool init, const LmStart<S> &start,
                           const double *own_result, int own_index, bool prefetched,
                           LmStateWords state_words) {
  if constexpr (FIXED_N > 0 && FIXED_N <= kMaxParams)  // (the caller's costs fix the parameter count)
    return lmStepBodyFor<S, kMaxParams, false, FIXED_N>(P, init, start, own_result, own_index, prefetched,
                                                        state_words);
  if (P.n <= kMaxParams)
    return lmStepBodyFor<S, kMaxParams>(P, init, start, own_result, own_index, prefetched, state_words);
  return lmStepBodyFor<S, kMaxWideParams>(P, init, start, own_result, own_index, prefetched, state_words);
}

}  // namespace
}  // namespace mopt
