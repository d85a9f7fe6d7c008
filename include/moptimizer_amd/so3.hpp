// x = (tx, ty, tz, wx, wy, wz)  ->  rigid transform [ Exp(w) t ; 0 1 ].
//
// Host-side, evaluated once per parameter vector (1x per analytic sweep, 1+n per
// forward-difference sweep) and handed to the kernels by value, so the device never calls
// sin/cos and every rank sees bit-identical R, t.
//
// Follows the behaviour of so3::convert6DOFParameterToMatrix and so3::Exp(Ref, Ref) in
// /root/reference/src/so3.cpp:7-19 and :43-57: theta = |w|; for theta > 10*eps the rotation is
// Rodrigues' I + sin(theta) K + (1 - cos(theta)) K^2 with K = skew(w / theta)
// (skew layout: include/moptimizer/so3.h:4); otherwise the identity.  K^2 is expanded in closed
// form here (K^2 = a a^T - I for a unit axis a).
#pragma once

#include <cmath>
#include <limits>

// The rotation must come out bit-identical wherever this header is compiled (the library's host
// code, a caller's own translation unit, the test oracle's independent matrix-form statement):
// forward differences divide R(x + h e_j) - R(x) by h ~ 1e-8, so a 1-ulp difference made by one
// compiler fusing a multiply-add that another does not would move a whole Jacobian column by
// eps / h (measured: 16 % of the entries differed by an ulp between -O3 -march=x86-64-v3 and a
// baseline build before this).  Every operation below is therefore rounded on its own.
#if defined(__clang__)
#define MOPT_SO3_EXACT
#define MOPT_SO3_EXACT_BODY _Pragma("clang fp contract(off)")
#elif defined(__GNUC__)
#define MOPT_SO3_EXACT __attribute__((optimize("fp-contract=off")))
#define MOPT_SO3_EXACT_BODY
#else
#define MOPT_SO3_EXACT
#define MOPT_SO3_EXACT_BODY
#endif

namespace moptimizer {
namespace so3 {

// sin and cos of one angle through glibc's sincos: GCC turns the reference's
// `std::sin(t) ... std::cos(t)` (src/so3.cpp:51-52) into that call, clang keeps two calls, and the
// two libm entry points differ by an ulp for about one argument in two thousand — which forward
// differences then amplify by 1 / h.  One explicit call makes every build agree.
inline void sinCos(double t, double *s, double *c) { ::sincos(t, s, c); }
inline void sinCos(float t, float *s, float *c) { ::sincosf(t, s, c); }

// Row-major 3x4 [R | t]; what the kernels consume.
template <typename Scalar>
struct Rigid3 {
  Scalar m[12];
  Scalar r(int i, int j) const { return m[i * 4 + j]; }
  Scalar t(int i) const { return m[i * 4 + 3]; }
};

template <typename Scalar>
MOPT_SO3_EXACT inline void expSO3(const Scalar *w, Scalar R[9] /* row-major */) {
  MOPT_SO3_EXACT_BODY
  const Scalar theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (theta > Scalar(10) * std::numeric_limits<Scalar>::epsilon()) {
    const Scalar ax = w[0] / theta, ay = w[1] / theta, az = w[2] / theta;
    Scalar s, c;
    sinCos(theta, &s, &c);
    const Scalar c1 = Scalar(1) - c;
    // K = [[0,-az,ay],[az,0,-ax],[-ay,ax,0]],  K^2 = a a^T - |a|^2 I (|a|^2 kept explicit so
    // that rounding in the normalisation is carried the same way a matrix product would).
    const Scalar xx = ax * ax, yy = ay * ay, zz = az * az;
    R[0] = Scalar(1) + c1 * (-(yy + zz));
    R[1] = s * (-az) + c1 * (ax * ay);
    R[2] = s * (ay) + c1 * (ax * az);
    R[3] = s * (az) + c1 * (ax * ay);
    R[4] = Scalar(1) + c1 * (-(xx + zz));
    R[5] = s * (-ax) + c1 * (ay * az);
    R[6] = s * (-ay) + c1 * (ax * az);
    R[7] = s * (ax) + c1 * (ay * az);
    R[8] = Scalar(1) + c1 * (-(xx + yy));
  } else {
    R[0] = 1; R[1] = 0; R[2] = 0;
    R[3] = 0; R[4] = 1; R[5] = 0;
    R[6] = 0; R[7] = 0; R[8] = 1;
  }
}

template <typename Scalar>
inline Rigid3<Scalar> rigidFrom6DOF(const Scalar *x) {
  Scalar R[9];
  expSO3<Scalar>(x + 3, R);
  Rigid3<Scalar> T;
  for (int i = 0; i < 3; ++i) {
    T.m[i * 4 + 0] = R[i * 3 + 0];
    T.m[i * 4 + 1] = R[i * 3 + 1];
    T.m[i * 4 + 2] = R[i * 3 + 2];
    T.m[i * 4 + 3] = x[i];
  }
  return T;
}

// 4x4 column-major homogeneous matrix, the layout user models written against the reference
// expect from so3::convert6DOFParameterToMatrix.
template <typename Scalar>
inline void convert6DOFParameterToMatrix(const Scalar *x, Scalar T[16]) {
  const Rigid3<Scalar> rt = rigidFrom6DOF<Scalar>(x);
  for (int c = 0; c < 4; ++c) {
    for (int r = 0; r < 3; ++r) T[c * 4 + r] = rt.m[r * 4 + c];
    T[c * 4 + 3] = (c == 3) ? Scalar(1) : Scalar(0);
  }
}

// Inverse map for reporting / tests: rotation matrix (row-major) -> rotation vector.
template <typename Scalar>
inline void logSO3(const Scalar R[9], Scalar w[3]) {
  const Scalar tr = R[0] + R[4] + R[8];
  Scalar c = (tr - Scalar(1)) / Scalar(2);
  c = c > Scalar(1) ? Scalar(1) : (c < Scalar(-1) ? Scalar(-1) : c);
  const Scalar theta = std::acos(c);
  const Scalar vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
  const Scalar k = (theta < Scalar(1e-6)) ? Scalar(0.5) : theta / (Scalar(2) * std::sin(theta));
  w[0] = k * vx;
  w[1] = k * vy;
  w[2] = k * vz;
}

// so3::Log as the reference states it (src/so3.cpp:96-105): theta = 0 when trace > 3 - 1e-6, else
// acos((trace - 1) / 2); w = K / 2 for |theta| < 1e-3, else theta / (2 sin theta) K, with
// K = (R21 - R12, R02 - R20, R10 - R01).
template <typename Scalar>
inline void logSO3Reference(const Scalar R[9] /* row-major */, Scalar w[3]) {
  const Scalar trace = R[0] + R[4] + R[8];
  const Scalar theta =
      (trace > Scalar(3.0) - Scalar(1e-6)) ? Scalar(0) : std::acos(Scalar(0.5) * (trace - Scalar(1)));
  const Scalar K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const Scalar k = (std::fabs(theta) < Scalar(0.001)) ? Scalar(0.5)
                                                      : Scalar(0.5) * theta / std::sin(theta);
  for (int i = 0; i < 3; ++i) w[i] = k * K[i];
}

// x (+) delta on SE(3) — what the reference's "TODO Manifold operation"
// (src/levenberg_marquadt_dyn.cpp:82-83) would do in place of `xi_ = x0_map_ + delta_`, as a LEFT
// perturbation (the one MOPT_JAC_ANALYTIC_LEFT differentiates with respect to):
//   R' = Exp(delta_w) Exp(x_w),   t' = Exp(delta_w) x_t + delta_t,   x' = (t', Log(R')).
template <typename Scalar>
inline void se3Plus(const Scalar *x, const Scalar *delta, Scalar *out) {
  Scalar R[9], D[9], RR[9];
  expSO3<Scalar>(x + 3, R);
  expSO3<Scalar>(delta + 3, D);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      RR[i * 3 + j] = (D[i * 3 + 0] * R[0 * 3 + j] + D[i * 3 + 1] * R[1 * 3 + j]) + D[i * 3 + 2] * R[2 * 3 + j];
  Scalar t[3];
  for (int i = 0; i < 3; ++i)
    t[i] = ((D[i * 3 + 0] * x[0] + D[i * 3 + 1] * x[1]) + D[i * 3 + 2] * x[2]) + delta[i];
  Scalar w[3];
  logSO3Reference<Scalar>(RR, w);
  for (int i = 0; i < 3; ++i) {
    out[i] = t[i];
    out[3 + i] = w[i];
  }
}

// The same update with the rotation composed on the RIGHT, the form the reference's two sketches of
// a rotation update take (`parameter_matrix * Exp(delta)`, tst/manifold.cpp:47; `rot_ * rhs_rot`
// with the linear part added, tst/state_model.cpp:28-34) and MOPT_JAC_ANALYTIC_RIGHT differentiates
// with respect to:
//   R' = Exp(x_w) Exp(delta_w),   t' = x_t + delta_t,   x' = (t', Log(R')).
template <typename Scalar>
inline void se3PlusRight(const Scalar *x, const Scalar *delta, Scalar *out) {
  Scalar R[9], D[9], RR[9];
  expSO3<Scalar>(x + 3, R);
  expSO3<Scalar>(delta + 3, D);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      RR[i * 3 + j] = (R[i * 3 + 0] * D[0 * 3 + j] + R[i * 3 + 1] * D[1 * 3 + j]) + R[i * 3 + 2] * D[2 * 3 + j];
  Scalar w[3];
  logSO3Reference<Scalar>(RR, w);
  for (int i = 0; i < 3; ++i) {
    out[i] = x[i] + delta[i];
    out[3 + i] = w[i];
  }
}

}  // namespace so3
}  // namespace moptimizer
