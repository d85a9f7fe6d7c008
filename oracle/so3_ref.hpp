// ORACLE — TEST INFRASTRUCTURE ONLY (see cost_computation.hpp).
//
// Matrix-form restatement of the two so3 functions on the path, written independently of the
// product's closed-form include/moptimizer_amd/so3.hpp so that each checks the other:
//   so3::convert6DOFParameterToMatrix   /root/reference/src/so3.cpp:7-19
//   so3::Exp(Ref, Ref)                  /root/reference/src/so3.cpp:43-57
//   so3::Log                            /root/reference/src/so3.cpp:94-105
//   SKEW_SYMMETRIC_FROM                 /root/reference/include/moptimizer/so3.h:4
#pragma once

#include <cmath>
#include <limits>

// Rounded operation by operation whatever the optimisation flags (the oracle library is built
// -march=x86-64-v3, where GCC would otherwise fuse multiply-adds): forward differences amplify a
// 1-ulp difference in R by 1 / h ~ 1e8, and tests/test_so3_bitwise.py requires this statement and
// the product's closed form to agree bit for bit.
#if defined(__clang__)
#define ORACLE_SO3_EXACT
#define ORACLE_SO3_EXACT_BODY _Pragma("clang fp contract(off)")
#elif defined(__GNUC__)
#define ORACLE_SO3_EXACT __attribute__((optimize("fp-contract=off")))
#define ORACLE_SO3_EXACT_BODY
#else
#define ORACLE_SO3_EXACT
#define ORACLE_SO3_EXACT_BODY
#endif

namespace oracle {

// linearization.h:85-89 — the forward-difference step and the perturbed parameter, each operation
// rounded on its own (x + h must not become fma(min_step, |x|, x)).
template <typename Scalar>
ORACLE_SO3_EXACT inline void forwardStep(Scalar xj, Scalar min_step, Scalar *h, Scalar *x_plus) {
  ORACLE_SO3_EXACT_BODY
  Scalar step = min_step * std::fabs(xj);  // :85
  if (step == Scalar(0)) step = min_step;  // :87
  *h = step;
  *x_plus = xj + step;                     // :89
}

namespace so3 {

// 3x3, indexable as M[r][c]
template <typename Scalar>
struct Mat3 {
  Scalar v[3][3];
};

template <typename Scalar>
inline Mat3<Scalar> skew(const Scalar a[3]) {
  // 0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0   (row by row)
  Mat3<Scalar> K;
  K.v[0][0] = 0.0;   K.v[0][1] = -a[2]; K.v[0][2] = a[1];
  K.v[1][0] = a[2];  K.v[1][1] = 0.0;   K.v[1][2] = -a[0];
  K.v[2][0] = -a[1]; K.v[2][1] = a[0];  K.v[2][2] = 0.0;
  return K;
}

template <typename Scalar>
ORACLE_SO3_EXACT inline Mat3<Scalar> Exp(const Scalar delta[3]) {
  ORACLE_SO3_EXACT_BODY
  Mat3<Scalar> R;
  const Scalar delta_norm =
      std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (delta_norm > Scalar(10.0) * std::numeric_limits<Scalar>::epsilon()) {  // :47
    const Scalar axis[3] = {delta[0] / delta_norm, delta[1] / delta_norm, delta[2] / delta_norm};
    const Mat3<Scalar> K = skew<Scalar>(axis);
    // std::sin(delta_norm), std::cos(delta_norm) (:51-52): GCC, the reference's compiler, merges
    // the pair into one sincos call; written out so that the result does not depend on whether
    // this translation unit's compiler does (the two libm entry points differ by an ulp for
    // about one argument in two thousand).
    Scalar s, c;
    if constexpr (sizeof(Scalar) == 8) ::sincos(delta_norm, &s, &c); else ::sincosf(delta_norm, &s, &c);
    const Scalar c1 = Scalar(1.0) - c;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) {
        // :51-52.  Association as Eigen 3.4.0 (what ubuntu-22.04's libeigen3-dev, the CI's
        // Eigen, ships; the reference pins no version) evaluates `A + (c K) K` into a noalias
        // destination: dst = A, then dst += c * (K.lazyProduct(K)) with the scalar factored out.
        Scalar kk = 0;
        for (int k = 0; k < 3; ++k) kk += K.v[i][k] * K.v[k][j];
        R.v[i][j] = ((i == j ? Scalar(1) : Scalar(0)) + s * K.v[i][j]) + c1 * kk;
      }
    }
  } else {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R.v[i][j] = (i == j) ? Scalar(1) : Scalar(0);  // :54
  }
  return R;
}

// so3::Log (/root/reference/src/so3.cpp:94-105): theta = 0 when trace > 3 - 1e-6, else
// acos((trace - 1) / 2); delta = K / 2 for |theta| < 1e-3, else theta / (2 sin theta) K.
template <typename Scalar>
inline void Log(const Mat3<Scalar> &R, Scalar delta[3]) {
  const Scalar trace = R.v[0][0] + R.v[1][1] + R.v[2][2];
  const Scalar theta = (trace > 3.0 - 1e-6) ? Scalar(0.0) : std::acos(Scalar(0.5) * (trace - 1));  // :97
  const Scalar K[3] = {R.v[2][1] - R.v[1][2], R.v[0][2] - R.v[2][0], R.v[1][0] - R.v[0][1]};      // :98
  const Scalar k = (std::fabs(theta) < 0.001) ? Scalar(0.5) : Scalar(0.5) * theta / std::sin(theta);
  for (int i = 0; i < 3; ++i) delta[i] = k * K[i];  // :100, :102
}

// Column-major 4x4, as Eigen::Matrix<Scalar,4,4>::data().
template <typename Scalar>
inline void convert6DOFParameterToMatrix(const Scalar *x, Scalar T[16]) {
  for (int i = 0; i < 16; ++i) T[i] = 0;
  T[0] = T[5] = T[10] = T[15] = 1;  // setIdentity (:10)
  T[12] = x[0];                     // (0,3)
  T[13] = x[1];                     // (1,3)
  T[14] = x[2];                     // (2,3)
  const Scalar delta[3] = {x[3], x[4], x[5]};
  const Mat3<Scalar> R = Exp<Scalar>(delta);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T[c * 4 + r] = R.v[r][c];  // topLeftCorner (:18)
}

}  // namespace so3
}  // namespace oracle
