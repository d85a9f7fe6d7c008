// TEST SUPPORT — not part of the product (libmoptimizer_hip.so and include/ do not use it).
//
// Pivoted LDL^T for the host restatement of the LM caller (levenberg_marquadt.hpp next to this
// file): the reference asks Eigen for it (/root/reference/src/levenberg_marquadt_dyn.cpp:78-80)
// and Eigen is not part of this image.
#pragma once

#include <cmath>
#include <limits>
#include <vector>

#include "moptimizer_amd/dense.hpp"

namespace moptimizer {
namespace dense {

// Symmetric-indefinite-tolerant LDL^T with diagonal pivoting, the factorisation the
// reference's LM inner loop asks Eigen for (levenberg_marquadt_dyn.cpp:78): at step k the
// largest remaining |diagonal| is swapped to position k, the column below it is scaled by the
// pivot, and the trailing block is updated.  solve() skips (zeroes) components whose pivot is
// below the smallest normal number, which is what makes a rank-deficient H + lambda*D come
// back with a finite step instead of inf/NaN.
template <class Scalar>
class PivotedLDLT {
 public:
  explicit PivotedLDLT(const Matrix<Scalar> &a) { compute(a); }

  void compute(const Matrix<Scalar> &a) {
    n_ = a.rows();
    lower_ = a;
    perm_.resize(n_);
    for (int i = 0; i < n_; ++i) perm_[i] = i;
    Matrix<Scalar> &m = lower_;
    std::vector<Scalar> scaled(n_);

    for (int k = 0; k < n_; ++k) {
      // pivot search on the remaining diagonal
      int piv = k;
      Scalar best = std::fabs(m(k, k));
      for (int i = k + 1; i < n_; ++i) {
        const Scalar v = std::fabs(m(i, i));
        if (v > best) {
          best = v;
          piv = i;
        }
      }
      if (piv != k) swapSymmetric(k, piv);

      // d_k = a_kk - sum_j l_kj^2 d_j ; the products l_kj d_j are kept for the column update
      Scalar dk = m(k, k);
      for (int j = 0; j < k; ++j) {
        scaled[j] = m(k, j) * m(j, j);
        dk -= m(k, j) * scaled[j];
      }
      m(k, k) = dk;

      for (int i = k + 1; i < n_; ++i) {
        Scalar v = m(i, k);
        for (int j = 0; j < k; ++j) v -= m(i, j) * scaled[j];
        m(i, k) = v;
      }
      if (std::fabs(dk) > Scalar(0)) {
        for (int i = k + 1; i < n_; ++i) m(i, k) /= dk;
      }
    }
  }

  // x = A^{-1} rhs (pseudo-inverse on vanishing pivots)
  Matrix<Scalar> solve(const Matrix<Scalar> &rhs) const {
    Matrix<Scalar> y(n_, 1);
    for (int i = 0; i < n_; ++i) y[i] = rhs[perm_[i]];
    // L y = P rhs
    for (int i = 0; i < n_; ++i) {
      Scalar v = y[i];
      for (int j = 0; j < i; ++j) v -= lower_(i, j) * y[j];
      y[i] = v;
    }
    const Scalar tiny = std::numeric_limits<Scalar>::min();
    for (int i = 0; i < n_; ++i) {
      const Scalar d = lower_(i, i);
      y[i] = (std::fabs(d) > tiny) ? y[i] / d : Scalar(0);
    }
    // L^T z = y
    for (int i = n_ - 1; i >= 0; --i) {
      Scalar v = y[i];
      for (int j = i + 1; j < n_; ++j) v -= lower_(j, i) * y[j];
      y[i] = v;
    }
    Matrix<Scalar> x(n_, 1);
    for (int i = 0; i < n_; ++i) x[perm_[i]] = y[i];
    return x;
  }

 private:
  // Exchange rows/columns a<b of the symmetric matrix held in the lower triangle, dragging the
  // already-computed L rows along.
  void swapSymmetric(int a, int b) {
    Matrix<Scalar> &m = lower_;
    for (int j = 0; j < a; ++j) std::swap(m(a, j), m(b, j));
    for (int i = b + 1; i < n_; ++i) std::swap(m(i, a), m(i, b));
    for (int i = a + 1; i < b; ++i) std::swap(m(i, a), m(b, i));
    std::swap(m(a, a), m(b, b));
    std::swap(perm_[a], perm_[b]);
  }

  int n_ = 0;
  Matrix<Scalar> lower_;
  std::vector<int> perm_;
};

}  // namespace dense
}  // namespace moptimizer
