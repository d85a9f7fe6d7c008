// Small column-major dense matrix + pivoted LDL^T, enough for the LM caller of the
// linearization path.  The reference leans on Eigen3 for this (Eigen::Matrix<Scalar,
// Dynamic, Dynamic>, Eigen::LDLT — /root/reference/src/levenberg_marquadt_dyn.cpp:78-80,
// include/moptimizer/covariance/covariance.h:10-13); Eigen is not part of this build, so the
// few operations the path's caller needs are written out here.  Method names follow the
// Eigen subset the reference calls (resize / setIdentity / setZero / data / rows / cols /
// operator()) so reference-style user code keeps compiling.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <vector>

namespace moptimizer {
namespace dense {

template <class Scalar>
class Matrix {
 public:
  Matrix() = default;
  Matrix(int rows, int cols) { resize(rows, cols); }

  // New storage is zero-filled (Eigen leaves it uninitialised; every caller on the path
  // overwrites it right away).
  void resize(int rows, int cols) {
    rows_ = rows;
    cols_ = cols;
    store_.assign(static_cast<std::size_t>(rows) * static_cast<std::size_t>(cols), Scalar(0));
  }
  void resize(int rows) { resize(rows, 1); }

  int rows() const { return rows_; }
  int cols() const { return cols_; }
  int size() const { return rows_ * cols_; }

  Scalar *data() { return store_.data(); }
  const Scalar *data() const { return store_.data(); }

  Scalar &operator()(int r, int c) { return store_[static_cast<std::size_t>(c) * rows_ + r]; }
  const Scalar &operator()(int r, int c) const {
    return store_[static_cast<std::size_t>(c) * rows_ + r];
  }
  // Linear (column-major) index, as Eigen's operator()(Index) on a plain matrix.
  Scalar &operator()(int i) { return store_[i]; }
  const Scalar &operator()(int i) const { return store_[i]; }
  Scalar &operator[](int i) { return store_[i]; }
  const Scalar &operator[](int i) const { return store_[i]; }

  void setZero() { std::fill(store_.begin(), store_.end(), Scalar(0)); }
  void setConstant(Scalar v) { std::fill(store_.begin(), store_.end(), v); }
  void setIdentity() {
    setZero();
    const int d = std::min(rows_, cols_);
    for (int i = 0; i < d; ++i) (*this)(i, i) = Scalar(1);
  }

  Matrix &operator*=(Scalar s) {
    for (auto &v : store_) v *= s;
    return *this;
  }
  Matrix &operator+=(const Matrix &o) {
    for (std::size_t i = 0; i < store_.size(); ++i) store_[i] += o.store_[i];
    return *this;
  }

  Scalar maxAbsCoeff() const {
    Scalar m = 0;
    for (const auto &v : store_) m = std::max(m, static_cast<Scalar>(std::fabs(v)));
    return m;
  }

  bool isSymmetric() const {
    if (rows_ != cols_) return false;
    for (int c = 0; c < cols_; ++c)
      for (int r = c + 1; r < rows_; ++r)
        if ((*this)(r, c) != (*this)(c, r)) return false;
    return true;
  }
  bool isIdentity() const {
    if (rows_ != cols_) return false;
    for (int c = 0; c < cols_; ++c)
      for (int r = 0; r < rows_; ++r)
        if ((*this)(r, c) != (r == c ? Scalar(1) : Scalar(0))) return false;
    return true;
  }

 private:
  int rows_ = 0;
  int cols_ = 0;
  std::vector<Scalar> store_;
};

template <class Scalar>
using Vector = Matrix<Scalar>;

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
