// Small column-major dense matrix: the storage behind covariance::Matrix at the boundary (the
// reference uses Eigen::Matrix<Scalar, Dynamic, Dynamic> —
// /root/reference/include/moptimizer/covariance/covariance.h:10-13; Eigen is not part of this
// build).  Method names follow the
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

}  // namespace dense
}  // namespace moptimizer
