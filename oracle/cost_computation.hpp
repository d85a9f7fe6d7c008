// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it.
//
// CPU restatement of the reference's linearization loops,
//   /root/reference/include/moptimizer/linearization.h
//     computeCost              :36-47      -> CostComputation::computeCost
//     parallelComputeCost      :49-63      -> CostComputation::parallelComputeCost
//     computeHessianNumerical  :65-124     -> CostComputation::computeHessianNumerical
//     computeHessian           :126-158    -> CostComputation::computeHessian
// in plain C++17 (the original needs Eigen3 + oneTBB, neither of which exists in this image, so
// the reference itself cannot be compiled here — see DESIGN.md).  Arithmetic is sequential in
// `Scalar`, in index order, exactly as the reference's single-threaded loops; the Jacobian
// scratch is row-major m x n (:17-18), the hessian output column-major n x n (:16,:131).
//
// Parity pin: the reference's own known-answer tests, replayed by
// oracle/replay_reference_tests.cpp (list in that file's header).
#pragma once

#include <cmath>
#include <cstdlib>
#include <limits>
#include <thread>
#include <vector>

#include "moptimizer_amd/host_api.hpp"
#include "so3_ref.hpp"

namespace oracle {

template <class Scalar>
class CostComputation {
 public:
  using ModelPtr = typename moptimizer::IBaseModel<Scalar>::Ptr;
  using LossPtr = typename moptimizer::loss::ILossFunction<Scalar>::Ptr;

  CostComputation(int parameter_dim, int output_dim)
      : n_(parameter_dim),
        m_(output_dim),
        residuals_(output_dim),
        residuals_plus_(output_dim),
        jacobian_(static_cast<std::size_t>(output_dim) * parameter_dim),
        sj_(static_cast<std::size_t>(output_dim) * parameter_dim),
        sr_(output_dim) {}

  // linearization.h:36-47
  Scalar computeCost(const Scalar *x, ModelPtr model, int num_elements) {
    model->setup(x);
    Scalar sum = 0.0;
    for (int i = 0; i < num_elements; ++i) {
      if (model->f(x, residuals_.data(), i)) sum += dot(residuals_.data(), residuals_.data(), m_);
    }
    return sum;
  }

  // linearization.h:49-63.  The reference hands every TBB worker the same residual scratch
  // (a data race, :56); here each worker owns its scratch and the partials are joined with
  // plus<Scalar> in worker order.  Identity is the float literal 0.0f as at :53.
  // TBB keeps a pool of workers and splits the range by grain; this restatement starts
  // std::threads per call, so the worker count is capped at one per kParallelGrain elements and a
  // single worker runs on the calling thread: otherwise a small problem times pthread_create (256
  // launches per call on a 256-core host: 50 ms for 1 k points), not the reference's loop.
  static constexpr int kParallelGrain = 4096;
  Scalar parallelComputeCost(const Scalar *x, ModelPtr model, int num_elements,
                             int num_threads = 0) {
    model->setup(x);
    int workers = num_threads > 0 ? num_threads : int(std::thread::hardware_concurrency());
    const int by_grain = (num_elements + kParallelGrain - 1) / kParallelGrain;
    if (num_threads <= 0 && workers > by_grain) workers = by_grain;
    if (workers > num_elements) workers = num_elements;
    if (workers < 1) workers = 1;
    std::vector<Scalar> partial(workers, Scalar(0.0f));
    std::vector<std::thread> pool;
    const int m = m_;
    auto work = [&](int t) {
      const long long lo = (long long)num_elements * t / workers;
      const long long hi = (long long)num_elements * (t + 1) / workers;
      std::vector<Scalar> r(m);
      Scalar init = Scalar(0.0f);
      for (long long it = lo; it < hi; ++it) {
        if (model->f(x, r.data(), (unsigned int)it)) init += dot(r.data(), r.data(), m);
      }
      partial[t] = init;
    };
    if (workers == 1) {
      work(0);
    } else {
      for (int t = 0; t < workers; ++t) pool.emplace_back(work, t);
    }
    for (auto &th : pool) th.join();
    Scalar total = Scalar(0.0f);
    for (int t = 0; t < workers; ++t) total = total + partial[t];
    return total;
  }

  // linearization.h:65-124
  Scalar computeHessianNumerical(const Scalar *x, const Scalar *covariance_data, LossPtr loss,
                                 Scalar *hessian_data, Scalar *b_data, ModelPtr model,
                                 int num_elements) {
    Scalar sum = 0.0;
    const Scalar min_step_size = std::sqrt(std::numeric_limits<Scalar>::epsilon());  // :78

    std::vector<Scalar> h(n_);
    std::vector<std::vector<Scalar>> x_plus(n_, std::vector<Scalar>(x, x + n_));
    std::vector<ModelPtr> models_plus(n_);
    for (int j = 0; j < n_; ++j) {
      // :85 — unqualified abs(); with Eigen's headers in scope it resolves to the
      // floating-point overload on the reference's platform, hence fabs.
      forwardStep<Scalar>(x[j], min_step_size, &h[j], &x_plus[j][j]);  // :85-89 (so3_ref.hpp)
      models_plus[j] = model->clone();        // :91
      models_plus[j]->setup(x_plus[j].data());
    }

    model->setup(x);  // :95
    zero(hessian_data, n_ * n_);
    zero(b_data, n_);

    for (int i = 0; i < num_elements; ++i) {
      if (model->f(x, residuals_.data(), i)) {  // :102
        for (int j = 0; j < n_; ++j) {
          models_plus[j]->f(x_plus[j].data(), residuals_plus_.data(), i);  // result ignored, :104
          for (int r = 0; r < m_; ++r)
            jacobian_[r * n_ + j] = (residuals_plus_[r] - residuals_[r]) / h[j];  // :105
        }
        const Scalar w = loss->weight(dot(residuals_.data(), residuals_.data(), m_));  // :108
        accumulate(w, covariance_data, hessian_data, b_data);                         // :113-114
        sum += dot(residuals_.data(), residuals_.data(), m_);                         // :115
      }
    }
    return sum;
  }

  // linearization.h:126-158
  Scalar computeHessian(const Scalar *x, const Scalar *covariance_data, LossPtr loss,
                        Scalar *hessian_data, Scalar *b_data, ModelPtr model, int num_elements) {
    Scalar sum = 0.0;
    model->setup(x);  // :137
    zero(hessian_data, n_ * n_);
    zero(b_data, n_);
    for (int i = 0; i < num_elements; ++i) {
      if (model->f_df(x, residuals_.data(), jacobian_.data(), i)) {                     // :144
        const Scalar w = loss->weight(dot(residuals_.data(), residuals_.data(), m_));  // :145
        accumulate(w, covariance_data, hessian_data, b_data);                         // :150-151
        sum += dot(residuals_.data(), residuals_.data(), m_);                         // :152
      }
    }
    return sum;  // unweighted, no covariance (:157)
  }

 private:
  static Scalar dot(const Scalar *a, const Scalar *b, int k) {
    Scalar s = 0;
    for (int i = 0; i < k; ++i) s += a[i] * b[i];
    return s;
  }
  static void zero(Scalar *p, int k) {
    for (int i = 0; i < k; ++i) p[i] = 0;
  }

  // H += w * J^T * S * J ; b += w * J^T * S * r   with S (m x m) column-major, J row-major.
  // Evaluated as (w J^T) (S J) and (w J^T) (S r); any association agrees to rounding with
  // Eigen's nested small products (:113-114, :150-151).
  void accumulate(Scalar w, const Scalar *cov, Scalar *H, Scalar *b) {
    for (int a = 0; a < m_; ++a) {
      for (int j = 0; j < n_; ++j) {
        Scalar v = 0;
        for (int c = 0; c < m_; ++c) v += cov[c * m_ + a] * jacobian_[c * n_ + j];
        sj_[a * n_ + j] = v;
      }
      Scalar v = 0;
      for (int c = 0; c < m_; ++c) v += cov[c * m_ + a] * residuals_[c];
      sr_[a] = v;
    }
    for (int j = 0; j < n_; ++j) {
      for (int i = 0; i < n_; ++i) {
        Scalar v = 0;
        for (int a = 0; a < m_; ++a) v += (w * jacobian_[a * n_ + i]) * sj_[a * n_ + j];
        H[j * n_ + i] += v;
      }
    }
    for (int i = 0; i < n_; ++i) {
      Scalar v = 0;
      for (int a = 0; a < m_; ++a) v += (w * jacobian_[a * n_ + i]) * sr_[a];
      b[i] += v;
    }
  }

  int n_;
  int m_;
  std::vector<Scalar> residuals_;
  std::vector<Scalar> residuals_plus_;
  std::vector<Scalar> jacobian_;  // row-major m x n
  std::vector<Scalar> sj_;        // S * J, row-major m x n
  std::vector<Scalar> sr_;        // S * r
};

}  // namespace oracle
