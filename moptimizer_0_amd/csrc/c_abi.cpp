// C ABI of libmoptimizer_hip.so (declared in include/moptimizer_hip.h): cost-object lifetime,
// the once-per-x host work (SE(3) transforms, forward-difference steps), kernel selection and
// the RCCL-combined device group.  There is no CPU implementation behind any entry point: when
// HIP cannot run the work the call returns an error code.
#include "cost_state.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <atomic>
#include <cstdio>
#include <mutex>
#include <new>

#include "moptimizer_amd/so3.hpp"

using namespace mopt_detail;

namespace {
thread_local std::string g_last_error;
}  // namespace
namespace mopt_detail {
thread_local bool g_creating_for_search = false;
}

int mopt_detail::fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

namespace {


inline int resultCount(const mopt_cost *c) { return c->n_params * c->n_params + c->n_params + 1; }
inline int costOffset(const mopt_cost *c) { return c->n_params * c->n_params + c->n_params; }

// MOPT_BLOCKS_PER_CU (tuning override of the workgroups launched per CU) is read once per process.
int blocksPerCu(int fallback) {
  static const int forced = envInt("MOPT_BLOCKS_PER_CU", 0);
  return forced > 0 ? forced : fallback;
}

int gridFor(const mopt_cost *c, int blocks_per_cu) {
  long long g = (long long)c->num_cus * blocks_per_cu;
  if (g > c->num_tiles) g = c->num_tiles;
  if (g > c->max_grid) g = c->max_grid;
  if (g < 1) g = 1;
  return int(g);
}

// Every pair leaves pending_events exactly once, whatever fails: a pair whose timing cannot be read
// is dropped from the statistics and its events destroyed (not recycled: they may still be pending).
int resolvePendingEvents(mopt_cost *c) {
  int rc = MOPT_OK;
  for (auto &pr : c->pending_events) {
    float ms = 0.f;
    hipError_t e = rc == MOPT_OK ? hipEventSynchronize(pr.second) : hipErrorUnknown;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.first, pr.second);
    if (e == hipSuccess) {
      c->sweep_ms_total += double(ms);
      c->sweep_launches += 1;
      c->free_events.push_back(pr.first);
      c->free_events.push_back(pr.second);
    } else {
      if (rc == MOPT_OK)
        rc = fail(MOPT_ERR_HIP, std::string("sweep timing events: ") + hipGetErrorString(e));
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  }
  c->pending_events.clear();
  return rc;
}

// Times the dominant kernel of a sweep with a pair of HIP events on the launch stream when
// profiling samples this launch.  The moments / cost sweeps have their dispatch timestamped into
// the events (hipExtLaunchKernelGGL: agrees with rocprofv3's kernel trace to 0.2 us, costs the
// host ~10 us, which is why launches are sampled); the other sweeps get a recorded pair around
// the launch (reads 1-3 us longer than the kernel, costs ~6 us).
struct SweepTimer {
  mopt_cost *c;
  mopt::LaunchSite site;
  hipEvent_t start = nullptr, stop_ev = nullptr;
  bool dispatch_stamped = false;
  bool aql_used = false;  // the sweep kernel went to the cost's own queue (aql.hpp): so must its finalize
  mopt_detail::AqlSite finalize_site;  // the same queue, never timed (the timing is the sweep's)
  const mopt_detail::AqlSite *aql() {
    if (!aql_used) return nullptr;
    c->stat_direct_sweeps += 1;
    c->sweep_went_direct = true;
    finalize_site = site.aql;
    finalize_site.timed_for = nullptr;
    return &finalize_site;
  }
  SweepTimer(const SweepTimer &) = delete;
  SweepTimer(mopt_cost *cost, hipStream_t stream, bool stamp_dispatch = false) : c(cost) {
    site.stream = stream;
    if (cost->aql_now.queue && stream == cost->stream) {
      site.aql = cost->aql_now;
      site.aql_used = &aql_used;
    }
    // Streaming (non-temporal) loads once the tiles exceed the 32 MiB of aggregate L2: measured
    // faster both beyond the 256 MiB Infinity Cache (10 M points: 82 -> 77 us) and inside it
    // (1 M points: 12.3 -> 11.6 us); below that the default policy lets sweeps re-hit L2.
    static const int force = envInt("MOPT_STREAMING_LOADS", 0);  // 1 = never, 2 = always (tuning)
    const size_t bytes = size_t(cost->count) * 6 * size_t(cost->scalar_bytes);
    site.streaming = force == 2 || (force != 1 && bytes > (size_t(32) << 20));
    if (cost->profiling <= 0 || (cost->profiling_tick++ % cost->profiling) != 0) return;
    if (site.aql.queue && stamp_dispatch) {
      // the direct path stamps its own dispatch: the packet processor's start and end of this very
      // kernel, read after the call's results have arrived (blockingSweep)
      site.aql.timed_for = cost;
      cost->aql_timed = true;
      return;
    }
    site.aql = mopt_detail::AqlSite();  // (a sweep timed with recorded events goes to the stream)
    site.aql_used = nullptr;
    if (cost->pending_events.size() >= 4096 && resolvePendingEvents(cost) != MOPT_OK) return;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (auto &e : ev) {
      if (!cost->free_events.empty()) {
        e = cost->free_events.back();
        cost->free_events.pop_back();
      } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
      }
    }
    if (!ev[0] || !ev[1]) {  // keep the one that was obtained
      for (auto e : ev)
        if (e) cost->free_events.push_back(e);
      return;
    }
    if (stamp_dispatch) {
      start = site.time_start = ev[0];
      stop_ev = site.time_stop = ev[1];
      dispatch_stamped = true;
    } else if (hipEventRecord(ev[0], stream) == hipSuccess) {
      start = ev[0];
      stop_ev = ev[1];
    } else {
      cost->free_events.push_back(ev[0]);
      cost->free_events.push_back(ev[1]);
    }
  }
  void stop() {
    if (!start || !stop_ev) return;
    if (dispatch_stamped || hipEventRecord(stop_ev, site.stream) == hipSuccess)
      c->pending_events.emplace_back(start, stop_ev);
  }
};

template <typename S>
void defaultReprojConstants(double K[12], double C[16]) {
  // tst/camera_calibration.cpp:22-30
  const double k[12] = {586.122314453125, 0, 638.8477694496105, 0, 0, 722.3973388671875,
                        323.031267074588, 0, 0, 0, 1, 0};
  std::memcpy(K, k, sizeof k);
  const double a = M_PI_2;
  const double rx[3][3] = {{1, 0, 0}, {0, std::cos(a), -std::sin(a)}, {0, std::sin(a), std::cos(a)}};
  const double rz[3][3] = {{std::cos(a), -std::sin(a), 0}, {std::sin(a), std::cos(a), 0}, {0, 0, 1}};
  for (int i = 0; i < 16; ++i) C[i] = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C[r * 4 + c] = rx[r][0] * rz[0][c] + rx[r][1] * rz[1][c] + rx[r][2] * rz[2][c];
  C[15] = 1.0;
}

// Forward-difference points of the reference (linearization.h:78-92): h_j = sqrt(eps) |x_j|,
// or sqrt(eps) when that is zero; x_plus_j = x + h_j e_j.
template <typename S>
void forwardSteps(const S *x, S h[kNumParams], S x_plus[kNumParams][kNumParams]) {
  MOPT_SO3_EXACT_BODY  // x + h is a rounded product added to x, never a fused multiply-add
  const S min_step = std::sqrt(std::numeric_limits<S>::epsilon());
  for (int j = 0; j < kNumParams; ++j) {
    h[j] = min_step * std::fabs(x[j]);
    if (h[j] == S(0)) h[j] = min_step;
    for (int k = 0; k < kNumParams; ++k) x_plus[j][k] = x[k];
    x_plus[j][j] += h[j];
  }
}

template <typename S>
void fillP2PArgs(const mopt_cost *c, const S *x, bool with_steps, mopt::P2PSweepArgs<S> &a) {
  a.tiles = static_cast<const S *>(c->d_tiles);
  a.count = c->count;
  a.num_tiles = c->num_tiles;
  a.loss_kind = c->loss_kind;
  a.loss_param = S(c->loss_param);
  a.partials = c->d_partials;
  const auto T0 = moptimizer::so3::rigidFrom6DOF<S>(x);
  std::memcpy(a.T[0], T0.m, sizeof T0.m);
  for (int j = 0; j < kNumParams; ++j) {
    std::memcpy(a.T[1 + j], T0.m, sizeof T0.m);
    a.inv_h[j] = S(0);
  }
  if (with_steps) {
    S h[kNumParams], xp[kNumParams][kNumParams];
    forwardSteps<S>(x, h, xp);
    for (int j = 0; j < kNumParams; ++j) {
      const auto Tj = moptimizer::so3::rigidFrom6DOF<S>(xp[j]);
      std::memcpy(a.T[1 + j], Tj.m, sizeof Tj.m);
      a.inv_h[j] = S(1) / h[j];
    }
  }
  for (int k = 0; k < 9; ++k) a.cov[k] = S(c->cov[k]);
}

// The 3x6 row-major Jacobian pattern of a point, host copy of the device formulas; used to
// derive the affine basis J(p) = J0 + px Jx + py Jy + pz Jz.
void analyticPattern(int jac_mode, const double p[3], double J[18]) {
  for (int k = 0; k < 18; ++k) J[k] = 0.0;
  if (jac_mode == MOPT_JAC_ANALYTIC) {
    J[0 * 6 + 0] = 1; J[1 * 6 + 1] = 1; J[2 * 6 + 2] = 1;
    J[0 * 6 + 4] = p[2];  J[0 * 6 + 5] = -p[1];
    J[1 * 6 + 3] = -p[2]; J[1 * 6 + 5] = p[0];
    J[2 * 6 + 3] = p[1];  J[2 * 6 + 4] = -p[0];
  } else {
    J[0 * 6 + 0] = 1; J[0 * 6 + 4] = 1;
    J[1 * 6 + 2] = 1; J[1 * 6 + 4] = -p[2]; J[1 * 6 + 5] = p[1];
    J[2 * 6 + 0] = p[2]; J[2 * 6 + 2] = -p[0]; J[2 * 6 + 3] = -p[1]; J[2 * 6 + 4] = p[0];
  }
}

template <typename S>
void fillBasis(const mopt_cost *c, int jac_mode, const mopt::P2PSweepArgs<S> &a,
               mopt::AffineBasis &B) {
  if (jac_mode == MOPT_JAC_NUMERIC) {
    // column j of J is ((R_j - R) p + (t_j - t)) / h_j
    for (int r = 0; r < 3; ++r)
      for (int j = 0; j < kNumParams; ++j) {
        B.J[0][r * 6 + j] = double((a.T[1 + j][r * 4 + 3] - a.T[0][r * 4 + 3]) * a.inv_h[j]);
        for (int k = 0; k < 3; ++k)
          B.J[1 + k][r * 6 + j] = double((a.T[1 + j][r * 4 + k] - a.T[0][r * 4 + k]) * a.inv_h[j]);
      }
  } else if (jac_mode == MOPT_JAC_ANALYTIC_LEFT) {
    // J(p) = [I | -skew(R p + t)] is the conformant pattern taken at the warped point, which is
    // affine in p: J0 at t, J_k the change along column k of R
    double w[3] = {double(a.T[0][3]), double(a.T[0][7]), double(a.T[0][11])};
    analyticPattern(MOPT_JAC_ANALYTIC, w, B.J[0]);
    for (int k = 0; k < 3; ++k) {
      const double wk[3] = {w[0] + double(a.T[0][0 * 4 + k]), w[1] + double(a.T[0][1 * 4 + k]),
                            w[2] + double(a.T[0][2 * 4 + k])};
      analyticPattern(MOPT_JAC_ANALYTIC, wk, B.J[1 + k]);
      for (int q = 0; q < 18; ++q) B.J[1 + k][q] -= B.J[0][q];
    }
  } else if (jac_mode == MOPT_JAC_ANALYTIC_RIGHT) {
    // J(p) = [I | -R skew(p)]: J0 = [I | 0], J_k = [0 | -R skew(e_k)]
    for (int q = 0; q < 18; ++q) B.J[0][q] = B.J[1][q] = B.J[2][q] = B.J[3][q] = 0.0;
    B.J[0][0 * 6 + 0] = B.J[0][1 * 6 + 1] = B.J[0][2 * 6 + 2] = 1.0;
    for (int k = 0; k < 3; ++k) {
      double e[3] = {0, 0, 0};
      e[k] = 1.0;
      const double skew[3][3] = {{0, -e[2], e[1]}, {e[2], 0, -e[0]}, {-e[1], e[0], 0}};
      for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col) {
          double v = 0.0;
          for (int m = 0; m < 3; ++m) v += double(a.T[0][r * 4 + m]) * skew[m][col];
          B.J[1 + k][r * 6 + 3 + col] = -v;
        }
    }
  } else {
    const double origin[3] = {0, 0, 0};
    analyticPattern(jac_mode, origin, B.J[0]);
    for (int k = 0; k < 3; ++k) {
      double e[3] = {0, 0, 0};
      e[k] = 1.0;
      analyticPattern(jac_mode, e, B.J[1 + k]);
      for (int q = 0; q < 18; ++q) B.J[1 + k][q] -= B.J[0][q];
    }
  }
  for (int k = 0; k < 9; ++k) B.cov[k] = c->cov[k];
}

// AUTO's choice for forward differences.  The moments sweep forms column j of J as
// ((R_j - R) p + (t_j - t)) / h_j — the reference's quotient without its per-point cancellation
// error eps |R p + t| / h_j.  With h_j = sqrt(eps) |x_j| (linearization.h:85) that error is part of
// what the reference computes once |x_j| is small: measured distance between the two evaluations
// 2e-8 / |x_j| (tests/tools/parity_table.py; the literal sweep matches the reference to 1e-14 at every
// step size).  So below |x_j| = 0.08, where 4 x that measurement would pass the 1e-6 bar, AUTO
// evaluates literally; x_j = 0 takes the fixed step sqrt(eps) and is fine.
template <typename S>
bool hasSmallForwardStep(const S *x) {
  for (int j = 0; j < kNumParams; ++j) {
    const double a = std::fabs(double(x[j]));
    if (a > 0.0 && a < 0.08) return true;
  }
  return false;
}

template <typename S>
int p2pLinearizeAsync(mopt_cost *c, int jac_mode, const S *x, double *d_result, hipStream_t s,
                      const mopt::HostPublish &pub) {
  mopt::P2PSweepArgs<S> args;
  fillP2PArgs<S>(c, x, jac_mode == MOPT_JAC_NUMERIC, args);
  bool moments;
  switch (c->variant) {
    case MOPT_KERNEL_LITERAL: moments = false; break;
    case MOPT_KERNEL_MOMENTS_ALWAYS: moments = true; break;
    default: moments = !(jac_mode == MOPT_JAC_NUMERIC && hasSmallForwardStep<S>(x)); break;
  }
  if (moments) {
    mopt::AffineBasis basis;
    fillBasis<S>(c, jac_mode, args, basis);
    const int grid = gridFor(c, blocksPerCu(1));
    SweepTimer timer(c, s, true);
    MOPT_HIP_TRY(mopt::launchP2PMoments<S>(args, grid, timer.site));
    timer.stop();
    MOPT_HIP_TRY(mopt::launchFinalizeMoments(c->d_partials, grid, basis, d_result, pub, s, c->launch_peers,
                                             timer.aql()));
  } else {
    // forward differences: two workgroups per CU (VALU-heavy: the second wave per SIMD pays); the
    // analytic patterns stream like the moments sweep: one (70.8 vs 72.1 us at 10 M, and half the
    // partial rows for the finalize kernel)
    const int grid = gridFor(c, blocksPerCu(jac_mode == MOPT_JAC_NUMERIC ? 2 : 1));
    const int nacc = c->cov_mode == mopt::kCovGeneral ? mopt::kAccFull : mopt::kAccSym;
    // (tiled launch signature: dispatch timestamps like the moments sweep)
    SweepTimer timer(c, s, true);
    MOPT_HIP_TRY(mopt::launchP2PLinearizeLiteral<S>(args, jac_mode, c->cov_mode, grid, timer.site));
    timer.stop();
    MOPT_HIP_TRY(mopt::launchFinalizeDense(c->d_partials, grid, nacc, kNumParams, d_result, pub, s,
                                           c->launch_peers, timer.aql()));
  }
  return MOPT_OK;
}

template <typename S>
int p2pCostAsync(mopt_cost *c, const S *x, double *d_sum, hipStream_t s,
                 const mopt::HostPublish &pub) {
  mopt::P2PSweepArgs<S> args;
  fillP2PArgs<S>(c, x, false, args);
  const int grid = gridFor(c, blocksPerCu(1));
  SweepTimer timer(c, s, true);
  MOPT_HIP_TRY(mopt::launchP2PCost<S>(args, grid, timer.site));
  timer.stop();
  MOPT_HIP_TRY(mopt::launchFinalizeCost(c->d_partials, grid, d_sum, pub, s, c->launch_peers, timer.aql()));
  return MOPT_OK;
}

// (K * T) * C, row-major 3x4, the matrix products of tst/camera_calibration.cpp:37.
void projectionFor(const mopt_cost *c, const double *x, double M[12]) {
  MOPT_SO3_EXACT_BODY  // bit-identical to the oracle's products: forward differences amplify an ulp
  const auto T = moptimizer::so3::rigidFrom6DOF<double>(x);
  double T4[16];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 4; ++k) T4[r * 4 + k] = T.m[r * 4 + k];
  T4[12] = T4[13] = T4[14] = 0.0;
  T4[15] = 1.0;
  double KT[12];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 4; ++k) {
      double v = 0.0;
      for (int q = 0; q < 4; ++q) v += c->camera[r * 4 + q] * T4[q * 4 + k];
      KT[r * 4 + k] = v;
    }
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 4; ++k) {
      double v = 0.0;
      for (int q = 0; q < 4; ++q) v += KT[r * 4 + q] * c->frame[q * 4 + k];
      M[r * 4 + k] = v;
    }
}

void fillReprojArgs(const mopt_cost *c, const double *x, bool with_steps,
                    mopt::ReprojSweepArgs &a) {
  a.tiles = static_cast<const unsigned char *>(c->d_tiles);
  a.count = c->count;
  a.num_tiles = c->num_tiles;
  a.loss_kind = c->loss_kind;
  a.loss_param = c->loss_param;
  a.partials = c->d_partials;
  projectionFor(c, x, a.M[0]);
  for (int j = 0; j < kNumParams; ++j) {
    std::memcpy(a.M[1 + j], a.M[0], sizeof a.M[0]);
    a.inv_h[j] = 0.0;
  }
  if (with_steps) {
    double h[kNumParams], xp[kNumParams][kNumParams];
    forwardSteps<double>(x, h, xp);
    for (int j = 0; j < kNumParams; ++j) {
      projectionFor(c, xp[j], a.M[1 + j]);
      a.inv_h[j] = 1.0 / h[j];
    }
  }
  // row-major 2x2 out of the 3x3 slot
  a.cov[0] = c->cov[0]; a.cov[1] = c->cov[1];
  a.cov[2] = c->cov[3]; a.cov[3] = c->cov[4];
}

int reprojLinearizeAsync(mopt_cost *c, int jac_mode, const double *x, double *d_result,
                         hipStream_t s, const mopt::HostPublish &pub) {
  if (jac_mode != MOPT_JAC_NUMERIC)
    return fail(MOPT_ERR_UNSUPPORTED,
                "the reprojection model has no analytic Jacobian (BaseModel, numeric only)");
  mopt::ReprojSweepArgs args;
  fillReprojArgs(c, x, true, args);
  const int grid = gridFor(c, blocksPerCu(2));
  const int nacc = c->cov_mode == mopt::kCovGeneral ? mopt::kAccFull : mopt::kAccSym;
  SweepTimer timer(c, s, true);
  MOPT_HIP_TRY(mopt::launchReprojLinearize(args, c->cov_mode, grid, timer.site));
  timer.stop();
  MOPT_HIP_TRY(mopt::launchFinalizeDense(c->d_partials, grid, nacc, kNumParams, d_result, pub, s,
                                         c->launch_peers, timer.aql()));
  return MOPT_OK;
}

int reprojCostAsync(mopt_cost *c, const double *x, double *d_sum, hipStream_t s,
                    const mopt::HostPublish &pub) {
  mopt::ReprojSweepArgs args;
  fillReprojArgs(c, x, false, args);
  const int grid = gridFor(c, blocksPerCu(2));
  SweepTimer timer(c, s, true);
  MOPT_HIP_TRY(mopt::launchReprojCost(args, grid, timer.site));
  timer.stop();
  MOPT_HIP_TRY(mopt::launchFinalizeCost(c->d_partials, grid, d_sum, pub, s, c->launch_peers, timer.aql()));
  return MOPT_OK;
}

template <typename S>
void fillScalarArgs(const mopt_cost *c, const S *x, mopt::ScalarSweepArgs<S> &a) {
  a.data = static_cast<const S *>(c->d_tiles);
  a.count = c->count;
  a.stride = c->data_stride;
  a.loss_kind = c->loss_kind;
  a.loss_param = S(c->loss_param);
  a.partials = c->d_partials;
  const S min_step = std::sqrt(std::numeric_limits<S>::epsilon());  // linearization.h:78
  for (int j = 0; j < mopt::kMaxParams; ++j) {
    a.x[j] = j < c->n_params ? x[j] : S(0);
    S h = min_step * std::fabs(a.x[j]);  // :85
    if (h == S(0)) h = min_step;         // :87
    a.h[j] = h;
  }
  for (int k = 0; k < 16; ++k) a.cov[k] = S(c->cov_m[k]);
}

bool scalarModelHasJacobian(int kind) {
  return kind != mopt::kScalarExpCurve && kind != mopt::kScalarExpCurveMarked;
}

// workgroups (= partial rows) of a built-in scalar model's sweep: a workgroup per 256 packs of 16
// bytes, at most four per CU
int scalarGrid(const mopt_cost *c) {
  const long long per_block = (long long)mopt::kBlockThreads * (16 / c->scalar_bytes);
  long long blocks = (c->count + per_block - 1) / per_block;
  if (blocks > c->num_cus * 4) blocks = c->num_cus * 4;
  return blocks < 1 ? 1 : int(blocks);
}

template <typename S>
int scalarSweepAsync(mopt_cost *c, bool cost_only, int jac_mode, const S *x, double *d_out,
                     hipStream_t s, const mopt::HostPublish &pub) {
  if (!cost_only) {
    if (jac_mode == MOPT_JAC_ANALYTIC_TST_LAYOUT || jac_mode == MOPT_JAC_ANALYTIC_LEFT ||
        jac_mode == MOPT_JAC_ANALYTIC_RIGHT)
      return fail(MOPT_ERR_UNSUPPORTED,
                  "the as-written layout and the perturbation Jacobians apply to point2point only");
    if (jac_mode == MOPT_JAC_ANALYTIC && !scalarModelHasJacobian(c->scalar_model))
      return fail(MOPT_ERR_UNSUPPORTED,
                  "Non implemented non-jacobian model function `f_df` being used.");
  }
  mopt::ScalarSweepArgs<S> args;
  fillScalarArgs<S>(c, x, args);
  const int grid = scalarGrid(c);
  const int n = c->n_params;
  SweepTimer timer(c, s);
  MOPT_HIP_TRY(mopt::launchScalarModel<S>(args, c->scalar_model, cost_only, jac_mode, c->cov_mode,
                                          grid, timer.site));
  timer.stop();
  if (cost_only) {
    MOPT_HIP_TRY(mopt::launchFinalizeCost(c->d_partials, grid, d_out, pub, s, c->launch_peers, timer.aql()));
  } else {
    const int nacc = c->cov_mode == mopt::kCovGeneral ? n * n + n + 1 : n * (n + 1) / 2 + n + 1;
    MOPT_HIP_TRY(mopt::launchFinalizeDense(c->d_partials, grid, nacc, n, d_out, pub, s, c->launch_peers,
                                           timer.aql()));
  }
  return MOPT_OK;
}

template <typename S>
void fillJitArgs(const mopt_cost *c, const S *x, mopt::JitArgs<S> &args) {
  mopt::ScalarSweepArgs<S> filled;
  fillScalarArgs<S>(c, x, filled);
  args.data = filled.data;
  args.count = filled.count;
  args.stride = filled.stride;
  args.loss_kind = filled.loss_kind;
  args.pad_[0] = args.pad_[1] = args.pad_[2] = 0;
  args.loss_param = filled.loss_param;
  for (int k = 0; k < 8; ++k) {
    args.x[k] = filled.x[k];
    args.h[k] = filled.h[k];
  }
  for (int k = 0; k < 16; ++k) args.cov[k] = filled.cov[k];
  args.partials = c->d_partials;
}

template <typename S>
void fillJitWideArgs(const mopt_cost *c, const S *x, mopt::JitWideArgs<S> &args) {
  args.data = static_cast<const S *>(c->d_tiles);
  args.count = c->count;
  args.stride = c->data_stride;
  args.loss_kind = c->loss_kind;
  args.pad_[0] = args.pad_[1] = args.pad_[2] = 0;
  args.loss_param = S(c->loss_param);
  const S min_step = std::sqrt(std::numeric_limits<S>::epsilon());  // linearization.h:78
  for (int j = 0; j < mopt::kMaxWideParams; ++j) {
    args.x[j] = j < c->n_params ? x[j] : S(0);
    S h = min_step * std::fabs(args.x[j]);  // :85
    if (h == S(0)) h = min_step;            // :87
    args.h[j] = h;
  }
  for (int k = 0; k < mopt::kMaxWideOutputs * mopt::kMaxWideOutputs; ++k) args.cov[k] = S(c->cov_m[k]);
  args.partials = c->d_partials;
}

int jitGrid(const mopt_cost *c) {
  if (c->jit.wide) {
    long long blocks = (c->count + mopt::kJitWideElementsPerBlock - 1) / mopt::kJitWideElementsPerBlock;
    if (blocks > c->num_cus * 2) blocks = c->num_cus * 2;  // rows of up to 273 doubles
    return blocks < 1 ? 1 : int(blocks);
  }
  const long long per_block = (long long)mopt::kBlockThreads * (16 / c->scalar_bytes);
  long long blocks = (c->count + per_block - 1) / per_block;
  if (blocks > c->num_cus * 4) blocks = c->num_cus * 4;
  if (blocks < 1) blocks = 1;
  return int(blocks);
}

// User-defined (hipRTC) model: same sweep shape as the built-in scalar models, full-form rows.
template <typename S>
int jitSweepAsync(mopt_cost *c, bool cost_only, int jac_mode, const S *x, double *d_out,
                  hipStream_t s, const mopt::HostPublish &pub) {
  if (!cost_only) {
    if (jac_mode == MOPT_JAC_ANALYTIC_TST_LAYOUT || jac_mode == MOPT_JAC_ANALYTIC_LEFT ||
        jac_mode == MOPT_JAC_ANALYTIC_RIGHT)
      return fail(MOPT_ERR_UNSUPPORTED,
                  "the as-written layout and the perturbation Jacobians apply to point2point only");
    if (jac_mode == MOPT_JAC_ANALYTIC && !c->jit.has_jacobian)
      return fail(MOPT_ERR_UNSUPPORTED,
                  "Non implemented non-jacobian model function `f_df` being used.");
  }
  const int mode = cost_only ? 0 : (jac_mode == MOPT_JAC_NUMERIC ? 2 : 1);
  const bool cov_symmetric = !c->jit.wide && c->cov_mode != mopt::kCovGeneral;
  const mopt::JitVariant *variant = mopt::jitVariant(c->jit, mode, c->cov_mode);
  if (!variant) return fail(MOPT_ERR_INVALID_ARGUMENT, mopt::jitLastError());
  const int grid = jitGrid(c);
  const int n = c->n_params;
  SweepTimer timer(c, s);
  if (c->jit.wide) {
    mopt::JitWideArgs<S> args;
    fillJitWideArgs<S>(c, x, args);
    MOPT_HIP_TRY(mopt::jitLaunch(*variant, &args, sizeof args, grid, s));
  } else {
    mopt::JitArgs<S> args;
    fillJitArgs<S>(c, x, args);
    MOPT_HIP_TRY(mopt::jitLaunch(*variant, &args, sizeof args, grid, s));
  }
  timer.stop();
  if (cost_only) {
    MOPT_HIP_TRY(mopt::launchFinalizeCost(c->d_partials, grid, d_out, pub, s, c->launch_peers));
  } else {
    const int nacc = cov_symmetric ? n * (n + 1) / 2 + n + 1 : n * n + n + 1;
    MOPT_HIP_TRY(mopt::launchFinalizeDense(c->d_partials, grid, nacc, n, d_out, pub, s, c->launch_peers));
  }
  return MOPT_OK;
}

}  // namespace

namespace mopt_detail {
int linearizeAsyncImpl(mopt_cost *c, int jac_mode, const void *x, double *d_result, hipStream_t s,
                       const mopt::HostPublish &pub) {
  if (jac_mode < MOPT_JAC_ANALYTIC || jac_mode > MOPT_JAC_ANALYTIC_RIGHT)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown jacobian_mode");
  if (c->model == kModelJit)
    return c->scalar_bytes == 8
               ? jitSweepAsync<double>(c, false, jac_mode, static_cast<const double *>(x), d_result,
                                       s, pub)
               : jitSweepAsync<float>(c, false, jac_mode, static_cast<const float *>(x), d_result, s,
                                      pub);
  if (c->model == kModelScalar)
    return c->scalar_bytes == 8
               ? scalarSweepAsync<double>(c, false, jac_mode, static_cast<const double *>(x),
                                          d_result, s, pub)
               : scalarSweepAsync<float>(c, false, jac_mode, static_cast<const float *>(x),
                                         d_result, s, pub);
  if (c->model == kModelReprojection)
    return reprojLinearizeAsync(c, jac_mode, static_cast<const double *>(x), d_result, s, pub);
  if (c->scalar_bytes == 8)
    return p2pLinearizeAsync<double>(c, jac_mode, static_cast<const double *>(x), d_result, s, pub);
  return p2pLinearizeAsync<float>(c, jac_mode, static_cast<const float *>(x), d_result, s, pub);
}

int costAsyncImpl(mopt_cost *c, const void *x, double *d_sum, hipStream_t s,
                  const mopt::HostPublish &pub) {
  if (c->model == kModelJit)
    return c->scalar_bytes == 8
               ? jitSweepAsync<double>(c, true, 0, static_cast<const double *>(x), d_sum, s, pub)
               : jitSweepAsync<float>(c, true, 0, static_cast<const float *>(x), d_sum, s, pub);
  if (c->model == kModelScalar)
    return c->scalar_bytes == 8
               ? scalarSweepAsync<double>(c, true, 0, static_cast<const double *>(x), d_sum, s, pub)
               : scalarSweepAsync<float>(c, true, 0, static_cast<const float *>(x), d_sum, s, pub);
  if (c->model == kModelReprojection)
    return reprojCostAsync(c, static_cast<const double *>(x), d_sum, s, pub);
  if (c->scalar_bytes == 8)
    return p2pCostAsync<double>(c, static_cast<const double *>(x), d_sum, s, pub);
  return p2pCostAsync<float>(c, static_cast<const float *>(x), d_sum, s, pub);
}

}  // namespace mopt_detail

namespace {
// Blocking completion without a copy or a stream synchronisation: the last kernel of the call
// stores the results into mapped host memory and then releases `sequence` into the flag word;
// the host polls that word.  The stream is queried now and then so that a faulted kernel turns
// into an error instead of an endless wait.
int waitPublished(mopt_cost *c, unsigned long long sequence) {
  unsigned long long spins = 0;
  const auto started = std::chrono::steady_clock::now();
  while (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) != sequence) {
    if ((++spins & 0x3fff) == 0) {
      if (c->waiting_direct) {
        // the sweep went through the cost's own queue (aql.hpp): nothing on the HIP stream to ask; a
        // faulted kernel reaches the queue's error callback
        if (mopt_detail::aqlFaulted(c->aql_queue))
          return fail(MOPT_ERR_HIP, "sweep failed: the direct-dispatch queue reported an error");
        const auto waited = std::chrono::steady_clock::now() - started;
        if (waited > std::chrono::seconds(60))
          return fail(MOPT_ERR_HIP, "timed out waiting for the sweep result (60 s)");
        // A sweep takes milliseconds at most.  The one known way for this wait to stall is a tool that
        // intercepts the queue and holds packets back (a per-dispatch counter profiler: aql.cpp recognises
        // rocprofv3's by its environment variables and stays off; one it does not recognise ends up here)
        // — say once what to do instead of waiting out the minute in silence.
        static std::atomic<bool> hinted{false};
        if (waited > std::chrono::seconds(3) && !hinted.exchange(true))
          std::fprintf(stderr,
                       "libmoptimizer_hip: a blocking sweep dispatched through the library's own HSA queue has "
                       "not completed after 3 s; if a tool is intercepting GPU queues (a counter-collecting "
                       "profiler), run with MOPT_AQL=0 to keep every launch on HIP streams\n");
        __builtin_ia32_pause();
        continue;
      }
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == sequence) break;
        return fail(MOPT_ERR_HIP, "stream drained without publishing a result");
      }
      if (q != hipErrorNotReady)
        return fail(MOPT_ERR_HIP, std::string("sweep failed: ") + hipGetErrorString(q));
      // a wedged device must not wedge the caller: give up after a generous bound
      if (std::chrono::steady_clock::now() - started > std::chrono::seconds(60))
        return fail(MOPT_ERR_HIP, "timed out waiting for the sweep result (60 s)");
    }
    __builtin_ia32_pause();
  }
  return MOPT_OK;
}

// The HIP runtime keeps every command it has queued on a stream alive until a marker behind it
// completes; then a thread of the runtime walks the whole batch and releases it (ROCclr's submission
// batch: a marker goes in at a synchronisation, a query, an event, or after
// DEBUG_CLR_MAX_BATCH_SIZE = 1000 commands).  A loop of blocking sweeps queues two kernels per call
// and never synchronises, so the releases come a thousand commands at a time, and for the ~0.5-1 ms
// each takes the launching thread runs 10-15 us per call slower (allocator and lock traffic with the
// releasing thread; scripts/probe_sync_effect.py: 40 against 24 us per step of two linked
// reprojection costs for the ~20 steps after a hipStreamSynchronize, 30 against 20 us at 1 M
// correspondences).  Handing the runtime a marker every few launches keeps the batches short: the
// release then costs a few microseconds on the other thread, spread evenly.  The query is issued
// after this call's kernels are queued, while they run; its status is not used.
void boundCommandBatch(mopt_cost *c) {
  static const int every = envInt("MOPT_MARKER_EVERY", 32);  // sweeps; a huge value switches it off
  if (c->waiting_direct) return;  // (the sweep bypassed the runtime: aql.hpp)
  if (++c->launches_since_marker < every) return;
  c->launches_since_marker = 0;
  (void)hipStreamQuery(c->stream);
}

// A profiled sweep that went through the direct path: once its results have arrived, the packet
// processor's own start and end of that dispatch (aql.hpp) go into the cost's sweep statistics.
void collectDirectTiming(mopt_cost *c, int rc) {
  if (!c->aql_timed) return;
  c->aql_timed = false;
  const double ns = rc == MOPT_OK ? mopt_detail::aqlDispatchNanoseconds(c->aql_queue, c) : -1.0;
  if (ns >= 0.0) {
    c->sweep_ms_total += ns * 1e-6;
    c->sweep_launches += 1;
  }
}

// Where the blocking sweep about to be launched goes: the cost's own queue (aql.hpp) unless the call
// needs the HIP stream — RCCL selected (its all-reduce launches there), a cost with a correspondence
// search (its update(x) is queued on the stream every outer iteration: a sweep on another queue would
// have to wait for it with a synchronisation), a model whose kernels live in hipRTC modules (run-time
// compiled).  Profiled sweeps go the same way as unprofiled ones (the queue stamps its own dispatch).
// A switch of path waits once for whatever the other path still has queued for this cost.
void chooseDispatchPath(mopt_cost *c) {
  c->aql_now = mopt_detail::AqlSite();
  // A shard of a multi-rank job (a combine transport attached) takes the direct path like any other
  // cost — the host-slot and peer combines live inside the finalize kernel and its wait — unless
  // MOPT_AQL_SHARDED=0: ranks that SHARE a GPU (rehearsals, tests) add two hardware queues per
  // process on it, and 4 ranks + their parent oversubscribed the GPU's queues until the combines'
  // bounded waits ran out (profiles/NOTES.md round 5).
  static const bool sharded_allowed = [] {
    const char *v = std::getenv("MOPT_AQL_SHARDED");
    return !(v && v[0] == '0');
  }();
  const bool sharded = c->combine.host_block != nullptr || c->combine.peer_attached || c->comm != nullptr;
  const bool eligible = (c->model == kModelPoint2Point || c->model == kModelReprojection ||
                         c->model == kModelScalar) &&
                        (c->combine.mode == MOPT_COMBINE_NONE || c->combine.mode == MOPT_COMBINE_HOST ||
                         c->combine.mode == MOPT_COMBINE_PEER) &&
                        (!sharded || sharded_allowed) && !c->matcher;
  if (eligible && !c->aql_tried) {
    c->aql_tried = true;
    c->aql_queue = mopt_detail::aqlAcquireQueue(c->device);
    if (c->aql_queue) {
      mopt_detail::AqlSite site;
      site.queue = c->aql_queue;
      site.device = c->device;
      if (!mopt::aqlFinalizersLoaded(site)) c->aql_queue = nullptr;
    }
  }
  if (eligible && c->aql_queue && !mopt_detail::aqlFaulted(c->aql_queue)) {
    if (c->hip_pending) {  // e.g. a correspondence search queued on the stream: the sweep reads its output
      (void)hipStreamSynchronize(c->stream);
      c->hip_pending = false;
    }
    c->aql_now.queue = c->aql_queue;
    c->aql_now.device = c->device;
    c->sweep_went_direct = false;
    c->waiting_direct = true;  // provisional: settleDispatchPath corrects it once the launch has happened
    return;
  }
  if (c->aql_touched && c->aql_queue) {  // (a superseded prefetch may still be queued there)
    (void)mopt_detail::aqlDrain(c->aql_queue);
    c->aql_touched = false;
  }
  c->hip_pending = true;
  c->waiting_direct = false;
  c->sweep_went_direct = false;
}

// After the launch: where the sweep really went.  chooseDispatchPath only OFFERS the direct path; the
// launch itself can still end up on the HIP stream — a profiled sweep of a model that is timed with a
// recorded event pair (scalar models), a kernel the loader lookup did not find.  Then the whole call is
// a HIP-stream call: the wait queries the stream (a faulted kernel is an error at once, not a 60 s
// time-out), command batches are bounded, and the next direct sweep first waits for the stream.
void settleDispatchPath(mopt_cost *c) {
  c->aql_now = mopt_detail::AqlSite();
  if (c->sweep_went_direct) {
    c->aql_touched = true;
    return;
  }
  if (c->waiting_direct) {
    c->waiting_direct = false;
    c->hip_pending = true;
    c->aql_timed = false;
  }
}

mopt::HostPublish nextPublish(mopt_cost *c, int offset) {
  mopt::HostPublish pub;
  pub.host_result = c->h_result_dev + offset;
  pub.host_flag = c->h_flag_dev;
  pub.sequence = ++c->sequence;
  return pub;
}

// The host side of MOPT_COMBINE_HOST: wait until every rank's finalize kernel has published
// sweep `sequence` into its slot of the shared block, then add the slots in rank order.
int waitAndSumHostSlots(mopt_cost *c, unsigned long long sequence, int offset, int count) {
  const ShardCombine &sc = c->combine;
  const auto started = std::chrono::steady_clock::now();
  const auto limit = std::chrono::milliseconds(envInt("MOPT_PEER_TIMEOUT_MS", 5000));
  unsigned long long spins = 0;
  for (int k = 0; k < sc.num_ranks; ++k) {
    const double *slot = sc.host_block + mopt::slotIndex(sequence, sc.num_ranks, k);
    const unsigned long long *flag =
        reinterpret_cast<const unsigned long long *>(slot + mopt::kSlotFlag);
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < sequence) {
      if ((++spins & 0x3fff) == 0) {
        // this rank's own sweep first: a local fault must not be reported as a peer that stays away
        if (c->waiting_direct) {
          if (mopt_detail::aqlFaulted(c->aql_queue))
            return fail(MOPT_ERR_HIP, "sweep failed: the direct-dispatch queue reported an error");
        } else {
          const hipError_t q = hipStreamQuery(c->stream);
          if (q != hipSuccess && q != hipErrorNotReady)
            return fail(MOPT_ERR_HIP, std::string("sweep failed: ") + hipGetErrorString(q));
        }
        if (std::chrono::steady_clock::now() - started > limit)
          return fail(MOPT_ERR_PEER_TIMEOUT, "rank " + std::to_string(k) +
                                                 " did not publish its sums (MOPT_PEER_TIMEOUT_MS)");
      }
      __builtin_ia32_pause();
    }
  }
  double *out = c->h_result + offset;
  for (int q = 0; q < count; ++q) out[q] = 0.0;
  for (int k = 0; k < sc.num_ranks; ++k) {
    const double *slot = sc.host_block + mopt::slotIndex(sequence, sc.num_ranks, k) + offset;
    for (int q = 0; q < count; ++q) out[q] += slot[q];
  }
  return MOPT_OK;
}

mopt::PeerCombine nextPeerCombine(mopt_cost *c, int offset) {
  ShardCombine &sc = c->combine;
  mopt::PeerCombine pc;
  for (int k = 0; k < sc.num_ranks; ++k) pc.blocks[k] = sc.peer_blocks[k];
  pc.rank = sc.rank;
  pc.num_ranks = sc.num_ranks;
  pc.offset = offset;
  pc.sequence = ++sc.sequence;
  pc.timeout_ticks = sc.peer_timeout_ticks;
  return pc;
}

// One blocking sweep on the cost's own stream: kernels, the sum over the ranks of a sharded cost
// (combine mode), and the published result in c->h_result[offset .. offset + count).
int blockingSweep(mopt_cost *c, bool cost_only, int jac_mode, const void *x) {
  const int offset = cost_only ? costOffset(c) : 0;
  const int count = cost_only ? 1 : resultCount(c);
  c->stat_sweeps += 1;
  chooseDispatchPath(c);
  auto launch = [&](const mopt::HostPublish &pub) {
    const int rc = cost_only ? costAsyncImpl(c, x, c->d_result + offset, c->stream, pub)
                             : linearizeAsyncImpl(c, jac_mode, x, c->d_result, c->stream, pub);
    settleDispatchPath(c);
    return rc;
  };
  switch (c->combine.mode) {
    case MOPT_COMBINE_RCCL: {  // also for a 1-rank communicator: same code path as N ranks
      int rc = launch(mopt::HostPublish());
      if (rc != MOPT_OK) return rc;
      MOPT_NCCL_TRY(ncclAllReduce(c->d_result + offset, c->d_result + offset, count, ncclDouble,
                                  ncclSum, c->comm, c->stream));
      mopt::HostPublish pub = nextPublish(c, offset);
      MOPT_HIP_TRY(mopt::launchPublish(c->d_result + offset, count, pub, c->stream));
      boundCommandBatch(c);
      return waitPublished(c, pub.sequence);
    }
    case MOPT_COMBINE_HOST: {
      ShardCombine &sc = c->combine;
      const unsigned long long seq = ++sc.sequence;
      double *slot = sc.host_block_dev + mopt::slotIndex(seq, sc.num_ranks, sc.rank);
      mopt::HostPublish pub;
      pub.host_result = slot + offset;
      pub.host_flag = reinterpret_cast<unsigned long long *>(slot + mopt::kSlotFlag);
      pub.sequence = seq;
      c->aql_timed = false;
      int rc = launch(pub);
      if (rc != MOPT_OK) return rc;
      boundCommandBatch(c);
      rc = waitAndSumHostSlots(c, seq, offset, count);
      collectDirectTiming(c, rc);
      return rc;
    }
    case MOPT_COMBINE_PEER: {
      const mopt::PeerCombine pc = nextPeerCombine(c, offset);
      mopt::HostPublish pub = nextPublish(c, offset);
      pub.host_status = c->h_flag_dev + 1;
      c->launch_peers = &pc;
      c->aql_timed = false;
      int rc = launch(pub);
      c->launch_peers = nullptr;
      if (rc != MOPT_OK) return rc;
      boundCommandBatch(c);
      rc = waitPublished(c, pub.sequence);
      collectDirectTiming(c, rc);
      if (rc != MOPT_OK) return rc;
      if (__atomic_load_n(c->h_flag + 1, __ATOMIC_ACQUIRE) == mopt::kStatusPeerTimeout)
        return fail(MOPT_ERR_PEER_TIMEOUT,
                    "a rank did not deliver its sums to this device in time (MOPT_PEER_TIMEOUT_MS)");
      return MOPT_OK;
    }
    default: {
      mopt::HostPublish pub = nextPublish(c, offset);
      c->aql_timed = false;
      int rc = launch(pub);
      if (rc != MOPT_OK) return rc;
      boundCommandBatch(c);
      rc = waitPublished(c, pub.sequence);
      collectDirectTiming(c, rc);
      return rc;
    }
  }
}

}  // namespace

namespace mopt_detail {

// ---- resident sweeps for the device-resident LM (lm.cpp) ---------------------------------------
namespace {
template <typename Args>
int uploadArgs(mopt_cost *c, const Args &host_value, hipStream_t s) {
  if (!c->d_lm_args) MOPT_HIP_TRY(deviceAlloc(&c->d_lm_args, 4096));  // >= every Args struct
  static_assert(sizeof(Args) <= 4096, "resident argument block too small");
  MOPT_HIP_TRY(mopt::launchStoreArgs<Args>(host_value, static_cast<Args *>(c->d_lm_args), s));
  return MOPT_OK;
}

template <typename S>
int residentPrepareP2P(mopt_cost *c, int jac_mode, bool moments, double *partials, hipStream_t s) {
  mopt::P2PSweepArgs<S> args;
  const S zero[kNumParams] = {0, 0, 0, 0, 0, 0};
  fillP2PArgs<S>(c, zero, false, args);  // data, loss, covariance; the step kernel writes T, 1/h
  args.partials = partials;
  int rc = uploadArgs(c, args, s);
  if (rc != MOPT_OK) return rc;
  if (moments) {
    mopt::AffineBasis basis;
    // analytic modes in the Euclidean parameters: the whole basis is independent of x; forward
    // differences and the left-perturbation form: the step kernel rewrites J per point and only the
    // covariance stays
    const bool per_point = jac_mode == MOPT_JAC_NUMERIC || jac_mode == MOPT_JAC_ANALYTIC_LEFT ||
                           jac_mode == MOPT_JAC_ANALYTIC_RIGHT;
    fillBasis<S>(c, per_point ? MOPT_JAC_ANALYTIC : jac_mode, args, basis);
    if (!c->d_lm_basis)
      MOPT_HIP_TRY(deviceAlloc(reinterpret_cast<void **>(&c->d_lm_basis), sizeof(mopt::AffineBasis)));
    MOPT_HIP_TRY(mopt::launchStoreArgs<mopt::AffineBasis>(basis, c->d_lm_basis, s));
  }
  return MOPT_OK;
}

// Rows of moments, except under MOPT_KERNEL_LITERAL.  Forward differences under AUTO / MOMENTS are the
// one case where that is not the whole answer: where some 0 < |x_j| < 0.08 the blocking calls evaluate
// literally (hasSmallForwardStep), and so does the device-resident loop, point by point — such a cost is
// swept by a kernel that holds both forms and the step kernel names the one that runs (residentPerIterate,
// sweep.hpp kLmGateMoments).  Literal forward differences at EVERY iterate were measured
// in round 5 at 21.5 against 17.9 us per evaluated point at 1 M, 101.6 against 84.4 at 10 M; the choice per
// point pays that only at the points that need it (and 0.6-2 us per point for the kernels that hold both
// forms: profiles/r6_device_loop_choice.txt).
bool usesMoments(const mopt_cost *c, int jac_mode) {
  (void)jac_mode;
  return c->variant != MOPT_KERNEL_LITERAL;
}
}  // namespace

bool residentPerIterate(const mopt_cost *c, int jac_mode) {
  static const bool enabled = envInt("MOPT_LM_PER_ITERATE", 1) != 0;  // 0: moments at every point (round 5)
  return enabled && c->model == kModelPoint2Point && jac_mode == MOPT_JAC_NUMERIC &&
         (c->variant == MOPT_KERNEL_AUTO || c->variant == MOPT_KERNEL_MOMENTS);
}

int residentGrid(const mopt_cost *c, int jac_mode) {
  switch (c->model) {
    case kModelPoint2Point:
      return gridFor(c, blocksPerCu(usesMoments(c, jac_mode) ? 1 : 2));
    case kModelReprojection:
      return gridFor(c, blocksPerCu(2));
    case kModelScalar:
      return scalarGrid(c);
    case kModelJit:
      return jitGrid(c);
    default:
      return 1;
  }
}

int residentDenseRow(const mopt_cost *c, int jac_mode) {
  const int n = c->n_params;
  switch (c->model) {
    case kModelPoint2Point:
      if (usesMoments(c, jac_mode)) return 0;  // rows of moments, contracted by their own finalize kernel
      return c->cov_mode == mopt::kCovGeneral ? mopt::kAccFull : mopt::kAccSym;
    case kModelReprojection:
      return c->cov_mode == mopt::kCovGeneral ? mopt::kAccFull : mopt::kAccSym;
    case kModelJit:
      if (c->jit.wide) return n * n + n + 1;  // the column-per-lane sweep has one row form
      return c->cov_mode == mopt::kCovGeneral ? n * n + n + 1 : n * (n + 1) / 2 + n + 1;
    case kModelScalar:
      return c->cov_mode == mopt::kCovGeneral ? n * n + n + 1 : n * (n + 1) / 2 + n + 1;
    default:
      return 0;
  }
}

int residentPrepare(mopt_cost *c, int jac_mode, hipStream_t s, mopt::LmCostDesc *desc,
                    double *partials_override) {
  double *const partials = partials_override ? partials_override : c->d_partials;
  if (jac_mode < MOPT_JAC_ANALYTIC || jac_mode > MOPT_JAC_ANALYTIC_RIGHT)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown jacobian_mode");
  desc->jac_mode = jac_mode;
  desc->n_out = c->n_out;
  desc->moments = 0;
  desc->result = c->d_result;
  int rc = MOPT_OK;
  const bool stale = c->lm_uploaded_version != c->state_version ||
                     c->lm_uploaded_mode != jac_mode || c->lm_uploaded_partials != partials ||
                     !c->d_lm_args;
  switch (c->model) {
    case kModelPoint2Point: {
      desc->model = mopt::kLmPoint2Point;
      desc->moments = usesMoments(c, jac_mode) ? 1 : 0;
      if (stale)
        rc = c->scalar_bytes == 8
                 ? residentPrepareP2P<double>(c, jac_mode, desc->moments, partials, s)
                 : residentPrepareP2P<float>(c, jac_mode, desc->moments, partials, s);
      break;
    }
    case kModelReprojection: {
      if (jac_mode != MOPT_JAC_NUMERIC)
        return fail(MOPT_ERR_UNSUPPORTED,
                    "the reprojection model has no analytic Jacobian (BaseModel, numeric only)");
      desc->model = mopt::kLmReprojection;
      std::memcpy(desc->camera, c->camera, sizeof desc->camera);
      std::memcpy(desc->frame, c->frame, sizeof desc->frame);
      if (stale) {
        mopt::ReprojSweepArgs args;
        const double zero[kNumParams] = {0, 0, 0, 0, 0, 0};
        fillReprojArgs(c, zero, false, args);
        args.partials = partials;
        rc = uploadArgs(c, args, s);
      }
      break;
    }
    case kModelScalar: {
      if (jac_mode == MOPT_JAC_ANALYTIC_TST_LAYOUT || jac_mode == MOPT_JAC_ANALYTIC_LEFT ||
        jac_mode == MOPT_JAC_ANALYTIC_RIGHT)
        return fail(MOPT_ERR_UNSUPPORTED,
                    "the as-written layout and the perturbation Jacobians apply to point2point only");
      if (jac_mode == MOPT_JAC_ANALYTIC && !scalarModelHasJacobian(c->scalar_model))
        return fail(MOPT_ERR_UNSUPPORTED,
                    "Non implemented non-jacobian model function `f_df` being used.");
      desc->model = mopt::kLmScalar;
      desc->x_offset = c->scalar_bytes == 8 ? int(offsetof(mopt::ScalarSweepArgs<double>, x))
                                            : int(offsetof(mopt::ScalarSweepArgs<float>, x));
      if (stale) {
        if (c->scalar_bytes == 8) {
          mopt::ScalarSweepArgs<double> args;
          const double zero[mopt::kMaxParams] = {0};
          fillScalarArgs<double>(c, zero, args);
          args.partials = partials;
          rc = uploadArgs(c, args, s);
        } else {
          mopt::ScalarSweepArgs<float> args;
          const float zero[mopt::kMaxParams] = {0};
          fillScalarArgs<float>(c, zero, args);
          args.partials = partials;
          rc = uploadArgs(c, args, s);
        }
      }
      break;
    }
    case kModelJit: {
      if (jac_mode == MOPT_JAC_ANALYTIC_TST_LAYOUT || jac_mode == MOPT_JAC_ANALYTIC_LEFT ||
        jac_mode == MOPT_JAC_ANALYTIC_RIGHT)
        return fail(MOPT_ERR_UNSUPPORTED,
                    "the as-written layout and the perturbation Jacobians apply to point2point only");
      if (jac_mode == MOPT_JAC_ANALYTIC && !c->jit.has_jacobian)
        return fail(MOPT_ERR_UNSUPPORTED,
                    "Non implemented non-jacobian model function `f_df` being used.");
      // compile (first use) before anything is queued: a source error must surface here
      if (!mopt::jitVariant(c->jit, jac_mode == MOPT_JAC_NUMERIC ? 2 : 1, c->cov_mode))
        return fail(MOPT_ERR_INVALID_ARGUMENT, mopt::jitLastError());
      desc->model = mopt::kLmJit;
      if (c->jit.wide) {  // x[16] | h[16] in the wide sweep's argument block
        desc->x_slots = mopt::kMaxWideParams;
        desc->x_offset = c->scalar_bytes == 8 ? int(offsetof(mopt::JitWideArgs<double>, x))
                                              : int(offsetof(mopt::JitWideArgs<float>, x));
        if (stale) {
          if (c->scalar_bytes == 8) {
            mopt::JitWideArgs<double> args;
            const double zero[mopt::kMaxWideParams] = {0};
            fillJitWideArgs<double>(c, zero, args);
            args.partials = partials;
            rc = uploadArgs(c, args, s);
          } else {
            mopt::JitWideArgs<float> args;
            const float zero[mopt::kMaxWideParams] = {0};
            fillJitWideArgs<float>(c, zero, args);
            args.partials = partials;
            rc = uploadArgs(c, args, s);
          }
        }
        break;
      }
      desc->x_offset = c->scalar_bytes == 8 ? int(offsetof(mopt::JitArgs<double>, x))
                                            : int(offsetof(mopt::JitArgs<float>, x));
      if (stale) {
        if (c->scalar_bytes == 8) {
          mopt::JitArgs<double> args;
          const double zero[mopt::kMaxParams] = {0};
          fillJitArgs<double>(c, zero, args);
          args.partials = partials;
          rc = uploadArgs(c, args, s);
        } else {
          mopt::JitArgs<float> args;
          const float zero[mopt::kMaxParams] = {0};
          fillJitArgs<float>(c, zero, args);
          args.partials = partials;
          rc = uploadArgs(c, args, s);
        }
      }
      break;
    }
    default:
      return fail(MOPT_ERR_UNSUPPORTED, "mopt_lm_minimize: unknown model kind");
  }
  if (rc != MOPT_OK) return rc;
  c->lm_uploaded_version = c->state_version;
  c->lm_uploaded_mode = jac_mode;
  c->lm_uploaded_partials = partials;
  desc->args = c->d_lm_args;
  desc->basis = c->d_lm_basis;
  return MOPT_OK;
}

int residentFinalizeMerged(mopt_cost *last, int rows, int row_length, mopt::LmControl *control,
                           hipStream_t s, const mopt::LmProblem *step, int own_index) {
  MOPT_HIP_TRY(mopt::launchFinalizeDenseResident(last->d_partials, rows, row_length, last->n_params,
                                                 last->d_result, control, s, nullptr, step,
                                                 own_index, last->scalar_bytes));
  return MOPT_OK;
}

// Can the resident sweeps of these costs (whose rows already lie behind one another: `merged` in
// lm.cpp) go out as ONE launch?  They must run the same kernel: same model kind, scalar type,
// Jacobian mode and covariance form — and for run-time compiled models the same source.
bool residentSetSupported(mopt_cost *const *costs, int num_costs, const int *jac_modes) {
  if (num_costs < 2) return false;
  const mopt_cost *first = costs[0];
  for (int k = 0; k < num_costs; ++k) {
    const mopt_cost *c = costs[k];
    if (c->model != first->model || c->cov_mode != first->cov_mode ||
        c->scalar_bytes != first->scalar_bytes || jac_modes[k] != jac_modes[0] || c->matcher)
      return false;
    switch (c->model) {
      case kModelReprojection:
        break;
      case kModelPoint2Point:
        if (usesMoments(c, jac_modes[k])) return false;  // rows of moments: one finalize per cost anyway
        break;
      case kModelScalar:
        if (c->scalar_model != first->scalar_model) return false;
        break;
      case kModelJit:
        // the shape reaches the compiler as -D options, not as source text: two costs with the same
        // bodies and another m, plane or aux count are different kernels
        if (c->jit.wide || c->jit.source != first->jit.source ||
            c->jit.n_params != first->jit.n_params || c->jit.n_outputs != first->jit.n_outputs ||
            c->jit.n_planes != first->jit.n_planes || c->jit.n_aux != first->jit.n_aux)
          return false;
        break;
      default:
        return false;
    }
  }
  return true;
}

int residentSweepSet(mopt_cost *const *costs, int num_costs, const int *jac_modes,
                     const int *first_row, mopt::LmControl *control, hipStream_t s) {
  mopt::ResidentSweepSet set;
  set.num_costs = num_costs;
  size_t bytes = 0;
  for (int k = 0; k < num_costs; ++k) {
    set.args[k] = costs[k]->d_lm_args;
    set.first_block[k] = first_row[k];
    // bytes the launch streams: 40 per reprojection element, 6 scalars per correspondence
    bytes += costs[k]->model == kModelReprojection
                 ? size_t(costs[k]->count) * 40
                 : size_t(costs[k]->count) * 6 * size_t(costs[k]->scalar_bytes);
  }
  set.first_block[num_costs] = first_row[num_costs];
  mopt::LaunchSite site;
  site.stream = s;
  site.streaming = bytes > (size_t(32) << 20);
  mopt_cost *first = costs[0];
  const int jac_mode = jac_modes[0];
  switch (first->model) {
    case kModelReprojection:
      MOPT_HIP_TRY(mopt::launchReprojResidentSet(set, control, first->cov_mode, site));
      return MOPT_OK;
    case kModelPoint2Point:
      if (first->scalar_bytes == 8)
        MOPT_HIP_TRY(mopt::launchP2PLiteralResidentSet<double>(set, control, jac_mode,
                                                               first->cov_mode, site));
      else
        MOPT_HIP_TRY(mopt::launchP2PLiteralResidentSet<float>(set, control, jac_mode,
                                                              first->cov_mode, site));
      return MOPT_OK;
    case kModelScalar:
      if (first->scalar_bytes == 8)
        MOPT_HIP_TRY(mopt::launchScalarModelResidentSet<double>(set, control, first->scalar_model,
                                                                jac_mode, first->cov_mode, s));
      else
        MOPT_HIP_TRY(mopt::launchScalarModelResidentSet<float>(set, control, first->scalar_model,
                                                               jac_mode, first->cov_mode, s));
      return MOPT_OK;
    case kModelJit: {
      const mopt::JitVariant *variant =
          mopt::jitVariant(first->jit, jac_mode == MOPT_JAC_NUMERIC ? 2 : 1, first->cov_mode);
      if (!variant) return fail(MOPT_ERR_INVALID_ARGUMENT, mopt::jitLastError());
      MOPT_HIP_TRY(mopt::jitLaunchResidentSet(*variant, set, control, s));
      return MOPT_OK;
    }
    default:
      return fail(MOPT_ERR_UNSUPPORTED, "no one-launch resident sweep for this model");
  }
}

int residentSweep(mopt_cost *c, int jac_mode, mopt::LmControl *control, hipStream_t s,
                  unsigned long long base_sequence, const mopt::LmProblem *step, int own_index,
                  bool finalize) {
  mopt::PeerCombine pc;
  const mopt::PeerCombine *peers = nullptr;
  if (c->combine.mode == MOPT_COMBINE_PEER) {
    const ShardCombine &sc = c->combine;
    for (int k = 0; k < sc.num_ranks; ++k) pc.blocks[k] = sc.peer_blocks[k];
    pc.rank = sc.rank;
    pc.num_ranks = sc.num_ranks;
    pc.offset = 0;
    pc.sequence = base_sequence;  // + control->trial, added on the device
    pc.timeout_ticks = sc.peer_timeout_ticks;
    peers = &pc;
  }
  const int n = c->n_params;
  mopt::LaunchSite site;
  site.stream = s;
  const size_t bytes = size_t(c->count) * 6 * size_t(c->scalar_bytes);
  site.streaming = bytes > (size_t(32) << 20);
  if (c->matcher) {
    // the model's update(x): re-search the correspondences at the point the step kernel has just
    // proposed, when it says so (a linearization point), before that point is swept
    if (c->scalar_bytes == 8) {
      mopt::IcpMatchArgs<double> a;
      fillIcpArgs<double>(c, a);
      MOPT_HIP_TRY(mopt::launchIcpMatchResident<double>(
          a, static_cast<const mopt::P2PSweepArgs<double> *>(c->d_lm_args), control, s));
    } else {
      mopt::IcpMatchArgs<float> a;
      fillIcpArgs<float>(c, a);
      MOPT_HIP_TRY(mopt::launchIcpMatchResident<float>(
          a, static_cast<const mopt::P2PSweepArgs<float> *>(c->d_lm_args), control, s));
    }
  }
  switch (c->model) {
    case kModelPoint2Point: {
      if (residentPerIterate(c, jac_mode)) {
        // one sweep launch that holds both forward-difference forms and runs the one the step kernel named
        // for this point (sweep.hpp kLmGateMoments), and one finalize kernel that reads which it was
        const int grid_l = gridFor(c, blocksPerCu(2));
        // the moments form over the first half of the launch's workgroups (one per CU, as the single-purpose
        // moments sweep runs; the rest leave at once) or over all of them: MOPT_LM_EITHER_MOMENTS_BLOCKS = 1 /
        // 2 — measured 82.8-83.5 against 84.8-85.7 us per sweep at 10 M, 18.7 against 21.2 at 1 M
        static const int moments_blocks = envInt("MOPT_LM_EITHER_MOMENTS_BLOCKS", 1);
        const int grid_m = moments_blocks >= 2 ? grid_l : std::min(grid_l, residentGrid(c, jac_mode));
        const int nacc = c->cov_mode == mopt::kCovGeneral ? mopt::kAccFull : mopt::kAccSym;
        if (c->scalar_bytes == 8)
          MOPT_HIP_TRY(mopt::launchForwardDiffEitherResident<double>(
              static_cast<const double *>(c->d_tiles), c->num_tiles,
              static_cast<const mopt::P2PSweepArgs<double> *>(c->d_lm_args), control, c->cov_mode, grid_l,
              grid_m, site));
        else
          MOPT_HIP_TRY(mopt::launchForwardDiffEitherResident<float>(
              static_cast<const float *>(c->d_tiles), c->num_tiles,
              static_cast<const mopt::P2PSweepArgs<float> *>(c->d_lm_args), control, c->cov_mode, grid_l,
              grid_m, site));
        if (finalize)
          MOPT_HIP_TRY(mopt::launchFinalizeEitherResident(c->d_partials, grid_m, grid_l, nacc, c->d_lm_basis,
                                                          c->d_result, control, s, peers, step, own_index,
                                                          c->scalar_bytes));
        return MOPT_OK;
      }
      if (usesMoments(c, jac_mode)) {
        const int grid = residentGrid(c, jac_mode);
        if (c->scalar_bytes == 8)
          MOPT_HIP_TRY(mopt::launchP2PMomentsResident<double>(
              static_cast<const double *>(c->d_tiles), c->num_tiles,
              static_cast<const mopt::P2PSweepArgs<double> *>(c->d_lm_args), control, grid, site));
        else
          MOPT_HIP_TRY(mopt::launchP2PMomentsResident<float>(
              static_cast<const float *>(c->d_tiles), c->num_tiles,
              static_cast<const mopt::P2PSweepArgs<float> *>(c->d_lm_args), control, grid, site));
        if (finalize)
          MOPT_HIP_TRY(mopt::launchFinalizeMomentsResident(c->d_partials, grid, c->d_lm_basis,
                                                         c->d_result, control, s, peers, step,
                                                         own_index, c->scalar_bytes));
      } else {
        const int grid = residentGrid(c, jac_mode);
        const int nacc = residentDenseRow(c, jac_mode);
        if (c->scalar_bytes == 8)
          MOPT_HIP_TRY(mopt::launchP2PLiteralResident<double>(
              static_cast<const mopt::P2PSweepArgs<double> *>(c->d_lm_args), control, jac_mode,
              c->cov_mode, grid, site));
        else
          MOPT_HIP_TRY(mopt::launchP2PLiteralResident<float>(
              static_cast<const mopt::P2PSweepArgs<float> *>(c->d_lm_args), control, jac_mode,
              c->cov_mode, grid, site));
        if (finalize)
          MOPT_HIP_TRY(mopt::launchFinalizeDenseResident(c->d_partials, grid, nacc, kNumParams,
                                                         c->d_result, control, s, peers, step,
                                                         own_index, c->scalar_bytes));
      }
      return MOPT_OK;
    }
    case kModelReprojection: {
      const int grid = residentGrid(c, jac_mode);
      const int nacc = residentDenseRow(c, jac_mode);
      MOPT_HIP_TRY(mopt::launchReprojResident(
          static_cast<const mopt::ReprojSweepArgs *>(c->d_lm_args), control, c->cov_mode, grid, site));
      if (finalize)
        MOPT_HIP_TRY(mopt::launchFinalizeDenseResident(c->d_partials, grid, nacc, kNumParams,
                                                     c->d_result, control, s, peers, step, own_index,
                                                     c->scalar_bytes));
      return MOPT_OK;
    }
    case kModelScalar: {
      const int grid = residentGrid(c, jac_mode);
      const int nacc = residentDenseRow(c, jac_mode);
      if (c->scalar_bytes == 8)
        MOPT_HIP_TRY(mopt::launchScalarModelResident<double>(
            static_cast<const mopt::ScalarSweepArgs<double> *>(c->d_lm_args), control,
            c->scalar_model, jac_mode, c->cov_mode, grid, s));
      else
        MOPT_HIP_TRY(mopt::launchScalarModelResident<float>(
            static_cast<const mopt::ScalarSweepArgs<float> *>(c->d_lm_args), control,
            c->scalar_model, jac_mode, c->cov_mode, grid, s));
      if (finalize)
        MOPT_HIP_TRY(mopt::launchFinalizeDenseResident(c->d_partials, grid, nacc, n, c->d_result,
                                                     control, s, peers, step, own_index,
                                                     c->scalar_bytes));
      return MOPT_OK;
    }
    case kModelJit: {
      const mopt::JitVariant *variant =
          mopt::jitVariant(c->jit, jac_mode == MOPT_JAC_NUMERIC ? 2 : 1, c->cov_mode);
      if (!variant) return fail(MOPT_ERR_INVALID_ARGUMENT, mopt::jitLastError());
      const int grid = residentGrid(c, jac_mode);
      const int nacc = residentDenseRow(c, jac_mode);
      MOPT_HIP_TRY(mopt::jitLaunchResident(*variant, c->d_lm_args, control, grid, s));
      if (finalize)
        MOPT_HIP_TRY(mopt::launchFinalizeDenseResident(c->d_partials, grid, nacc, n, c->d_result,
                                                     control, s, peers, step, own_index,
                                                     c->scalar_bytes));
      return MOPT_OK;
    }
    default:
      return fail(MOPT_ERR_UNSUPPORTED, "no resident sweep for this model");
  }
}

void releaseResident(mopt_cost *c) {
  deviceRelease(c->d_lm_args);
  deviceRelease(c->d_lm_basis);
  deviceRelease(c->d_lm_control);
  deviceRelease(c->d_lm_state);
  c->d_lm_args = nullptr;
  c->d_lm_basis = nullptr;
  c->d_lm_control = nullptr;
  c->d_lm_state = nullptr;
  if (c->h_lm_report) (void)hipHostFree(c->h_lm_report);
  c->h_lm_report = c->h_lm_report_dev = nullptr;
}

}  // namespace mopt_detail

namespace mopt_detail {
// For the single-process device group (group.cpp): a shard's sweep + finalize published into its
// own mapped host memory, launched by one thread and awaited by another.
int launchPublishedSweep(mopt_cost *c, bool cost_only, int jac_mode, const void *x,
                         unsigned long long *sequence_out) {
  const int offset = cost_only ? costOffset(c) : 0;
  const mopt::HostPublish pub = nextPublish(c, offset);
  c->stat_sweeps += 1;
  *sequence_out = pub.sequence;
  chooseDispatchPath(c);
  c->aql_timed = false;
  const int rc = cost_only ? costAsyncImpl(c, x, c->d_result + offset, c->stream, pub)
                           : linearizeAsyncImpl(c, jac_mode, x, c->d_result, c->stream, pub);
  settleDispatchPath(c);
  if (rc == MOPT_OK) boundCommandBatch(c);
  return rc;
}
int waitPublishedSweep(mopt_cost *c, unsigned long long sequence) {
  const int rc = waitPublished(c, sequence);
  collectDirectTiming(c, rc);
  return rc;
}

mopt::HostPublish nextHostPublish(mopt_cost *c, int offset) { return nextPublish(c, offset); }
int waitHostPublished(mopt_cost *c, unsigned long long sequence) { return waitPublished(c, sequence); }
}  // namespace mopt_detail

namespace mopt_detail {
namespace {
std::mutex g_stream_pool_mutex;
std::map<int, std::vector<hipStream_t>> g_stream_pool;
}  // namespace

hipError_t acquireStream(int device, hipStream_t *out) {
  {
    std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
    auto &pool = g_stream_pool[device];
    if (!pool.empty()) {
      *out = pool.back();
      pool.pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);  // current device == `device`
}

void releaseStream(int device, hipStream_t stream) {
  if (!stream) return;
  std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
  g_stream_pool[device].push_back(stream);
}

void storeResult(const mopt_cost *c, const double *res, void *hessian, void *b, void *sum_sq) {
  const int n = c->n_params, nn = n * n;
  if (c->scalar_bytes == 8) {
    if (hessian) std::memcpy(hessian, res, nn * sizeof(double));
    if (b) std::memcpy(b, res + nn, n * sizeof(double));
    if (sum_sq) *static_cast<double *>(sum_sq) = res[nn + n];
  } else {
    if (hessian)
      for (int k = 0; k < nn; ++k) static_cast<float *>(hessian)[k] = float(res[k]);
    if (b)
      for (int k = 0; k < n; ++k) static_cast<float *>(b)[k] = float(res[nn + k]);
    if (sum_sq) *static_cast<float *>(sum_sq) = float(res[nn + n]);
  }
}

int commonCreate(mopt_cost *c, int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MOPT_ERR_NO_DEVICE, "no HIP device is visible to this process");
  if (device < 0 || device >= ndev)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "device index out of range");
  c->device = device;
  MOPT_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  MOPT_HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount;
  c->max_grid = c->num_cus * 16;
  MOPT_HIP_TRY(acquireStream(device, &c->stream));
  MOPT_HIP_TRY(deviceAlloc(reinterpret_cast<void **>(&c->d_partials),
                         size_t(c->max_grid) * kPartialRowSlots * sizeof(double)));
  MOPT_HIP_TRY(deviceAlloc(reinterpret_cast<void **>(&c->d_result), kResultSlots * sizeof(double)));
  // results (43) + padding + flag word in one mapped, coherent host allocation
  MOPT_HIP_TRY(mappedHostAlloc(reinterpret_cast<void **>(&c->h_result),
                               (kResultSlots + 16) * sizeof(double)));
  std::memset(c->h_result, 0, (kResultSlots + 16) * sizeof(double));
  c->h_flag = reinterpret_cast<unsigned long long *>(c->h_result + kResultSlots);
  MOPT_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_result_dev), c->h_result, 0));
  c->h_flag_dev = reinterpret_cast<unsigned long long *>(c->h_result_dev + kResultSlots);
  // counted as a user of the device's direct-dispatch queues (aql.hpp; created when a cost first takes that path)
  c->aql_retained = mopt_detail::aqlRetain(device);
  return MOPT_OK;
}

hipError_t quiesceCost(mopt_cost *c) {
  hipError_t first = hipSuccess;
  if (c->aql_touched && c->aql_queue) {  // the finalize kernel of the last direct sweep may still be retiring
    (void)mopt_detail::aqlDrain(c->aql_queue);
    c->aql_touched = false;
  }
  if (c->aql_queue) mopt_detail::aqlForgetStamp(c->aql_queue, c);  // (a timed dispatch nobody collected)
  c->aql_timed = false;
  if (c->stream) first = hipStreamSynchronize(c->stream);
  c->hip_pending = false;
  c->own_async_pending = false;
  if (c->foreign_pending && c->foreign_done) {
    const hipError_t e = hipEventSynchronize(c->foreign_done);
    if (first == hipSuccess) first = e;
    c->foreign_pending = false;
  }
  return first;
}

void destroyCost(mopt_cost *c) {
  if (!c) return;
  for (mopt_cost *s : c->siblings)
    s->siblings.erase(std::remove(s->siblings.begin(), s->siblings.end(), c), s->siblings.end());
  c->siblings.clear();
  (void)hipSetDevice(c->device);
  (void)quiesceCost(c);
  if (c->foreign_done) (void)hipEventDestroy(c->foreign_done);
  for (auto &pr : c->pending_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  if (c->comm) ncclCommDestroy(c->comm);
  c->comm = nullptr;
  releaseCombine(c);
  if (c->matcher) {
    deviceRelease(c->matcher->d_sorted);
    deviceRelease(c->matcher->d_cell_start);
    deviceRelease(c->matcher->d_matched);
    deviceRelease(c->matcher->d_order);
  }
  mopt::jitRelease(c->jit);
  releaseResident(c);
  deviceRelease(c->d_tiles);
  deviceRelease(c->d_partials);
  deviceRelease(c->d_result);
  mappedHostRelease(c->device, c->h_result, (kResultSlots + 16) * sizeof(double));
  releaseStream(c->device, c->stream);  // synchronised at the top of this function
  if (c->aql_retained) mopt_detail::aqlRelease(c->device);
  delete c;
}

}  // namespace mopt_detail

namespace {
constexpr size_t kBounceBytes = size_t(8) << 20;
std::mutex g_bounce_mutex;
void *g_bounce = nullptr;  // pinned, allocated on first use, kept for the life of the process

// Copies (or adopts) two input arrays into device staging memory and returns device pointers.
struct Staging {
  void *a = nullptr, *b = nullptr;
  void *block = nullptr;  // one allocation behind both copies
  hipStream_t stream = nullptr;
  ~Staging() {
    if (!block) return;
    (void)hipStreamSynchronize(stream);  // an early error return may leave a copy in flight
    deviceRelease(block);
  }
};

int stageInputs(const void *ha, size_t bytes_a, const void *hb, size_t bytes_b, unsigned flags,
                hipStream_t s, Staging &st) {
  if (flags & MOPT_INPUT_DEVICE) {
    st.a = const_cast<void *>(ha);
    st.b = const_cast<void *>(hb);
    return MOPT_OK;
  }
  const size_t offset_b = (bytes_a + 255) & ~size_t(255);
  if (bytes_a + bytes_b == 0) return MOPT_OK;
  st.stream = s;
  MOPT_HIP_TRY(deviceAlloc(&st.block, offset_b + bytes_b));
  st.a = st.block;
  st.b = static_cast<char *>(st.block) + offset_b;
  // Small inputs go through a pinned bounce buffer owned by the library: a copy straight from
  // pageable memory makes the runtime pin the caller's pages first, which costs 4-8 ms for an
  // address range it has not seen before, whatever its size (measured: a 30 k-point cost took
  // 8 ms to construct, 1.4 MB of input).  Large inputs amortise that and skip the extra host copy.
  if (offset_b + bytes_b <= kBounceBytes) {
    std::lock_guard<std::mutex> lock(g_bounce_mutex);
    if (!g_bounce) MOPT_HIP_TRY(hipHostMalloc(&g_bounce, kBounceBytes, hipHostMallocPortable));
    if (bytes_a) std::memcpy(g_bounce, ha, bytes_a);
    if (bytes_b) std::memcpy(static_cast<char *>(g_bounce) + offset_b, hb, bytes_b);
    MOPT_HIP_TRY(hipMemcpyAsync(st.block, g_bounce, offset_b + bytes_b, hipMemcpyHostToDevice, s));
    MOPT_HIP_TRY(hipStreamSynchronize(s));  // the bounce buffer is free again
    return MOPT_OK;
  }
  if (bytes_a) MOPT_HIP_TRY(hipMemcpyAsync(st.a, ha, bytes_a, hipMemcpyHostToDevice, s));
  if (bytes_b) MOPT_HIP_TRY(hipMemcpyAsync(st.b, hb, bytes_b, hipMemcpyHostToDevice, s));
  return MOPT_OK;
}

}  // namespace

namespace {
template <typename S>
void se3FromParams(const S *x, S *T, S *T_plus, S *h_out) {
  moptimizer::so3::convert6DOFParameterToMatrix<S>(x, T);
  if (!T_plus && !h_out) return;
  S h[kNumParams], xp[kNumParams][kNumParams];
  forwardSteps<S>(x, h, xp);
  for (int j = 0; j < kNumParams; ++j) {
    if (h_out) h_out[j] = h[j];
    if (T_plus) moptimizer::so3::convert6DOFParameterToMatrix<S>(xp[j], T_plus + 16 * j);
  }
}
}  // namespace

extern "C" {

const char *mopt_last_error(void) { return g_last_error.c_str(); }
const char *mopt_version(void) { return "moptimizer_hip 0.1 (gfx950)"; }

int mopt_device_count(int *count) {
  if (!count) return fail(MOPT_ERR_INVALID_ARGUMENT, "count is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return MOPT_OK;
}

int mopt_se3_from_params(int scalar_bytes, const void *x, void *T_out, void *T_plus_out,
                         void *h_out) {
  if (!x || !T_out) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (scalar_bytes == 8)
    se3FromParams<double>(static_cast<const double *>(x), static_cast<double *>(T_out),
                          static_cast<double *>(T_plus_out), static_cast<double *>(h_out));
  else if (scalar_bytes == 4)
    se3FromParams<float>(static_cast<const float *>(x), static_cast<float *>(T_out),
                         static_cast<float *>(T_plus_out), static_cast<float *>(h_out));
  else
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 or 8");
  return MOPT_OK;
}

int mopt_se3_plus(int scalar_bytes, const void *x, const void *delta, void *x_out) {
  if (!x || !delta || !x_out) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (scalar_bytes == 8)
    moptimizer::so3::se3Plus<double>(static_cast<const double *>(x),
                                     static_cast<const double *>(delta), static_cast<double *>(x_out));
  else if (scalar_bytes == 4)
    moptimizer::so3::se3Plus<float>(static_cast<const float *>(x), static_cast<const float *>(delta),
                                    static_cast<float *>(x_out));
  else
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 or 8");
  return MOPT_OK;
}

int mopt_se3_plus_right(int scalar_bytes, const void *x, const void *delta, void *x_out) {
  if (!x || !delta || !x_out) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (scalar_bytes == 8)
    moptimizer::so3::se3PlusRight<double>(static_cast<const double *>(x),
                                          static_cast<const double *>(delta),
                                          static_cast<double *>(x_out));
  else if (scalar_bytes == 4)
    moptimizer::so3::se3PlusRight<float>(static_cast<const float *>(x),
                                         static_cast<const float *>(delta), static_cast<float *>(x_out));
  else
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 or 8");
  return MOPT_OK;
}

int mopt_point2point_create(mopt_cost **out, int device, int scalar_bytes, const void *src_xyz,
                            const void *tgt_xyz, int64_t count, unsigned flags) {
  if (!out) return fail(MOPT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (scalar_bytes != 4 && scalar_bytes != 8)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 or 8");
  if (count < 0 || (count > 0 && (!src_xyz || !tgt_xyz)))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad point arrays / count");
  std::unique_ptr<mopt_cost, void (*)(mopt_cost *)> c(new (std::nothrow) mopt_cost, destroyCost);
  if (!c) return fail(MOPT_ERR_HIP, "out of host memory");
  c->scalar_bytes = scalar_bytes;
  c->model = kModelPoint2Point;
  c->n_out = 3;
  c->count = count;
  const int tile_points =
      scalar_bytes == 8 ? mopt::TileShape<double>::kPoints : mopt::TileShape<float>::kPoints;
  const long long tiles = (count + tile_points - 1) / tile_points;
  if (tiles > std::numeric_limits<int>::max())
    return fail(MOPT_ERR_INVALID_ARGUMENT, "count too large");
  c->num_tiles = int(tiles);
  int rc = commonCreate(c.get(), device);
  if (rc != MOPT_OK) return rc;
  if (!mopt_detail::g_creating_for_search) mopt_detail::aqlWarm(device);

  rc = mopt_point2point_set_data(c.get(), src_xyz, tgt_xyz, count, flags);
  if (rc != MOPT_OK) return rc;
  c->state_version = 0;
  *out = c.release();
  return MOPT_OK;
}

int mopt_point2point_set_data(mopt_cost *c, const void *src_xyz, const void *tgt_xyz,
                              int64_t count, unsigned flags) {
  if (!c || c->model != kModelPoint2Point)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "not a point2point cost");
  if (c->matcher)
    return fail(MOPT_ERR_UNSUPPORTED,
                "a cost made by mopt_icp_create owns its correspondences (mopt_icp_update)");
  if (count < 0 || (count > 0 && (!src_xyz || !tgt_xyz)))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad point arrays / count");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  const int tile_points = c->scalar_bytes == 8 ? mopt::TileShape<double>::kPoints
                                               : mopt::TileShape<float>::kPoints;
  const long long tiles = (count + tile_points - 1) / tile_points;
  if (tiles > std::numeric_limits<int>::max())
    return fail(MOPT_ERR_INVALID_ARGUMENT, "count too large");
  const size_t tile_bytes = size_t(tile_points) * 6 * c->scalar_bytes;
  MOPT_HIP_TRY(quiesceCost(c));
  if (tiles > c->capacity_tiles) {
    deviceRelease(c->d_tiles);  // nothing enqueued for this cost is still running
    c->d_tiles = nullptr;
    c->capacity_tiles = 0;
    MOPT_HIP_TRY(deviceAlloc(&c->d_tiles, tile_bytes * size_t(tiles)));
    c->capacity_tiles = tiles;
  }
  c->count = count;
  c->num_tiles = int(tiles);
  c->cache.valid = false;
  c->state_version += 1;
  if (tiles > 0) {
    Staging st;
    const size_t bytes = size_t(count) * 3 * c->scalar_bytes;
    const int rc = stageInputs(src_xyz, bytes, tgt_xyz, bytes, flags, c->stream, st);
    if (rc != MOPT_OK) return rc;
    if (c->scalar_bytes == 8)
      MOPT_HIP_TRY(mopt::launchRelayoutP2P<double>(
          static_cast<const double *>(st.a), static_cast<const double *>(st.b), count,
          static_cast<double *>(c->d_tiles), c->num_tiles, c->stream));
    else
      MOPT_HIP_TRY(mopt::launchRelayoutP2P<float>(
          static_cast<const float *>(st.a), static_cast<const float *>(st.b), count,
          static_cast<float *>(c->d_tiles), c->num_tiles, c->stream));
    MOPT_HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return MOPT_OK;
}


int mopt_reprojection_create(mopt_cost **out, int device, const double *points_xyzw,
                             const int32_t *pixels_uv, int64_t count, const double *camera_3x4,
                             const double *frame_4x4, unsigned flags) {
  if (!out) return fail(MOPT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (count <= 0 || !points_xyzw || !pixels_uv)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "Empty point / pixel list");
  std::unique_ptr<mopt_cost, void (*)(mopt_cost *)> c(new (std::nothrow) mopt_cost, destroyCost);
  if (!c) return fail(MOPT_ERR_HIP, "out of host memory");
  c->scalar_bytes = 8;
  c->model = kModelReprojection;
  c->n_out = 2;
  c->count = count;
  defaultReprojConstants<double>(c->camera, c->frame);
  if (camera_3x4) std::memcpy(c->camera, camera_3x4, sizeof c->camera);
  if (frame_4x4) std::memcpy(c->frame, frame_4x4, sizeof c->frame);
  const long long tiles = (count + mopt::kReprojTilePoints - 1) / mopt::kReprojTilePoints;
  if (tiles > std::numeric_limits<int>::max())
    return fail(MOPT_ERR_INVALID_ARGUMENT, "count too large");
  c->num_tiles = int(tiles);
  int rc = commonCreate(c.get(), device);
  if (rc != MOPT_OK) return rc;
  mopt_detail::aqlWarm(device);
  MOPT_HIP_TRY(deviceAlloc(&c->d_tiles, size_t(mopt::kReprojTileBytes) * c->num_tiles));
  Staging st;
  rc = stageInputs(points_xyzw, size_t(count) * 32, pixels_uv, size_t(count) * 8, flags, c->stream,
                   st);
  if (rc != MOPT_OK) return rc;
  MOPT_HIP_TRY(mopt::launchRelayoutReproj(static_cast<const double *>(st.a),
                                          static_cast<const int32_t *>(st.b), count,
                                          static_cast<unsigned char *>(c->d_tiles), c->num_tiles,
                                          c->stream));
  MOPT_HIP_TRY(hipStreamSynchronize(c->stream));
  *out = c.release();
  return MOPT_OK;
}

int mopt_scalar_model_create(mopt_cost **out, int device, int scalar_bytes, int model_kind,
                             const void *t, const void *y, int64_t stride_scalars, int64_t count) {
  if (!out) return fail(MOPT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (scalar_bytes != 4 && scalar_bytes != 8)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 or 8");
  int n = 0, m = 0, planes = 0;
  switch (model_kind) {
    case MOPT_MODEL_EXP_CURVE: n = 2; m = 1; planes = 2; break;
    case MOPT_MODEL_RATIONAL: n = 2; m = 1; planes = 2; break;
    case MOPT_MODEL_POWELL: n = 4; m = 4; planes = 0; break;
    default: return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown scalar model kind");
  }
  if (count < 1 || (planes > 0 && (!t || !y || stride_scalars < 1)))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad data arrays / count");
  std::unique_ptr<mopt_cost, void (*)(mopt_cost *)> c(new (std::nothrow) mopt_cost, destroyCost);
  if (!c) return fail(MOPT_ERR_HIP, "out of host memory");
  c->scalar_bytes = scalar_bytes;
  c->model = kModelScalar;
  c->scalar_model = model_kind;
  c->n_params = n;
  c->n_out = m;
  c->count = count;
  c->num_tiles = 1;
  // plane stride padded to whole 16-byte packs (the sweep loads 16 bytes per lane and plane); the
  // tail of the last pack is zero-filled and enters no sum
  const int vec = 16 / scalar_bytes;
  const int64_t padded = (count + vec - 1) / vec * vec;
  c->data_stride = padded;
  int rc = commonCreate(c.get(), device);
  if (rc != MOPT_OK) return rc;
  mopt_detail::aqlWarm(device);
  if (planes > 0) {
    // gather the (possibly interleaved) host arrays into contiguous planes t | y
    std::vector<unsigned char> staged(size_t(planes) * size_t(padded) * scalar_bytes, 0);
    const unsigned char *src[2] = {static_cast<const unsigned char *>(t),
                                   static_cast<const unsigned char *>(y)};
    for (int p = 0; p < planes; ++p)
      for (int64_t i = 0; i < count; ++i)
        std::memcpy(&staged[(size_t(p) * padded + size_t(i)) * scalar_bytes],
                    src[p] + size_t(i) * size_t(stride_scalars) * scalar_bytes, scalar_bytes);
    // an observation whose y is NaN is "not a residual" (f / f_df returning false: model.h:32,43):
    // data that carry the marker run the sweeps that look for it (sweep.hpp, ScalarModelKind)
    bool marked = false;
    for (int64_t i = 0; i < count && !marked; ++i) {
      const unsigned char *yi = &staged[(size_t(1) * padded + size_t(i)) * scalar_bytes];
      if (scalar_bytes == 8) {
        double v;
        std::memcpy(&v, yi, 8);
        marked = v != v;
      } else {
        float v;
        std::memcpy(&v, yi, 4);
        marked = v != v;
      }
    }
    if (marked)
      c->scalar_model = model_kind == MOPT_MODEL_EXP_CURVE ? int(mopt::kScalarExpCurveMarked)
                                                            : int(mopt::kScalarRationalMarked);
    MOPT_HIP_TRY(deviceAlloc(&c->d_tiles, staged.size()));
    MOPT_HIP_TRY(hipMemcpy(c->d_tiles, staged.data(), staged.size(), hipMemcpyHostToDevice));
  }
  mopt_cost_set_covariance(c.get(), nullptr);
  c->state_version = 0;
  *out = c.release();
  return MOPT_OK;
}

int mopt_jit_model_create(mopt_cost **out, int device, int scalar_bytes, int n_params,
                          int n_outputs, int n_planes, int n_aux, const char *setup_body,
                          const char *residual_body, const char *jacobian_body, const void *data,
                          int64_t plane_stride, int64_t count, unsigned flags) {
  if (!out) return fail(MOPT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (scalar_bytes != 4 && scalar_bytes != 8)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 or 8");
  if (count < 1 || (n_planes > 0 && (!data || plane_stride < count)))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad data planes / count");
  std::unique_ptr<mopt_cost, void (*)(mopt_cost *)> c(new (std::nothrow) mopt_cost, destroyCost);
  if (!c) return fail(MOPT_ERR_HIP, "out of host memory");
  c->scalar_bytes = scalar_bytes;
  c->model = kModelJit;
  c->n_params = n_params;
  c->n_out = n_outputs;
  c->count = count;
  c->num_tiles = 1;
  c->data_stride = count;
  int rc = commonCreate(c.get(), device);
  if (rc != MOPT_OK) return rc;
  if (!mopt::jitCreate(scalar_bytes, n_params, n_outputs, n_planes, n_aux, setup_body,
                       residual_body, jacobian_body, c->jit))
    return fail(MOPT_ERR_INVALID_ARGUMENT, mopt::jitLastError());
  if (n_planes > 0) {
    // plane stride padded to whole 16-byte packs (the sweep loads 16 bytes per lane and plane);
    // the tail of the last pack is zero-filled and enters no sum
    const int vec = 16 / scalar_bytes;
    const long long padded = (count + vec - 1) / vec * vec;
    c->data_stride = padded;
    const size_t row = size_t(count) * scalar_bytes, pitch = size_t(padded) * scalar_bytes;
    MOPT_HIP_TRY(deviceAlloc(&c->d_tiles, pitch * n_planes));
    MOPT_HIP_TRY(hipMemset(c->d_tiles, 0, pitch * n_planes));
    MOPT_HIP_TRY(hipMemcpy2D(c->d_tiles, pitch, data, size_t(plane_stride) * scalar_bytes, row,
                             size_t(n_planes),
                             (flags & MOPT_INPUT_DEVICE) ? hipMemcpyDeviceToDevice
                                                         : hipMemcpyHostToDevice));
  }
  mopt_cost_set_covariance(c.get(), nullptr);
  c->state_version = 0;
  *out = c.release();
  return MOPT_OK;
}

int mopt_cost_destroy(mopt_cost *cost) {
  destroyCost(cost);
  return MOPT_OK;
}

int mopt_cost_set_covariance(mopt_cost *c, const void *cov_colmajor) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  const int m = c->n_out;
  constexpr int kCovSlots = mopt::kMaxWideOutputs * mopt::kMaxWideOutputs;
  if (m < 1 || m * m > kCovSlots) return fail(MOPT_ERR_INVALID_ARGUMENT, "output dimension out of range");
  double dense[kCovSlots];  // row-major m x m
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double v = (a == b) ? 1.0 : 0.0;
      if (cov_colmajor)
        v = c->scalar_bytes == 8 ? static_cast<const double *>(cov_colmajor)[b * m + a]
                                 : double(static_cast<const float *>(cov_colmajor)[b * m + a]);
      dense[a * m + b] = v;
    }
  double previous[kCovSlots];
  std::memcpy(previous, c->cov_m, sizeof previous);
  for (int k = 0; k < kCovSlots; ++k) c->cov_m[k] = 0.0;
  for (int k = 0; k < m * m; ++k) c->cov_m[k] = dense[k];
  for (int k = 0; k < 9; ++k) c->cov[k] = (k % 4 == 0) ? 1.0 : 0.0;
  if (m <= 3)
    for (int a = 0; a < m; ++a)
      for (int b = 0; b < m; ++b) c->cov[a * 3 + b] = dense[a * m + b];
  bool identity = true, symmetric = true;
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      if (dense[a * m + b] != (a == b ? 1.0 : 0.0)) identity = false;
      if (dense[a * m + b] != dense[b * m + a]) symmetric = false;
    }
  c->cov_mode = identity ? mopt::kCovIdentity
                         : (symmetric ? mopt::kCovSymmetric : mopt::kCovGeneral);
  if (std::memcmp(previous, c->cov_m, sizeof previous) != 0) c->state_version += 1;
  return MOPT_OK;
}

int mopt_cost_set_loss(mopt_cost *c, int loss_kind, double parameter) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  if (loss_kind != MOPT_LOSS_NONE && loss_kind != MOPT_LOSS_GEMAN_MCCLURE)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown loss_kind");
  if (c->loss_kind != loss_kind || c->loss_param != parameter) c->state_version += 1;
  c->loss_kind = loss_kind;
  c->loss_param = parameter;
  return MOPT_OK;
}

int mopt_cost_set_kernel_variant(mopt_cost *c, int variant) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  if (variant < MOPT_KERNEL_AUTO || variant > MOPT_KERNEL_MOMENTS_ALWAYS)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown kernel variant");
  if ((variant == MOPT_KERNEL_MOMENTS || variant == MOPT_KERNEL_MOMENTS_ALWAYS) &&
      c->model != kModelPoint2Point)
    return fail(MOPT_ERR_UNSUPPORTED, "only point2point Jacobians are affine in the data point");
  if (c->variant != variant) c->state_version += 1;
  c->variant = variant;
  return MOPT_OK;
}

int mopt_cost_info(const mopt_cost *c, int64_t *count, int *n, int *m, int *scalar_bytes,
                   int *device) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  if (count) *count = c->count;
  if (n) *n = c->n_params;
  if (m) *m = c->n_out;
  if (scalar_bytes) *scalar_bytes = c->scalar_bytes;
  if (device) *device = c->device;
  return MOPT_OK;
}

namespace {
// A sweep went onto a stream this cost does not own: leave a marker behind it, so that
// destroy / set_data can wait for it before the cost's buffers go back to the pool.
int markForeignStream(mopt_cost *c, hipStream_t s) {
  if (s == c->stream) {
    c->own_async_pending = true;
    c->hip_pending = true;
    return MOPT_OK;
  }
  if (!c->foreign_done)
    MOPT_HIP_TRY(hipEventCreateWithFlags(&c->foreign_done, hipEventDisableTiming));
  MOPT_HIP_TRY(hipEventRecord(c->foreign_done, s));
  c->foreign_pending = true;
  return MOPT_OK;
}
}  // namespace

namespace {
// An asynchronous sweep writes the cost's partial rows from a HIP stream: a sweep a linked cost queued
// ahead for this one through the direct path (aql.hpp) is on another queue and must have finished.
void settleDirectPrefetch(mopt_cost *c) {
  if (c->prefetch.pending && c->aql_touched && c->aql_queue) {
    (void)mopt_detail::aqlDrain(c->aql_queue);
    c->aql_touched = false;
    c->prefetch.pending = false;
  }
}
}  // namespace

int mopt_cost_linearize_async(mopt_cost *c, int jacobian_mode, const void *x, double *d_result,
                              void *hip_stream) {
  if (!c || !x || !d_result) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  settleDirectPrefetch(c);
  // hip_stream is the hipStream_t itself; NULL is HIP's null (legacy default) stream, which is
  // also what torch's default stream is
  const hipStream_t s = static_cast<hipStream_t>(hip_stream);
  mopt::PeerCombine pc;
  if (c->combine.mode == MOPT_COMBINE_PEER) {  // the sums of all ranks end up in d_result
    pc = nextPeerCombine(c, 0);
    c->launch_peers = &pc;
  }
  const int rc = linearizeAsyncImpl(c, jacobian_mode, x, d_result, s);
  c->launch_peers = nullptr;
  return rc != MOPT_OK ? rc : markForeignStream(c, s);
}

int mopt_cost_compute_async(mopt_cost *c, const void *x, double *d_sum_sq, void *hip_stream) {
  if (!c || !x || !d_sum_sq) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  settleDirectPrefetch(c);
  const hipStream_t s = static_cast<hipStream_t>(hip_stream);
  mopt::PeerCombine pc;
  if (c->combine.mode == MOPT_COMBINE_PEER) {
    pc = nextPeerCombine(c, costOffset(c));
    c->launch_peers = &pc;
  }
  const int rc = costAsyncImpl(c, x, d_sum_sq, s);
  c->launch_peers = nullptr;
  return rc != MOPT_OK ? rc : markForeignStream(c, s);
}

namespace {
bool cacheMatches(const mopt_cost *c, const void *x, int mode_or_any) {
  return c->cache.valid && c->cache.version == c->state_version &&
         (mode_or_any < 0 || c->cache.mode == mode_or_any) &&
         std::memcmp(c->cache.x, x, size_t(c->n_params) * c->scalar_bytes) == 0;
}
// Where the linearization sweep is as cheap as the cost sweep, speculation always pays.
bool speculationIsFree(const mopt_cost *c) {
  return c->model == kModelPoint2Point && c->variant != MOPT_KERNEL_LITERAL && !c->matcher;
}
// A kept result is about to be replaced or ignored without having answered a linearize.
void noteKeptResultUnused(mopt_cost *c) {
  if (c->spec_kept_unused) {
    c->spec_kept_unused = false;
    c->spec_unused += 1;
  }
}
bool speculationPays(const mopt_cost *c) {
  return speculationIsFree(c) || c->spec_unused < 2 || c->spec_unused <= c->spec_used;
}
void cacheStore(mopt_cost *c, const void *x, int mode) {
  c->cache.valid = true;
  c->cache.mode = mode;
  c->cache.version = c->state_version;
  std::memcpy(c->cache.x, x, size_t(c->n_params) * c->scalar_bytes);
  std::memcpy(c->cache.result, c->h_result, resultCount(c) * sizeof(double));
}

// ---- linked costs: the sweeps of one problem's costs at one x, in flight together ---------------
bool prefetchable(const mopt_cost *c) { return c->combine.mode == MOPT_COMBINE_NONE && !c->matcher; }

bool prefetchMatches(const mopt_cost *c, const void *x, bool cost_only, int mode) {
  const auto &p = c->prefetch;
  return p.pending && p.version == c->state_version && p.cost_only == cost_only &&
         (cost_only || p.mode == mode) &&
         std::memcmp(p.x, x, size_t(c->n_params) * c->scalar_bytes) == 0;
}

// What mopt_cost_compute runs for this cost: the linearization sweep when its result is likely
// to be asked for next (speculation), else the cost sweep.
bool computeRunsLinearization(const mopt_cost *c) {
  return c->speculate && c->last_jac_mode >= 0 && speculationPays(c);
}

// What a linked cost's own call at this x is going to want (as prefetchSiblings decides it): false
// when nothing is to be queued for it.
bool siblingWants(mopt_cost *s, bool linearize_call, const void *x, bool *cost_only, int *mode) {
  *cost_only = false;
  *mode = s->last_jac_mode;
  if (linearize_call) {
    if (*mode < 0 || (s->speculate && cacheMatches(s, x, *mode))) return false;
  } else {
    if (s->speculate && cacheMatches(s, x, -1)) return false;
    *cost_only = !computeRunsLinearization(s);
    if (*cost_only) *mode = 0;
  }
  return !prefetchMatches(s, x, *cost_only, *mode);  // (else already in flight)
}

// `c` has just been asked at x (`linearize_call`: for a linearization, else for a cost): queue for
// every linked cost the sweep its own call at this x is going to want.  A guess that turns out
// wrong costs one unused sweep; failures here are left for the sibling's own call to report.
void prefetchSiblings(mopt_cost *c, bool linearize_call, const void *x) {
  for (mopt_cost *s : c->siblings) {
    if (s == c || !prefetchable(s) || s->scalar_bytes != c->scalar_bytes ||
        s->n_params != c->n_params)
      continue;
    bool cost_only = false;
    int mode = 0;
    if (!siblingWants(s, linearize_call, x, &cost_only, &mode)) continue;
    if (hipSetDevice(s->device) != hipSuccess) continue;
    unsigned long long sequence = 0;
    if (launchPublishedSweep(s, cost_only, mode, x, &sequence) != MOPT_OK) {
      s->prefetch.pending = false;
      continue;
    }
    s->prefetch.pending = true;
    s->prefetch.cost_only = cost_only;
    s->prefetch.mode = mode;
    s->prefetch.version = s->state_version;
    s->prefetch.sequence = sequence;
    std::memcpy(s->prefetch.x, x, size_t(s->n_params) * s->scalar_bytes);
  }
}

// One blocking sweep of `c`, answered by the sweep a linked cost queued for it when that matches;
// otherwise launched now, together with the linked costs' sweeps at the same x.
int sweepWithSiblings(mopt_cost *c, bool cost_only, int mode, const void *x, bool linearize_call) {
  if (prefetchMatches(c, x, cost_only, mode)) {
    c->prefetch.pending = false;
    c->stat_prefetch_hits += 1;
    return waitPublishedSweep(c, c->prefetch.sequence);
  }
  c->prefetch.pending = false;  // whatever was queued is superseded (same stream: ordered before)
  if (c->siblings.empty() || !prefetchable(c)) return blockingSweep(c, cost_only, mode, x);
  unsigned long long sequence = 0;
  const int rc = launchPublishedSweep(c, cost_only, mode, x, &sequence);
  if (rc != MOPT_OK) return rc;
  prefetchSiblings(c, linearize_call, x);
  MOPT_HIP_TRY(hipSetDevice(c->device));
  return waitPublishedSweep(c, sequence);
}
}  // namespace

int mopt_costs_link(mopt_cost *const *costs, int num_costs) {
  if (num_costs < 0 || (num_costs > 0 && !costs))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad cost list");
  for (int k = 0; k < num_costs; ++k)
    if (!costs[k]) return fail(MOPT_ERR_INVALID_ARGUMENT, "a cost is NULL");
  for (int k = 0; k < num_costs; ++k) {
    mopt_cost *c = costs[k];
    for (mopt_cost *old : c->siblings)  // leave the group it was in
      old->siblings.erase(std::remove(old->siblings.begin(), old->siblings.end(), c), old->siblings.end());
    c->siblings.clear();
  }
  for (int k = 0; k < num_costs; ++k)
    for (int j = 0; j < num_costs; ++j)
      if (costs[j] != costs[k] &&
          std::find(costs[k]->siblings.begin(), costs[k]->siblings.end(), costs[j]) == costs[k]->siblings.end())
        costs[k]->siblings.push_back(costs[j]);
  return MOPT_OK;
}

int mopt_cost_link_stats(const mopt_cost *c, int64_t *answered_ahead) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  if (answered_ahead) *answered_ahead = c->stat_prefetch_hits;
  return MOPT_OK;
}

int mopt_cost_linearize(mopt_cost *c, int jacobian_mode, const void *x, void *hessian, void *b,
                        void *sum_sq) {
  if (!c || !x) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (jacobian_mode < MOPT_JAC_ANALYTIC || jacobian_mode > MOPT_JAC_ANALYTIC_RIGHT)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown jacobian_mode");
  if (c->speculate && cacheMatches(c, x, jacobian_mode)) {
    c->stat_cache_hits += 1;
    if (c->spec_kept_unused) {
      c->spec_kept_unused = false;
      c->spec_used += 1;
    }
    c->last_jac_mode = jacobian_mode;
    storeResult(c, c->cache.result, hessian, b, sum_sq);
    return MOPT_OK;
  }
  noteKeptResultUnused(c);
  MOPT_HIP_TRY(hipSetDevice(c->device));
  const int rc = sweepWithSiblings(c, false, jacobian_mode, x, true);
  if (rc != MOPT_OK) return rc;
  c->last_jac_mode = jacobian_mode;
  if (c->speculate) cacheStore(c, x, jacobian_mode);
  storeResult(c, c->h_result, hessian, b, sum_sq);
  return MOPT_OK;
}

int mopt_cost_compute(mopt_cost *c, const void *x, void *sum_sq) {
  if (!c || !x || !sum_sq) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (c->speculate && cacheMatches(c, x, -1)) {
    c->stat_cache_hits += 1;
    storeResult(c, c->cache.result, nullptr, nullptr, sum_sq);
    return MOPT_OK;
  }
  MOPT_HIP_TRY(hipSetDevice(c->device));
  noteKeptResultUnused(c);
  if (computeRunsLinearization(c)) {
    // the linearization sweep also yields sum r^T r; keep all of it for the linearize that follows
    const int rc = sweepWithSiblings(c, false, c->last_jac_mode, x, false);
    if (rc != MOPT_OK) return rc;
    cacheStore(c, x, c->last_jac_mode);
    c->spec_kept_unused = true;
  } else {
    const int rc = sweepWithSiblings(c, true, 0, x, false);
    if (rc != MOPT_OK) return rc;
  }
  storeResult(c, c->h_result, nullptr, nullptr, sum_sq);
  return MOPT_OK;
}

int mopt_cost_set_speculation(mopt_cost *c, int enabled) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  c->speculate = enabled != 0;
  c->cache.valid = false;
  c->spec_kept_unused = false;
  c->spec_unused = c->spec_used = 0;
  return MOPT_OK;
}

int mopt_cost_direct_dispatches(const mopt_cost *c, int64_t *sweeps) {
  if (!c || !sweeps) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  *sweeps = c->stat_direct_sweeps;
  return MOPT_OK;
}

int mopt_cost_lm_choice_stats(const mopt_cost *c, int64_t *points, int64_t *literal_points) {
  if (!c || !points || !literal_points) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  *points = c->stat_lm_choice_points;
  *literal_points = c->stat_lm_literal_points;
  return MOPT_OK;
}

int mopt_device_trim(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "no such device");
  if (!mopt_detail::aqlTrim(device))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "a cost of this process still lives on the device");
  return MOPT_OK;
}

int mopt_cost_stats(const mopt_cost *c, int64_t *sweeps, int64_t *cache_hits) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  if (sweeps) *sweeps = c->stat_sweeps;
  if (cache_hits) *cache_hits = c->stat_cache_hits;
  return MOPT_OK;
}

// ---- multi-process shard group: one rank per GPU, RCCL over xGMI ----------------------------
int mopt_comm_unique_id(void *id_out, int id_bytes) {
  if (!id_out || id_bytes < int(sizeof(ncclUniqueId)))
    return fail(MOPT_ERR_INVALID_ARGUMENT, "id buffer must hold MOPT_COMM_ID_BYTES bytes");
  ncclUniqueId id;
  MOPT_NCCL_TRY(ncclGetUniqueId(&id));
  std::memset(id_out, 0, size_t(id_bytes));
  std::memcpy(id_out, &id, sizeof id);
  return MOPT_OK;
}

int mopt_cost_comm_init_rank(mopt_cost *c, const void *id, int rank, int num_ranks) {
  if (!c || !id || num_ranks < 1 || rank < 0 || rank >= num_ranks)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  if (c->comm) return fail(MOPT_ERR_INVALID_ARGUMENT, "a communicator is already attached");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  const ShardCombine &sc = c->combine;
  if ((sc.host_block || sc.peer_attached) && (sc.rank != rank || sc.num_ranks != num_ranks))
    return fail(MOPT_ERR_INVALID_ARGUMENT,
                "rank / num_ranks differ from the transport already attached to this cost");
  MOPT_NCCL_TRY(ncclCommInitRank(&c->comm, num_ranks, uid, rank));
  c->comm_size = num_ranks;
  c->combine.rank = rank;
  c->combine.num_ranks = num_ranks;
  c->combine.mode = MOPT_COMBINE_RCCL;
  c->cache.valid = false;
  return MOPT_OK;
}

int mopt_cost_comm_info(const mopt_cost *c, int *num_ranks, int *rank) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  int count = 0, user_rank = -1;
  if (c->comm) {
    MOPT_NCCL_TRY(ncclCommCount(c->comm, &count));
    MOPT_NCCL_TRY(ncclCommUserRank(c->comm, &user_rank));
  }
  if (num_ranks) *num_ranks = count;
  if (rank) *rank = user_rank;
  return MOPT_OK;
}

int mopt_cost_stream(mopt_cost *c, void **hip_stream) {
  if (!c || !hip_stream) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  *hip_stream = c->stream;
  return MOPT_OK;
}

int mopt_cost_synchronize(mopt_cost *c) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  if (c->aql_touched && c->aql_queue) {  // whatever went through the direct path (aql.hpp) as well
    (void)mopt_detail::aqlDrain(c->aql_queue);
    c->aql_touched = false;
  }
  MOPT_HIP_TRY(hipStreamSynchronize(c->stream));
  c->own_async_pending = false;
  c->hip_pending = false;
  return MOPT_OK;
}

int mopt_cost_set_profiling(mopt_cost *c, int enabled) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  const int rc = resolvePendingEvents(c);
  if (rc != MOPT_OK) return rc;
  c->profiling = enabled > 0 ? enabled : 0;
  c->profiling_tick = 0;
  c->sweep_ms_total = 0.0;
  c->sweep_launches = 0;
  return MOPT_OK;
}

int mopt_cost_profile(mopt_cost *c, double *sweep_ms_total, int64_t *sweep_launches) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  const int rc = resolvePendingEvents(c);
  if (rc != MOPT_OK) return rc;
  if (sweep_ms_total) *sweep_ms_total = c->sweep_ms_total;
  if (sweep_launches) *sweep_launches = c->sweep_launches;
  return MOPT_OK;
}

}  // extern "C"
