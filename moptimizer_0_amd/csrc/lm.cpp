// mopt_lm_minimize: the Levenberg-Marquardt loop of src/levenberg_marquadt_dyn.cpp:34-119 with the
// iteration resident on the device (step kernel: lm_kernels.hip; resident sweeps:
// sweep_kernels.hip).  The host only queues launches a few trial points ahead of the device and
// watches a progress word in mapped memory; it takes no decision and copies nothing per iteration.
#include "cost_state.hpp"

#include <chrono>
#include <cmath>
#include <cstring>

using namespace mopt_detail;

namespace {

int ensureWorkspace(mopt_cost *c) {
  if (!c->d_lm_control)
    MOPT_HIP_TRY(deviceAlloc(reinterpret_cast<void **>(&c->d_lm_control),
                             mopt::kLmControlBlocks * sizeof(mopt::LmControl)));
  if (!c->d_lm_state) MOPT_HIP_TRY(deviceAlloc(&c->d_lm_state, 4096));  // >= LmState<double>
  if (!c->h_lm_report) {
    MOPT_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_lm_report), sizeof(mopt::LmReport),
                               hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_lm_report, 0, sizeof(mopt::LmReport));
    MOPT_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_lm_report_dev),
                                         c->h_lm_report, 0));
  }
  return MOPT_OK;
}

// The progress word: 2 * (step runs completed) + 1 once the loop has stopped.  The payload of the
// report is written once, before the word gets its low bit.
inline unsigned long long progressWord(const mopt::LmReport *live) {
  return __atomic_load_n(&live->flag, __ATOMIC_ACQUIRE);
}

}  // namespace

extern "C" {

int mopt_lm_minimize(mopt_cost *const *costs, int num_costs, const int *jacobian_modes, void *x,
                     const mopt_lm_options *options, mopt_lm_report *report_out) {
  if (!costs || num_costs < 1 || !jacobian_modes || !x)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument / no costs");
  if (num_costs > mopt::kLmMaxCosts)
    return fail(MOPT_ERR_UNSUPPORTED, "mopt_lm_minimize sums at most 4 costs");
  mopt_cost *lead = costs[0];
  for (int k = 0; k < num_costs; ++k) {
    const mopt_cost *c = costs[k];
    if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "a cost is NULL");
    if (c->device != lead->device || c->scalar_bytes != lead->scalar_bytes ||
        c->n_params != lead->n_params)
      return fail(MOPT_ERR_INVALID_ARGUMENT,
                  "the costs of one problem share device, scalar type and parameter count");
    if (c->combine.mode != MOPT_COMBINE_NONE && c->combine.mode != MOPT_COMBINE_PEER)
      return fail(MOPT_ERR_UNSUPPORTED,
                  "a sharded cost needs the sums of all ranks on the device: select "
                  "MOPT_COMBINE_PEER (mopt_cost_set_combine) for mopt_lm_minimize");
  }
  mopt_lm_options opt;
  opt.max_iterations = 15;    // optimizer.h:19
  opt.lm_max_iterations = 3;  // levenberg_marquadt_dyn.cpp:9
  opt.manifold = 0;
  opt.window = 0;
  if (options) opt = *options;
  if (opt.max_iterations < 0 || opt.lm_max_iterations < 0)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "Optimization::max_iterations cannot be less than 0.");
  if (opt.manifold < 0 || opt.manifold > 2)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "mopt_lm_options.manifold is 0 (none), 1 (left) or 2 (right)");
  if (opt.manifold && lead->n_params != kNumParams)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "the SE(3) update applies to 6-parameter poses");
  mopt_lm_report rep;
  std::memset(&rep, 0, sizeof rep);
  rep.status = MOPT_LM_MAXIMUM_ITERATIONS_REACHED;
  if (opt.max_iterations == 0) {  // the reference's loop body never runs
    if (report_out) *report_out = rep;
    return MOPT_OK;
  }
  if (opt.lm_max_iterations == 0) {
    // no trial point is ever formed (:77): every outer iteration linearizes at the same x0, finds
    // the same cost, and the loop runs into its limit — unless that cost is already ~0 (:62-64)
    double y0 = 0.0;
    for (int k = 0; k < num_costs; ++k) {
      double hb[mopt_detail::kResultSlots];
      double yk = 0.0;
      float yf = 0.f;
      if (costs[k]->matcher) {  // cost->update(x0) comes before every linearization (:54)
        const int rcu = mopt_icp_update(costs[k], x, nullptr);
        if (rcu != MOPT_OK) return rcu;
      }
      const int rck = mopt_cost_linearize(costs[k], jacobian_modes[k], x, hb, hb + 64,
                                          lead->scalar_bytes == 8 ? static_cast<void *>(&yk)
                                                                  : static_cast<void *>(&yf));
      if (rck != MOPT_OK) return rck;
      y0 += lead->scalar_bytes == 8 ? yk : double(yf);
    }
    const double eps = lead->scalar_bytes == 8 ? 2.220446049250313e-16 : 1.1920929e-7;
    rep.status = std::fabs(y0) < 8.0 * eps ? MOPT_LM_CONVERGED : MOPT_LM_MAXIMUM_ITERATIONS_REACHED;
    rep.iterations = rep.status == MOPT_LM_CONVERGED ? 0 : opt.max_iterations;
    rep.sweeps = 1;
    rep.cost = y0;
    if (report_out) *report_out = rep;
    return MOPT_OK;
  }
  MOPT_HIP_TRY(hipSetDevice(lead->device));
  int rc = ensureWorkspace(lead);
  if (rc != MOPT_OK) return rc;
  hipStream_t s = lead->stream;
  // the loop's kernels go to the HIP stream: a later blocking sweep of these costs on the direct
  // path (aql.hpp) waits for whatever the loop leaves queued (its window of early-exit launches)
  for (int k = 0; k < num_costs; ++k) costs[k]->hip_pending = true;

  mopt::LmProblem problem;
  problem.num_costs = num_costs;
  problem.n = lead->n_params;
  problem.max_iterations = opt.max_iterations;
  problem.lm_max_iterations = opt.lm_max_iterations;
  problem.manifold = opt.manifold;
  problem.control = lead->d_lm_control;
  problem.state = lead->d_lm_state;
  problem.report = lead->h_lm_report_dev;
  // Several costs whose sweeps leave rows of the same H | b | sum_sq form: the rows go behind one
  // another into the last cost's buffer and one finalize kernel — the one that takes the LM step —
  // reduces them all, instead of one finalize launch per cost.  (Not for sharded costs, whose
  // sums are exchanged per cost, nor for rows of moments, which each cost contracts itself.)
  mopt_cost *last = costs[num_costs - 1];
  const int row_length = residentDenseRow(last, jacobian_modes[num_costs - 1]);
  static const bool merge_enabled = [] {  // MOPT_LM_MERGE=0: one finalize per cost (for comparison)
    const char *v = std::getenv("MOPT_LM_MERGE");
    return !(v && v[0] == '0');
  }();
  bool merged = merge_enabled && num_costs > 1 && row_length > 0;
  int row_offset[mopt::kLmMaxCosts + 1] = {0};
  for (int k = 0; k < num_costs; ++k) {
    merged = merged && residentDenseRow(costs[k], jacobian_modes[k]) == row_length && !costs[k]->matcher &&
             costs[k]->combine.mode == MOPT_COMBINE_NONE;
    row_offset[k + 1] = row_offset[k] + residentGrid(costs[k], jacobian_modes[k]);
    for (int j = 0; j < k; ++j) merged = merged && costs[j] != costs[k];
  }
  // all rows must fit the last cost's buffer in doubles, not only in rows: a wide model's row is up
  // to 273 values against the kPartialRowSlots the buffer is sized by
  merged = merged && row_offset[num_costs] <= last->max_grid &&
           size_t(row_offset[num_costs]) * size_t(row_length) <=
               size_t(last->max_grid) * mopt_detail::kPartialRowSlots;
  problem.merged = merged ? 1 : 0;
  static const bool set_enabled = [] {  // MOPT_LM_SET=0: a sweep launch per cost (for comparison)
    const char *v = std::getenv("MOPT_LM_SET");
    return !(v && v[0] == '0');
  }();
  const bool one_launch =
      set_enabled && merged && residentSetSupported(costs, num_costs, jacobian_modes);
  for (int k = 0; k < num_costs; ++k) {
    rc = residentPrepare(costs[k], jacobian_modes[k], s, &problem.cost[k],
                         merged ? last->d_partials + size_t(row_offset[k]) * row_length : nullptr);
    if (rc != MOPT_OK) return rc;
    costs[k]->cache.valid = false;
    if (residentPerIterate(costs[k], jacobian_modes[k])) problem.fd_per_iterate = 1;
    if (costs[k]->matcher) {
      problem.rematch = 1;               // its update(x) runs on the device, inside the loop
      costs[k]->state_version += 1;      // the correspondences will have changed
    }
  }
  // The loop of all costs runs on the lead cost's stream.  Sweeps a caller queued asynchronously
  // for another cost (its own stream, or the caller's) write the same partial rows: wait for them.
  // (Running each cost's sweeps on its own stream instead, joined by events, was measured and
  // dropped: a cross-queue dependency costs 7-13 us on this runtime, more than the sweeps that
  // would overlap — profiles/NOTES.md, "Device-resident Levenberg-Marquardt".)
  // So does a sweep queued ahead for a linked cost (mopt_costs_link) that nobody has consumed yet.
  for (int k = 0; k < num_costs; ++k) {
    mopt_cost *ck = costs[k];
    const bool other_stream = ck->stream != s;
    // (a sweep queued ahead through the direct path — aql.hpp — is on a queue of its own whichever
    // cost leads: stream order does not hold it back)
    if (ck->foreign_pending || (other_stream && (ck->own_async_pending || ck->prefetch.pending)) ||
        (ck->prefetch.pending && ck->aql_touched))
      MOPT_HIP_TRY(quiesceCost(ck));
    ck->prefetch.pending = false;  // its partial rows and result are about to be overwritten
  }
  unsigned long long base_sequence[mopt::kLmMaxCosts];
  for (int k = 0; k < num_costs; ++k) base_sequence[k] = costs[k]->combine.sequence + 1;

  mopt::LmReport *live = lead->h_lm_report;
  __atomic_store_n(&live->flag, 0ull, __ATOMIC_RELEASE);
  const long long max_points =
      1 + (long long)opt.max_iterations *
              ((opt.lm_max_iterations > 0 ? opt.lm_max_iterations : 1) + (problem.rematch ? 1 : 0));
  // One small point2point cost (a few tiles: the reference's own test sizes): the whole loop in one
  // launch of one workgroup (sweep_kernels.hip p2pSolveSmallKernel) instead of two launches per
  // evaluated point — the same sums added in another order, so the same iterates to rounding.
  // MOPT_LM_ONE_LAUNCH_TILES: the largest tile count that goes this way (default and maximum
  // solveSmallMaxTiles(); a tile is 512 fp64 / 1024 fp32 correspondences; 0 = never), read per call.
  int small_tiles = mopt::solveSmallMaxTiles();
  if (const char *v = std::getenv("MOPT_LM_ONE_LAUNCH_TILES")) {
    const int asked = std::atoi(v);
    if (asked >= 0 && asked < small_tiles) small_tiles = asked;
  }
  const bool one_workgroup = num_costs == 1 && lead->model == kModelPoint2Point && !lead->matcher &&
                             lead->combine.mode == MOPT_COMBINE_NONE && problem.cost[0].moments &&
                             problem.n == kNumParams && lead->num_tiles >= 1 &&
                             lead->num_tiles <= small_tiles &&
                             max_points <= 4096;  // (one kernel for the whole loop: at ~8 us a point, <= 35 ms)
  if (one_workgroup) {
    if (lead->scalar_bytes == 8)
      MOPT_HIP_TRY(mopt::launchP2PSolveSmall<double>(
          static_cast<const double *>(lead->d_tiles), lead->num_tiles,
          static_cast<const mopt::P2PSweepArgs<double> *>(lead->d_lm_args), lead->d_lm_basis,
          lead->d_result, problem, static_cast<const double *>(x), int(max_points), lead->cov_mode, s));
    else
      MOPT_HIP_TRY(mopt::launchP2PSolveSmall<float>(
          static_cast<const float *>(lead->d_tiles), lead->num_tiles,
          static_cast<const mopt::P2PSweepArgs<float> *>(lead->d_lm_args), lead->d_lm_basis,
          lead->d_result, problem, static_cast<const float *>(x), int(max_points), lead->cov_mode, s));
  } else if (lead->scalar_bytes == 8)
    MOPT_HIP_TRY(mopt::launchLmStep<double>(problem, true, static_cast<const double *>(x), s));
  else
    MOPT_HIP_TRY(mopt::launchLmStep<float>(problem, true, static_cast<const float *>(x), s));

  // One sweep per evaluated point: the linearization at x0, then at most lm_max_iterations trial
  // points per outer iteration.  The host stays `window` points ahead of the device; whatever is
  // still queued when the step kernel stops finds control->done set and returns at once.
  // (with an ICP cost an accepted point is searched and swept again before it is linearized)
  static const int default_window = envInt("MOPT_LM_WINDOW", 3);
  const int window = opt.window > 0 ? opt.window : default_window;
  long long enqueued = 0;
  const auto started = std::chrono::steady_clock::now();
  unsigned long long spins = 0;
  for (;;) {
    const unsigned long long word = progressWord(live);
    if (word & 1ull) break;  // the loop has stopped and its report is complete
    const long long steps = (long long)(word >> 1);
    const long long completed = steps > 0 ? steps - 1 : 0;  // the init run counts one
    bool queued = false;
    while (!one_workgroup && enqueued < max_points && enqueued - completed < window) {
      // per point: every cost's sweep + finalize; the last finalize also takes the LM step
      if (one_launch) {
        rc = residentSweepSet(costs, num_costs, jacobian_modes, row_offset, problem.control, s);
        if (rc != MOPT_OK) return rc;
      }
      for (int k = 0; k < num_costs && !one_launch; ++k) {
        const bool is_last = k == num_costs - 1;
        rc = residentSweep(costs[k], jacobian_modes[k], problem.control, s, base_sequence[k],
                           is_last ? &problem : nullptr, k, !merged);
        if (rc != MOPT_OK) return rc;
      }
      if (merged) {
        rc = residentFinalizeMerged(last, row_offset[num_costs], row_length, problem.control, s,
                                    &problem, num_costs - 1);
        if (rc != MOPT_OK) return rc;
      }
      ++enqueued;
      queued = true;
    }
    if (queued) continue;
    if ((++spins & 0x3fff) == 0) {
      const hipError_t q = hipStreamQuery(s);
      if (q != hipSuccess && q != hipErrorNotReady)
        return fail(MOPT_ERR_HIP, std::string("device-resident LM failed: ") + hipGetErrorString(q));
      if (q == hipSuccess) {
        if (progressWord(live) & 1ull) break;
        if (one_workgroup || enqueued >= max_points)
          return fail(MOPT_ERR_HIP, "device-resident LM drained without reaching a status");
      }
      if (std::chrono::steady_clock::now() - started > std::chrono::seconds(120))
        return fail(MOPT_ERR_HIP, "timed out waiting for the device-resident LM (120 s)");
    }
    __builtin_ia32_pause();
  }
  mopt::LmReport snap;
  std::memcpy(&snap, live, sizeof snap);

  const long long trials = (long long)snap.trials;
  for (int k = 0; k < num_costs; ++k) {
    mopt_cost *c = costs[k];
    c->stat_sweeps += trials;
    if (c->combine.mode == MOPT_COMBINE_PEER) c->combine.sequence += (unsigned long long)trials;
    if (problem.fd_per_iterate && residentPerIterate(c, jacobian_modes[k])) {
      c->stat_lm_choice_points += trials;
      c->stat_lm_literal_points += (long long)snap.pad[0];
    }
  }
  if (lead->scalar_bytes == 8)
    for (int i = 0; i < problem.n; ++i) static_cast<double *>(x)[i] = snap.x[i];
  else
    for (int i = 0; i < problem.n; ++i) static_cast<float *>(x)[i] = float(snap.x[i]);
  rep.status = int(snap.status);
  rep.iterations = int(snap.iterations);
  rep.sweeps = trials;
  rep.cost = snap.cost;
  rep.lambda = snap.lambda;
  if (report_out) *report_out = rep;
  if ((unsigned long long)snap.peer_status == mopt::kStatusPeerTimeout)
    return fail(MOPT_ERR_PEER_TIMEOUT,
                "a rank did not deliver its sums to this device in time (MOPT_PEER_TIMEOUT_MS)");
  // the loop has stopped: what the window left queued on the stream are launches that find
  // control->done set and leave without touching anything — nothing a direct sweep has to wait for
  for (int k = 0; k < num_costs; ++k) costs[k]->hip_pending = false;
  return MOPT_OK;
}

}  // extern "C"
