// State behind the opaque handles of include/moptimizer_hip.h and the helpers the translation
// units of the C ABI share (c_abi.cpp: costs and sweeps; icp.cpp: correspondence search;
// group.cpp: single-process device group).  Internal to libmoptimizer_hip.so.
#pragma once

#include "moptimizer_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "jit_model.hpp"
#include "sweep.hpp"

namespace mopt_detail {

// records the message for mopt_last_error() (thread-local) and returns `code`
int fail(int code, const std::string &msg);


#define MOPT_HIP_TRY(expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(MOPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

#define MOPT_NCCL_TRY(expr)                                                                 \
  do {                                                                                      \
    ncclResult_t r_ = (expr);                                                               \
    if (r_ != ncclSuccess)                                                                  \
      return fail(MOPT_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(r_));      \
  } while (0)

enum ModelKind { kModelPoint2Point = 1, kModelReprojection = 2, kModelScalar = 3, kModelJit = 4 };
constexpr int kMaxParamBytes = mopt::kMaxWideParams * 8;
// results of one sweep: >= n*n + n + 1 for n <= 16 (273: run-time compiled wide models; every other
// model has n <= 8, 73 values, and the combine slots of sweep.hpp hold those)
constexpr int kResultSlots = 288;
constexpr int kPartialRowSlots = 96;  // the partial-row buffer is max_grid rows of this many doubles

inline int envInt(const char *name, int fallback) {
  const char *v = std::getenv(name);
  if (!v || !*v) return fallback;
  const int parsed = std::atoi(v);
  return parsed > 0 ? parsed : fallback;
}

}  // namespace mopt_detail

// Uniform grid over the target cloud of an ICP cost (built once, on the GPU: icp_grid.hip;
// resident in HBM).
struct IcpMatcher {
  void *d_sorted = nullptr;      // [num_targets][4] scalars grouped by cell
  int *d_cell_start = nullptr;   // [cells + 1]
  unsigned int *d_matched = nullptr;  // matched sources per wave of the last counting search
  long long num_waves = 0;
  long long num_sources = 0;  // as handed to mopt_icp_create (the cost holds those with finite coordinates)
  double origin[3] = {0, 0, 0};
  double cell = 1.0;
  int reach = 1;  // cells to the search radius (the search covers (2 reach + 1)^3 cells)
  int dims[3] = {1, 1, 1};
  double max_dist = 0.0;
  long long num_targets = 0;
  // Sources are stored in grid-cell order (of their un-warped position) so that neighbouring
  // lanes visit neighbouring cells; slot k of the tiles holds the caller's source d_order[k]
  // (device memory, `kept` entries: the sources with finite coordinates).
  int *d_order = nullptr;
  long long kept = 0;
};

// How the sums of a sharded cost are added over the ranks (MOPT_COMBINE_* of the header).
struct ShardCombine {
  int mode = MOPT_COMBINE_NONE;  // the transport blocking sweeps use now
  int rank = 0, num_ranks = 1;
  unsigned long long sequence = 0;  // collective sweeps so far; in step on every rank
  // MOPT_COMBINE_HOST: slot block in POSIX shared memory, registered with HIP so that the
  // finalize kernel of every rank publishes straight into it
  std::string shm_name;
  int shm_fd = -1;
  double *host_block = nullptr;      // host address
  double *host_block_dev = nullptr;  // the same memory as this device addresses it
  size_t host_bytes = 0;
  bool host_registered = false;
  // MOPT_COMBINE_PEER: uncached device memory, one block per rank, opened over IPC
  double *peer_own = nullptr;
  double *peer_blocks[mopt::kMaxPeers] = {nullptr};
  bool peer_opened[mopt::kMaxPeers] = {false};
  bool peer_attached = false;
  unsigned long long peer_timeout_ticks = 0;
  bool has(int m) const {
    return m == MOPT_COMBINE_HOST ? host_block != nullptr
                                  : (m == MOPT_COMBINE_PEER ? peer_attached : false);
  }
};

struct mopt_cost {
  int device = 0;
  int scalar_bytes = 8;
  int model = mopt_detail::kModelPoint2Point;
  int scalar_model = 0;  // mopt::ScalarModelKind when model == kModelScalar
  int n_params = mopt::kNumParams;
  int n_out = 3;
  long long data_stride = 0;  // scalar models: elements per data plane
  long long count = 0;
  int num_tiles = 0;
  int num_cus = 0;
  int max_grid = 0;

  void *d_tiles = nullptr;
  long long capacity_tiles = 0;  // tiles allocated in d_tiles
  double *d_partials = nullptr;
  double *d_result = nullptr;  // kResultDoubles
  // Mapped, fine-grained host memory the finalize kernel publishes into: 43 results + flag word.
  double *h_result = nullptr;
  unsigned long long *h_flag = nullptr;
  double *h_result_dev = nullptr;  // the same memory as the device addresses it
  unsigned long long *h_flag_dev = nullptr;
  unsigned long long sequence = 0;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;  // multi-process shard group (one rank per GPU), optional
  int comm_size = 1;
  ShardCombine combine;  // latency-optimised alternatives to the RCCL all-reduce (combine.cpp)
  // set around the launches of one sweep: the finalize kernel then also adds over the ranks
  const mopt::PeerCombine *launch_peers = nullptr;
  // Direct dispatch of the blocking sweeps (aql.hpp).  `aql_queue`: this cost's queue of the device's
  // pool (drawn on first use); `aql_now`: set around the launches of a sweep that may go there;
  // `aql_touched`: something was dispatched there since the queue was last drained (memory of this
  // cost must not be recycled before a drain); `hip_pending`: something was queued on `stream` that
  // a direct sweep would have to wait for (a path switch synchronises once).
  mopt_detail::AqlQueue *aql_queue = nullptr;
  bool aql_tried = false;
  bool aql_retained = false;  // counted by aqlRetain: aqlRelease when the cost goes
  mopt_detail::AqlSite aql_now;
  bool aql_touched = false;
  bool hip_pending = false;
  bool waiting_direct = false;  // the sweep being waited for went through the direct path
  bool sweep_went_direct = false;  // set by the launch itself (SweepTimer::aql) when it did
  bool aql_timed = false;       // ... and its dispatch is being timed (profiling)

  double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major, stride 3 (m <= 3), as double
  double cov_m[mopt::kMaxWideOutputs * mopt::kMaxWideOutputs] = {1};  // row-major m x m compact (generic models)
  int cov_mode = mopt::kCovIdentity;
  int loss_kind = MOPT_LOSS_NONE;
  double loss_param = 0.0;
  int variant = MOPT_KERNEL_AUTO;

  double camera[12];
  double frame[16];

  std::unique_ptr<IcpMatcher> matcher;  // point2point costs created by mopt_icp_create
  mopt::JitKernel jit;                  // kModelJit: the run-time compiled sweep

  // LM calls computeCost(xi) and, when the step is accepted, linearize(xi) right after
  // (levenberg_marquadt_dyn.cpp:86,112 then :55): with speculation on, computeCost runs the
  // linearization sweep (same HBM traffic as the cost sweep) and keeps its H | b | sum_sq, so the
  // following linearize at the same x costs no sweep at all.
  bool speculate = true;
  // Speculation is free only where the linearization sweep costs what the cost sweep costs (the
  // point2point moments sweep over fixed correspondences).  Elsewhere (forward differences of the
  // reprojection / scalar / user models: n + 1 residual evaluations per element; ICP costs, whose
  // update(x) invalidates the kept result before it can be used) it is kept only while it pays:
  // once more kept results have gone unused than used, computeCost goes back to cost-only sweeps.
  bool spec_kept_unused = false;  // a result kept by mopt_cost_compute that nobody has asked for yet
  int spec_unused = 0, spec_used = 0;
  int last_jac_mode = -1;
  unsigned long long state_version = 0;  // bumped when loss / covariance / variant change
  struct {
    bool valid = false;
    int mode = -1;
    unsigned long long version = 0;
    unsigned char x[mopt_detail::kMaxParamBytes] = {0};
    double result[mopt_detail::kResultSlots] = {0};
  } cache;
  long long stat_sweeps = 0;
  long long stat_direct_sweeps = 0;  // of them: dispatched by the library itself (aql.hpp)
  long long stat_lm_choice_points = 0;   // points of device-resident solves whose forward-difference sweep
  long long stat_lm_literal_points = 0;  // was chosen per point, and how many of them took the literal one
  // blocking sweeps launched on `stream` since the runtime was last handed a marker (boundCommandBatch)
  int launches_since_marker = 0;
  long long stat_cache_hits = 0;

  // Device-resident LM (mopt_lm_minimize, lm.cpp): this cost's sweep constants in HBM, written by
  // the step kernel per trial point; the static part (data, loss, covariance) is re-uploaded when
  // (state_version, Jacobian mode) change.  The first cost of a problem also owns the workspace.
  void *d_lm_args = nullptr;
  mopt::AffineBasis *d_lm_basis = nullptr;
  unsigned long long lm_uploaded_version = ~0ull;
  int lm_uploaded_mode = -1;
  const double *lm_uploaded_partials = nullptr;  // where the uploaded constants send the rows
  mopt::LmControl *d_lm_control = nullptr;
  void *d_lm_state = nullptr;
  mopt::LmReport *h_lm_report = nullptr;      // mapped host memory
  mopt::LmReport *h_lm_report_dev = nullptr;  // as the device addresses it

  // Sweeps enqueued on a caller's stream (the *_async entry points): the buffers of this cost must
  // not be recycled before that work has finished, and only an event on that stream can tell.
  hipEvent_t foreign_done = nullptr;
  // Costs of one problem: the caller's loop asks them one after the other at the same x
  // (levenberg_marquadt_dyn.cpp:52-59, :86).  Once linked (mopt_costs_link), the first one asked
  // queues the others' sweeps at that x too, each on its own stream; their own calls then only wait.
  std::vector<mopt_cost *> siblings;
  struct {
    bool pending = false;
    bool cost_only = false;
    int mode = 0;
    unsigned long long version = 0, sequence = 0;
    unsigned char x[mopt_detail::kMaxParamBytes] = {0};
  } prefetch;
  long long stat_prefetch_hits = 0;
  bool foreign_pending = false;
  bool own_async_pending = false;  // an asynchronous sweep was queued on this cost's own stream

  int profiling = 0;  // 0 off, N > 0: bracket every N-th sweep launch with events
  long long profiling_tick = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
  std::vector<hipEvent_t> free_events;
  double sweep_ms_total = 0.0;
  long long sweep_launches = 0;
};

namespace mopt_detail {

using mopt::kNumParams;
using mopt::kResultDoubles;

// Non-blocking streams are recycled per device: creating one costs the runtime ~1.5 ms and
// destroying it ~1.2 ms, which would dominate the construction of a small cost.  A released
// stream has been synchronised by its last owner.  (Streams still pooled at process exit are
// left to the runtime's own teardown.)
hipError_t acquireStream(int device, hipStream_t *out);
void releaseStream(int device, hipStream_t stream);

// Device memory of the current device through the size-class cache of device_pool.cpp.  Release
// only after the work that used the block has completed.
hipError_t deviceAlloc(void **out, size_t bytes);
void deviceRelease(void *p);
// mapped, coherent host blocks (a cost's published results), cached per device and size
hipError_t mappedHostAlloc(void **out, size_t bytes);
void mappedHostRelease(int device, void *p, size_t bytes);

int commonCreate(mopt_cost *c, int device);  // device, stream, partial / result buffers
// wait for everything enqueued for this cost, on its own stream and on callers' streams
hipError_t quiesceCost(mopt_cost *c);
void releaseCombine(mopt_cost *c);  // combine.cpp: unmaps / closes whatever was attached
// c_abi.cpp, for the device-resident LM (lm.cpp): upload the static part of this cost's sweep
// constants if it changed and describe the cost to the step kernel; enqueue one resident sweep +
// finalize on `s` (peer-combine sequence numbers = base_sequence + trials counted on the device)
// c_abi.cpp: the next (host_result, flag, sequence) of this cost's mapped host block, and the wait
mopt::HostPublish nextHostPublish(mopt_cost *c, int offset);
int waitHostPublished(mopt_cost *c, unsigned long long sequence);
// `partials_override`: write this cost's partial rows there instead of into its own buffer (the
// rows of several costs behind one another, reduced by one finalize kernel)
int residentPrepare(mopt_cost *c, int jac_mode, hipStream_t s, mopt::LmCostDesc *desc,
                    double *partials_override = nullptr);
int residentGrid(const mopt_cost *c, int jac_mode);      // workgroups (= partial rows) of a resident sweep
int residentDenseRow(const mopt_cost *c, int jac_mode);  // values per partial row; 0: rows of moments
// point2point forward differences under AUTO / MOMENTS: moments or literal, chosen per evaluated point
bool residentPerIterate(const mopt_cost *c, int jac_mode);
// set by mopt_icp_create around the point2point cost it builds on: a cost with a correspondence search
// never takes the direct dispatch path, so its creation does not warm a queue (aql.hpp aqlWarm)
extern thread_local bool g_creating_for_search;
// one finalize (+ LM step) over `rows` rows of `row_length` values in last->d_partials
int residentFinalizeMerged(mopt_cost *last, int rows, int row_length, mopt::LmControl *control,
                           hipStream_t s, const mopt::LmProblem *step, int own_index);
// `step` non-NULL: this is the last cost of the problem, its finalize kernel also runs the LM step
// the resident sweeps of all costs in one launch, where a kernel for that exists (reprojection
// costs with one covariance form); first_row[k] = where cost k's partial rows start
bool residentSetSupported(mopt_cost *const *costs, int num_costs, const int *jac_modes);
int residentSweepSet(mopt_cost *const *costs, int num_costs, const int *jac_modes,
                     const int *first_row, mopt::LmControl *control, hipStream_t s);
int residentSweep(mopt_cost *c, int jac_mode, mopt::LmControl *control, hipStream_t s,
                  unsigned long long base_sequence, const mopt::LmProblem *step = nullptr,
                  int own_index = 0, bool finalize = true);
void releaseResident(mopt_cost *c);
// icp.cpp: the pose-independent part of a correspondence search over this cost's grid
template <typename S>
void fillIcpArgs(const mopt_cost *c, mopt::IcpMatchArgs<S> &a);
void destroyCost(mopt_cost *c);
// enqueue one linearization / cost sweep + its finalize on `s`; results to d_result (+ optional
// hand-over to mapped host memory)
int linearizeAsyncImpl(mopt_cost *c, int jac_mode, const void *x, double *d_result, hipStream_t s,
                       const mopt::HostPublish &pub = mopt::HostPublish());
int costAsyncImpl(mopt_cost *c, const void *x, double *d_sum, hipStream_t s,
                  const mopt::HostPublish &pub = mopt::HostPublish());
// H | b | sum_sq as doubles -> the caller's arrays in the cost's scalar type
void storeResult(const mopt_cost *c, const double *res, void *hessian, void *b, void *sum_sq);

// Back to the device pool on scope exit, after whatever was queued on `stream` — the stream the
// block was used on — has run (an early return may leave work behind; on the normal path the
// stream has been synchronised already and this costs nothing).  Only that stream: a device-wide
// synchronisation here stalled every other cost sweeping on the GPU, five times per ICP cost made.
struct DeviceScratch {
  void *p = nullptr;
  hipStream_t stream = nullptr;
  explicit DeviceScratch(hipStream_t used_on = nullptr) : stream(used_on) {}
  DeviceScratch(const DeviceScratch &) = delete;
  DeviceScratch &operator=(const DeviceScratch &) = delete;
  ~DeviceScratch() {
    if (!p) return;
    (void)hipStreamSynchronize(stream);
    deviceRelease(p);
  }
  hipError_t alloc(size_t bytes) { return deviceAlloc(&p, bytes ? bytes : 16); }
  template <typename T>
  T *as() const { return static_cast<T *>(p); }
};

}  // namespace mopt_detail
