// Single-process multi-GPU group (mopt_group_*): contiguous shards of one correspondence set on
// several devices.  The sums of the shards replace the host accumulation
// `hessian_ += cost_hessian_` of levenberg_marquadt_dyn.cpp:57-59 across shards.
//
// Per sweep, each shard's two kernels are launched by a host thread of its own (a launch costs
// the host 3-5 us and hipSetDevice switches are not free either: one thread launching 8 devices
// in turn is host-bound before the GPUs are), every finalize kernel publishes its sums straight
// into that shard's mapped host memory, and the calling thread adds the shards in shard order as
// their sequence words arrive — no collective launch, no copy, no stream synchronisation.
// MOPT_GROUP_COLLECTIVE=rccl keeps the ncclAllReduce form (distinct devices only) for comparison.
#include "cost_state.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

using namespace mopt_detail;

namespace mopt_detail {
// c_abi.cpp: launch one sweep + finalize of a shard, published into its own mapped host memory,
// without waiting; and the wait for that publication
int launchPublishedSweep(mopt_cost *c, bool cost_only, int jac_mode, const void *x,
                         unsigned long long *sequence_out);
int waitPublishedSweep(mopt_cost *c, unsigned long long sequence);
}  // namespace mopt_detail

struct mopt_group {
  std::vector<mopt_cost *> shards;
  std::vector<ncclComm_t> comms;
  int scalar_bytes = 8;
  bool use_rccl = false;

  // launch workers: one per shard beyond the first (the caller's thread launches shard 0)
  struct Job {
    bool cost_only = false;
    int jac_mode = 0;
    unsigned char x[kMaxParamBytes] = {0};
  } job;
  std::vector<std::thread> workers;
  std::vector<int> rc;                         // per shard, of the current job
  std::vector<std::string> error;              // per shard
  std::vector<unsigned long long> sequence;    // per shard: what to wait for
  std::atomic<unsigned long long> generation{0};
  std::atomic<int> outstanding{0};
  std::atomic<int> sleepers{0};
  std::atomic<bool> stop{false};
  std::mutex mutex;
  std::condition_variable wake;
  double total[kResultSlots] = {0};
};

namespace {

void runShardJob(mopt_group *g, int k) {
  mopt_cost *c = g->shards[k];
  int rc = MOPT_OK;
  if (hipSetDevice(c->device) != hipSuccess) rc = fail(MOPT_ERR_HIP, "hipSetDevice failed");
  if (rc == MOPT_OK) {
    if (g->use_rccl) {
      rc = g->job.cost_only
               ? costAsyncImpl(c, g->job.x, c->d_result + (c->n_params * c->n_params + c->n_params),
                               c->stream)
               : linearizeAsyncImpl(c, g->job.jac_mode, g->job.x, c->d_result, c->stream);
    } else {
      rc = launchPublishedSweep(c, g->job.cost_only, g->job.jac_mode, g->job.x, &g->sequence[k]);
    }
  }
  g->rc[k] = rc;
  if (rc != MOPT_OK) g->error[k] = mopt_last_error();  // thread-local text of this worker
}

// Workers spin for a short while after a job (LM calls come back to back, tens of microseconds
// apart) and otherwise sleep on the condition variable.
void workerLoop(mopt_group *g, int k) {
  unsigned long long seen = 0;
  for (;;) {
    unsigned long long now = g->generation.load(std::memory_order_acquire);
    if (now == seen) {
      const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
      while ((now = g->generation.load(std::memory_order_acquire)) == seen &&
             !g->stop.load(std::memory_order_relaxed) &&
             std::chrono::steady_clock::now() < spin_until)
        __builtin_ia32_pause();
      if (now == seen && !g->stop.load(std::memory_order_relaxed)) {
        std::unique_lock<std::mutex> lock(g->mutex);
        g->sleepers.fetch_add(1);
        g->wake.wait(lock, [&] {
          return g->generation.load(std::memory_order_acquire) != seen || g->stop.load();
        });
        g->sleepers.fetch_sub(1);
        now = g->generation.load(std::memory_order_acquire);
      }
    }
    if (g->stop.load()) return;
    if (now == seen) continue;
    seen = now;
    runShardJob(g, k);
    g->outstanding.fetch_sub(1, std::memory_order_release);
  }
}

void destroyGroup(mopt_group *g) {
  if (!g) return;
  {
    std::lock_guard<std::mutex> lock(g->mutex);
    g->stop.store(true);
  }
  g->wake.notify_all();
  for (auto &t : g->workers)
    if (t.joinable()) t.join();
  for (auto comm : g->comms)
    if (comm) ncclCommDestroy(comm);
  for (auto *c : g->shards) destroyCost(c);
  delete g;
}

// Launch the job on every shard (workers for shards 1.., this thread for shard 0).
int launchOnAllShards(mopt_group *g, bool cost_only, int jac_mode, const void *x) {
  const int G = int(g->shards.size());
  g->job.cost_only = cost_only;
  g->job.jac_mode = jac_mode;
  std::memcpy(g->job.x, x, size_t(g->shards[0]->n_params) * g->scalar_bytes);
  for (int k = 0; k < G; ++k) g->rc[k] = MOPT_OK;
  if (G > 1) {
    g->outstanding.store(G - 1, std::memory_order_relaxed);
    {
      // the generation changes under the mutex so that a worker about to sleep cannot miss it
      std::lock_guard<std::mutex> lock(g->mutex);
      g->generation.fetch_add(1, std::memory_order_release);
    }
    if (g->sleepers.load() > 0) g->wake.notify_all();
  }
  runShardJob(g, 0);
  while (g->outstanding.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
  for (int k = 0; k < G; ++k)
    if (g->rc[k] != MOPT_OK) return fail(g->rc[k], g->error[k]);
  return MOPT_OK;
}

int reduceWithRccl(mopt_group *g, int offset, int n_doubles) {
  const int G = int(g->shards.size());
  MOPT_NCCL_TRY(ncclGroupStart());
  for (int k = 0; k < G; ++k) {
    mopt_cost *c = g->shards[k];
    ncclResult_t r = ncclAllReduce(c->d_result + offset, c->d_result + offset, n_doubles,
                                   ncclDouble, ncclSum, g->comms[k], c->stream);
    if (r != ncclSuccess) {
      ncclGroupEnd();
      return fail(MOPT_ERR_RCCL, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    }
  }
  MOPT_NCCL_TRY(ncclGroupEnd());
  mopt_cost *c0 = g->shards[0];
  MOPT_HIP_TRY(hipSetDevice(c0->device));
  MOPT_HIP_TRY(hipMemcpyAsync(g->total + offset, c0->d_result + offset, n_doubles * sizeof(double),
                              hipMemcpyDeviceToHost, c0->stream));
  for (int k = G - 1; k >= 0; --k) {
    MOPT_HIP_TRY(hipSetDevice(g->shards[k]->device));
    MOPT_HIP_TRY(hipStreamSynchronize(g->shards[k]->stream));
  }
  return MOPT_OK;
}

int groupSweep(mopt_group *g, bool cost_only, int jac_mode, const void *x) {
  const int G = int(g->shards.size());
  const mopt_cost *c0 = g->shards[0];
  const int offset = cost_only ? c0->n_params * c0->n_params + c0->n_params : 0;
  const int count = cost_only ? 1 : c0->n_params * c0->n_params + c0->n_params + 1;
  int rc = launchOnAllShards(g, cost_only, jac_mode, x);
  if (rc != MOPT_OK) return rc;
  if (g->use_rccl) return reduceWithRccl(g, offset, count);
  for (int q = 0; q < count; ++q) g->total[offset + q] = 0.0;
  for (int k = 0; k < G; ++k) {  // shard order: deterministic
    rc = waitPublishedSweep(g->shards[k], g->sequence[k]);
    if (rc != MOPT_OK) return rc;
    const double *part = g->shards[k]->h_result + offset;
    for (int q = 0; q < count; ++q) g->total[offset + q] += part[q];
  }
  return MOPT_OK;
}

}  // namespace

extern "C" {

int mopt_group_point2point_create(mopt_group **out, const int *devices, int num_devices,
                                  int scalar_bytes, const void *src_xyz, const void *tgt_xyz,
                                  int64_t count) {
  if (!out) return fail(MOPT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (!devices || num_devices < 1) return fail(MOPT_ERR_INVALID_ARGUMENT, "no devices given");
  if (scalar_bytes != 4 && scalar_bytes != 8)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "scalar_bytes must be 4 or 8");
  std::unique_ptr<mopt_group, void (*)(mopt_group *)> g(new (std::nothrow) mopt_group,
                                                        destroyGroup);
  if (!g) return fail(MOPT_ERR_HIP, "out of host memory");
  g->scalar_bytes = scalar_bytes;
  const char *src = static_cast<const char *>(src_xyz);
  const char *tgt = static_cast<const char *>(tgt_xyz);
  for (int k = 0; k < num_devices; ++k) {
    // contiguous index ranges [k N / G, (k + 1) N / G)
    const int64_t lo = count * k / num_devices, hi = count * (k + 1) / num_devices;
    mopt_cost *shard = nullptr;
    const int rc = mopt_point2point_create(&shard, devices[k], scalar_bytes,
                                           src + size_t(lo) * 3 * scalar_bytes,
                                           tgt + size_t(lo) * 3 * scalar_bytes, hi - lo,
                                           MOPT_INPUT_HOST);
    if (rc != MOPT_OK) return rc;
    g->shards.push_back(shard);
  }
  bool repeats = false;  // RCCL needs distinct devices
  for (int a = 0; a < num_devices; ++a)
    for (int b = a + 1; b < num_devices; ++b)
      if (devices[a] == devices[b]) repeats = true;
  const char *want = std::getenv("MOPT_GROUP_COLLECTIVE");
  if (want && std::strcmp(want, "rccl") == 0 && num_devices > 1 && !repeats) {
    g->comms.assign(num_devices, nullptr);
    MOPT_NCCL_TRY(ncclCommInitAll(g->comms.data(), num_devices, devices));
    g->use_rccl = true;
  }
  g->rc.assign(num_devices, MOPT_OK);
  g->error.assign(num_devices, std::string());
  g->sequence.assign(num_devices, 0);
  for (int k = 1; k < num_devices; ++k) g->workers.emplace_back(workerLoop, g.get(), k);
  *out = g.release();
  return MOPT_OK;
}

int mopt_group_destroy(mopt_group *group) {
  destroyGroup(group);
  return MOPT_OK;
}

int mopt_group_size(const mopt_group *g, int *num_devices) {
  if (!g || !num_devices) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  *num_devices = int(g->shards.size());
  return MOPT_OK;
}

int mopt_group_set_covariance(mopt_group *g, const void *cov_colmajor) {
  if (!g) return fail(MOPT_ERR_INVALID_ARGUMENT, "group is NULL");
  for (auto *c : g->shards) {
    const int rc = mopt_cost_set_covariance(c, cov_colmajor);
    if (rc != MOPT_OK) return rc;
  }
  return MOPT_OK;
}

int mopt_group_set_loss(mopt_group *g, int loss_kind, double parameter) {
  if (!g) return fail(MOPT_ERR_INVALID_ARGUMENT, "group is NULL");
  for (auto *c : g->shards) {
    const int rc = mopt_cost_set_loss(c, loss_kind, parameter);
    if (rc != MOPT_OK) return rc;
  }
  return MOPT_OK;
}

int mopt_group_linearize(mopt_group *g, int jacobian_mode, const void *x, void *hessian, void *b,
                         void *sum_sq) {
  if (!g || !x) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (jacobian_mode < MOPT_JAC_ANALYTIC || jacobian_mode > MOPT_JAC_NUMERIC)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "unknown jacobian_mode");
  const int rc = groupSweep(g, false, jacobian_mode, x);
  if (rc != MOPT_OK) return rc;
  storeResult(g->shards[0], g->total, hessian, b, sum_sq);
  return MOPT_OK;
}

int mopt_group_compute(mopt_group *g, const void *x, void *sum_sq) {
  if (!g || !x || !sum_sq) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  const int rc = groupSweep(g, true, 0, x);
  if (rc != MOPT_OK) return rc;
  storeResult(g->shards[0], g->total, nullptr, nullptr, sum_sq);
  return MOPT_OK;
}

}  // extern "C"
