// Single-process multi-GPU group (mopt_group_*): contiguous shards of one correspondence set on
// several devices, combined with one RCCL all-reduce per sweep.
#include "cost_state.hpp"

#include <new>

using namespace mopt_detail;

struct mopt_group {
  std::vector<mopt_cost *> shards;
  std::vector<ncclComm_t> comms;
  int scalar_bytes = 8;
  // RCCL needs distinct devices.  A device list with repeats (several shards on one GPU: a
  // rehearsal of the sharding on a smaller machine) combines the shard sums on the host instead.
  bool host_combine = false;
};

namespace {
void destroyGroup(mopt_group *g) {
  if (!g) return;
  for (auto comm : g->comms)
    if (comm) ncclCommDestroy(comm);
  for (auto *c : g->shards) destroyCost(c);
  delete g;
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
  for (int a = 0; a < num_devices; ++a)
    for (int b = a + 1; b < num_devices; ++b)
      if (devices[a] == devices[b]) g->host_combine = true;
  if (num_devices > 1 && !g->host_combine) {
    g->comms.assign(num_devices, nullptr);
    MOPT_NCCL_TRY(ncclCommInitAll(g->comms.data(), num_devices, devices));
  }
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

static int groupReduceAndFetch(mopt_group *g, int offset, int n_doubles) {
  const int G = int(g->shards.size());
  if (g->host_combine) {
    for (int k = 0; k < G; ++k) {
      mopt_cost *c = g->shards[k];
      MOPT_HIP_TRY(hipSetDevice(c->device));
      MOPT_HIP_TRY(hipMemcpyAsync(c->h_result + offset, c->d_result + offset,
                                  n_doubles * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    for (int k = 0; k < G; ++k) {
      MOPT_HIP_TRY(hipSetDevice(g->shards[k]->device));
      MOPT_HIP_TRY(hipStreamSynchronize(g->shards[k]->stream));
    }
    double *total = g->shards[0]->h_result + offset;
    for (int k = 1; k < G; ++k)  // shard order: deterministic
      for (int q = 0; q < n_doubles; ++q) total[q] += g->shards[k]->h_result[offset + q];
    return MOPT_OK;
  }
  if (G > 1) {
    // one all-reduce per sweep over xGMI; every rank ends with the full sums
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
  }
  mopt_cost *c0 = g->shards[0];
  MOPT_HIP_TRY(hipSetDevice(c0->device));
  MOPT_HIP_TRY(hipMemcpyAsync(c0->h_result + offset, c0->d_result + offset,
                              n_doubles * sizeof(double), hipMemcpyDeviceToHost, c0->stream));
  for (int k = G - 1; k >= 0; --k) {
    MOPT_HIP_TRY(hipSetDevice(g->shards[k]->device));
    MOPT_HIP_TRY(hipStreamSynchronize(g->shards[k]->stream));
  }
  return MOPT_OK;
}

int mopt_group_linearize(mopt_group *g, int jacobian_mode, const void *x, void *hessian, void *b,
                         void *sum_sq) {
  if (!g || !x) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  for (auto *c : g->shards) {
    MOPT_HIP_TRY(hipSetDevice(c->device));
    const int rc = linearizeAsyncImpl(c, jacobian_mode, x, c->d_result, c->stream);
    if (rc != MOPT_OK) return rc;
  }
  const int rc = groupReduceAndFetch(g, 0, kResultDoubles);
  if (rc != MOPT_OK) return rc;
  storeResult(g->shards[0], g->shards[0]->h_result, hessian, b, sum_sq);
  return MOPT_OK;
}

int mopt_group_compute(mopt_group *g, const void *x, void *sum_sq) {
  if (!g || !x || !sum_sq) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  for (auto *c : g->shards) {
    MOPT_HIP_TRY(hipSetDevice(c->device));
    const int rc = costAsyncImpl(c, x, c->d_result + 42, c->stream);
    if (rc != MOPT_OK) return rc;
  }
  const int rc = groupReduceAndFetch(g, 42, 1);
  if (rc != MOPT_OK) return rc;
  storeResult(g->shards[0], g->shards[0]->h_result, nullptr, nullptr, sum_sq);
  return MOPT_OK;
}

}  // extern "C"
