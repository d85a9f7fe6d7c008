// Shard combine without a collective launch (include/moptimizer_hip.h, "latency-optimised shard
// combine"; slot geometry in sweep.hpp).  This file owns the memory the ranks share:
//   MOPT_COMBINE_HOST  one slot block in POSIX shared memory, registered with HIP in every rank;
//   MOPT_COMBINE_PEER  one slot block per rank in uncached device memory, exported / opened as
//                      hipIpcMemHandle (xGMI stores between the GPUs of a node).
// The pushes and waits themselves are in the finalize kernels (sweep_kernels.hip: publishToHost,
// peerCombine) and in blockingSweep (c_abi.cpp).
#include "cost_state.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <mutex>
#include <vector>

using namespace mopt_detail;

static_assert(sizeof(hipIpcMemHandle_t) == MOPT_PEER_HANDLE_BYTES,
              "MOPT_PEER_HANDLE_BYTES must be the size of a hipIpcMemHandle_t");

namespace {

constexpr int kMaxHostRanks = 64;

int checkRanks(const mopt_cost *c, int rank, int num_ranks, int limit) {
  if (num_ranks < 1 || num_ranks > limit || rank < 0 || rank >= num_ranks)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "bad rank / num_ranks for this transport");
  const ShardCombine &sc = c->combine;
  const bool attached = sc.host_block || sc.peer_attached || c->comm;
  if (attached && (sc.rank != rank || sc.num_ranks != num_ranks))
    return fail(MOPT_ERR_INVALID_ARGUMENT,
                "rank / num_ranks differ from the transport already attached to this cost");
  return MOPT_OK;
}

// Slot blocks exported by THIS process.  hipIpcOpenMemHandle refuses a handle of the opening process's
// own allocation, and ranks need not be processes: one process may hold several ranks' costs (a thread
// per GPU, as mopt_group_* does; 8 ranks rehearsed on a box that admits fewer processes).  A handle
// found here is attached through the pointer it was made from.
struct OwnExport {
  hipIpcMemHandle_t handle;
  double *block;
  int device;
};
std::mutex g_own_exports_mutex;
std::vector<OwnExport> g_own_exports;

double *ownExport(const hipIpcMemHandle_t &handle, int *device) {
  std::lock_guard<std::mutex> lock(g_own_exports_mutex);
  for (const OwnExport &e : g_own_exports)
    if (std::memcmp(&e.handle, &handle, sizeof handle) == 0) {
      *device = e.device;
      return e.block;
    }
  return nullptr;
}

void forgetOwnExport(const double *block) {
  std::lock_guard<std::mutex> lock(g_own_exports_mutex);
  for (size_t k = 0; k < g_own_exports.size(); ++k)
    if (g_own_exports[k].block == block) {
      g_own_exports.erase(g_own_exports.begin() + long(k));
      return;
    }
}

unsigned long long peerTimeoutTicks(int device) {
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0)
    khz = 100000;  // gfx9: 100 MHz
  return (unsigned long long)khz * (unsigned long long)envInt("MOPT_PEER_TIMEOUT_MS", 5000);
}

}  // namespace

namespace mopt_detail {

void releaseCombine(mopt_cost *c) {
  ShardCombine &sc = c->combine;
  for (int k = 0; k < mopt::kMaxPeers; ++k) {
    if (sc.peer_opened[k] && sc.peer_blocks[k]) (void)hipIpcCloseMemHandle(sc.peer_blocks[k]);
    sc.peer_opened[k] = false;
    sc.peer_blocks[k] = nullptr;
  }
  if (sc.peer_own) {
    forgetOwnExport(sc.peer_own);
    (void)hipFree(sc.peer_own);
  }
  sc.peer_own = nullptr;
  sc.peer_attached = false;
  if (sc.host_registered) (void)hipHostUnregister(sc.host_block);
  sc.host_registered = false;
  if (sc.host_block) (void)munmap(sc.host_block, sc.host_bytes);
  sc.host_block = sc.host_block_dev = nullptr;
  if (sc.shm_fd >= 0) (void)close(sc.shm_fd);
  sc.shm_fd = -1;
  if (!sc.shm_name.empty()) (void)shm_unlink(sc.shm_name.c_str());  // ENOENT: a peer was first
  sc.shm_name.clear();
  sc.mode = c->comm ? MOPT_COMBINE_RCCL : MOPT_COMBINE_NONE;
}

}  // namespace mopt_detail

extern "C" {

int mopt_cost_hostcomm_attach(mopt_cost *c, const char *shm_name, int rank, int num_ranks) {
  if (!c || !shm_name || shm_name[0] != '/')
    return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL or shm_name does not start with '/'");
  int rc = checkRanks(c, rank, num_ranks, kMaxHostRanks);
  if (rc != MOPT_OK) return rc;
  ShardCombine &sc = c->combine;
  if (sc.host_block) return fail(MOPT_ERR_INVALID_ARGUMENT, "a host slot block is already attached");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  const size_t page = size_t(sysconf(_SC_PAGESIZE));
  const size_t bytes =
      (mopt::slotBlockDoubles(num_ranks) * sizeof(double) + page - 1) / page * page;
  // A fresh object reads as zeros and sequence numbers start at 1, so nobody has to initialise it
  // (and nobody may: a rank that attaches late must not wipe what an early one has published).
  bool created = true;  // by this call: then a failure below also removes the name again
  int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 && errno == EEXIST) {
    created = false;
    fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
  }
  if (fd < 0) return fail(MOPT_ERR_HIP, std::string("shm_open: ") + std::strerror(errno));
  auto abandon = [&](void *mapped, size_t mapped_bytes) {
    if (mapped) (void)munmap(mapped, mapped_bytes);
    (void)close(fd);
    if (created) (void)shm_unlink(shm_name);
  };
  if (ftruncate(fd, off_t(bytes)) != 0) {
    const std::string why = std::strerror(errno);
    abandon(nullptr, 0);
    return fail(MOPT_ERR_HIP, "ftruncate: " + why);
  }
  void *base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (base == MAP_FAILED) {
    const std::string why = std::strerror(errno);
    abandon(nullptr, 0);
    return fail(MOPT_ERR_HIP, "mmap: " + why);
  }
  double *dev = nullptr;
  bool registered = false;
  hipError_t e = hipHostRegister(base, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e == hipSuccess) {
    registered = true;
    e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), base, 0);
  }
  if (e != hipSuccess) {
    // undo what THIS call did and nothing else: a peer block or a communicator attached earlier
    // stays as it is (other ranks may be pushing into this rank's peer block right now), and so
    // does the selected transport.  The name is removed only if this call created the object.
    const std::string why = hipGetErrorString(e);
    (void)hipGetLastError();
    if (registered) (void)hipHostUnregister(base);
    abandon(base, bytes);
    return fail(MOPT_ERR_HIP, "registering the shared slot block with HIP: " + why);
  }
  sc.shm_fd = fd;
  sc.host_block = static_cast<double *>(base);
  sc.host_block_dev = dev;
  sc.host_bytes = bytes;
  sc.host_registered = true;
  sc.shm_name = shm_name;
  sc.rank = rank;
  sc.num_ranks = num_ranks;
  sc.mode = MOPT_COMBINE_HOST;
  c->cache.valid = false;
  return MOPT_OK;
}

int mopt_hostcomm_unlink(const char *shm_name) {
  if (!shm_name) return fail(MOPT_ERR_INVALID_ARGUMENT, "shm_name is NULL");
  if (shm_unlink(shm_name) != 0 && errno != ENOENT)
    return fail(MOPT_ERR_HIP, std::string("shm_unlink: ") + std::strerror(errno));
  return MOPT_OK;
}

int mopt_cost_peer_export(mopt_cost *c, int num_ranks, void *handle_out) {
  if (!c || !handle_out) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  if (num_ranks < 1 || num_ranks > mopt::kMaxPeers)
    return fail(MOPT_ERR_INVALID_ARGUMENT, "the device-side combine spans at most 8 ranks (one node)");
  ShardCombine &sc = c->combine;
  if (sc.peer_own) return fail(MOPT_ERR_INVALID_ARGUMENT, "a peer slot block was already exported");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = mopt::slotBlockDoubles(num_ranks) * sizeof(double);
  // Uncached (MTYPE UC) memory: stores arriving over xGMI go to HBM, not through this device's
  // L2, so the polling finalize kernel must not find the line in a cache either.
  void *p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess)
    return fail(MOPT_ERR_HIP, std::string("uncached slot block: ") + hipGetErrorString(e));
  sc.peer_own = static_cast<double *>(p);
  e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t handle;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&handle, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    sc.peer_own = nullptr;
    return fail(MOPT_ERR_HIP, std::string("exporting the slot block (hipIpcGetMemHandle): ") +
                                  hipGetErrorString(e));
  }
  {
    std::lock_guard<std::mutex> lock(g_own_exports_mutex);
    g_own_exports.push_back({handle, sc.peer_own, c->device});
  }
  std::memcpy(handle_out, &handle, sizeof handle);
  return MOPT_OK;
}

int mopt_cost_peer_attach(mopt_cost *c, const void *handles, int rank, int num_ranks) {
  if (!c || !handles) return fail(MOPT_ERR_INVALID_ARGUMENT, "NULL argument");
  int rc = checkRanks(c, rank, num_ranks, mopt::kMaxPeers);
  if (rc != MOPT_OK) return rc;
  ShardCombine &sc = c->combine;
  if (!sc.peer_own) return fail(MOPT_ERR_INVALID_ARGUMENT, "call mopt_cost_peer_export first");
  if (sc.peer_attached) return fail(MOPT_ERR_INVALID_ARGUMENT, "peers are already attached");
  MOPT_HIP_TRY(hipSetDevice(c->device));
  const char *bytes = static_cast<const char *>(handles);
  // a failure at rank k undoes the mappings made for the ranks before it.  The own block stays
  // allocated: other ranks may have opened its exported handle already (it is released with the
  // cost); mopt_cost_peer_attach may be called again with new handles
  auto undo = [&](int k) {
    for (int q = 0; q < k; ++q) {
      if (sc.peer_opened[q]) (void)hipIpcCloseMemHandle(sc.peer_blocks[q]);
      sc.peer_opened[q] = false;
      sc.peer_blocks[q] = nullptr;
    }
    sc.peer_blocks[rank] = nullptr;
  };
  for (int k = 0; k < num_ranks; ++k) {
    if (k == rank) {
      sc.peer_blocks[k] = sc.peer_own;
      continue;
    }
    hipIpcMemHandle_t handle;
    std::memcpy(&handle, bytes + size_t(k) * sizeof handle, sizeof handle);
    int owner_device = c->device;
    if (double *same_process = ownExport(handle, &owner_device)) {
      // a rank of this very process: no IPC mapping; another GPU's block needs peer access switched
      // on for this device (what hipIpcMemLazyEnablePeerAccess does for an opened handle)
      if (owner_device != c->device) {
        const hipError_t pe = hipDeviceEnablePeerAccess(owner_device, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
          (void)hipGetLastError();
          undo(k);
          return fail(MOPT_ERR_HIP, "peer access to the slot block of rank " + std::to_string(k) +
                                        " (hipDeviceEnablePeerAccess): " + hipGetErrorString(pe));
        }
        (void)hipGetLastError();
      }
      sc.peer_blocks[k] = same_process;
      continue;
    }
    void *p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, handle, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      undo(k);
      return fail(MOPT_ERR_HIP, "opening the slot block of rank " + std::to_string(k) +
                                    " (hipIpcOpenMemHandle): " + hipGetErrorString(e));
    }
    sc.peer_blocks[k] = static_cast<double *>(p);
    sc.peer_opened[k] = true;
  }
  sc.peer_timeout_ticks = peerTimeoutTicks(c->device);
  sc.peer_attached = true;
  sc.rank = rank;
  sc.num_ranks = num_ranks;
  sc.mode = MOPT_COMBINE_PEER;
  c->cache.valid = false;
  return MOPT_OK;
}

int mopt_cost_set_combine(mopt_cost *c, int mode) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  const bool ok = mode == MOPT_COMBINE_NONE || (mode == MOPT_COMBINE_RCCL && c->comm) ||
                  ((mode == MOPT_COMBINE_HOST || mode == MOPT_COMBINE_PEER) && c->combine.has(mode));
  if (!ok) return fail(MOPT_ERR_INVALID_ARGUMENT, "that combine transport is not attached to this cost");
  if (c->combine.mode != mode) c->cache.valid = false;
  c->combine.mode = mode;
  return MOPT_OK;
}

int mopt_cost_get_combine(const mopt_cost *c, int *mode, int *rank, int *num_ranks) {
  if (!c) return fail(MOPT_ERR_INVALID_ARGUMENT, "cost is NULL");
  if (mode) *mode = c->combine.mode;
  if (rank) *rank = c->combine.rank;
  if (num_ranks) *num_ranks = c->combine.num_ranks;
  return MOPT_OK;
}

}  // extern "C"
