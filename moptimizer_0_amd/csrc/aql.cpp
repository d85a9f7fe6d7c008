// Direct AQL dispatch of the blocking sweeps (aql.hpp says why and what).
#include "aql.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace mopt_detail {

// The code-object-v5 implicit arguments a dispatcher fills in (LLVM AMDGPUUsage, "Code object V5
// metadata": hidden_block_count_*, hidden_group_size_*, hidden_remainder_*, hidden_global_offset_*,
// hidden_grid_dims — in this order, from the first 8-byte boundary behind the explicit arguments).
// tests/test_code_objects.py checks every kernel of the library against this layout.
struct ImplicitArgs {
  uint32_t block_count[3];
  uint16_t group_size[3];
  uint16_t remainder[3];
  unsigned char reserved[16];
  uint64_t global_offset[3];
  uint16_t grid_dims;
};
static_assert(offsetof(ImplicitArgs, group_size) == 12 && offsetof(ImplicitArgs, remainder) == 18 &&
                  offsetof(ImplicitArgs, global_offset) == 40 && offsetof(ImplicitArgs, grid_dims) == 64,
              "code object v5 implicit kernel arguments");

constexpr size_t kArgSlotBytes = 4096;  // >= kAqlMaxExplicitArgs + the 256-byte implicit block
constexpr int kArgSlots = 256;          // a slot comes round again after this many dispatches of its queue
constexpr int kQueuesPerDevice = 2;  // (two linked costs sweep side by side; every queue is a hardware queue of the GPU)
constexpr uint32_t kQueuePackets = 1024;

struct AqlQueue {
  hsa_queue_t *queue = nullptr;
  hsa_signal_t drained{};             // completion signal of the drain's barrier packet
  hsa_signal_t stamped{};             // completion signal of the last timed dispatch
  hsa_agent_t agent{};
  const void *stamp_owner = nullptr;  // whose timed dispatch the signal belongs to (null: none outstanding)
  unsigned char *arg_ring = nullptr;  // device memory, host-writable
  uint64_t dispatched = 0;            // packets written (this library is the only producer)
  std::mutex mutex;                   // a queue is shared by the costs that drew it
  std::atomic<bool> faulted{false};
};

namespace {

struct DeviceState {
  bool tried = false, ok = false;
  hsa_agent_t gpu{}, cpu{};
  hsa_amd_memory_pool_t local_pool{};
  AqlQueue queues[kQueuesPerDevice];
  int users = 0;  // live costs on this device (aqlRetain / aqlRelease)
  std::atomic<unsigned> next{0};
  std::map<const void *, AqlKernel> kernels;  // by host function; the null kernel = looked up, absent
  std::mutex mutex;
};

std::mutex g_mutex;
bool g_hsa_tried = false, g_hsa_ok = false;
hsa_ven_amd_loader_1_03_pfn_t g_loader{};
std::map<int, DeviceState *> g_devices;

bool enabled() {
  static const bool on = [] {
    const char *v = std::getenv("MOPT_AQL");
    if (v && v[0] == '0') return false;
    // A profiler that collects hardware counters per dispatch (rocprofv3 --pmc: ROCPROF_COUNTER_COLLECTION /
    // ROCPROF_COUNTERS; rocprof v1 / v2: ROCP_INPUT) serialises the dispatches of every queue it
    // intercepts with packets and signals of its own; with this library's packets — no completion
    // signal, the host waiting on memory the kernel writes — `rocprofv3 --pmc` stopped making
    // progress (round 5, bench.py under `--pmc FETCH_SIZE`: killed after 7 silent minutes; kernel
    // tracing, `--kernel-trace --stats`, works and is how the direct path's kernels are profiled).
    // Counters are about the kernels, which are the same on either path: stay on the HIP stream.
    const char *cc = std::getenv("ROCPROF_COUNTER_COLLECTION");
    const char *counters = std::getenv("ROCPROF_COUNTERS");
    const char *v1 = std::getenv("ROCP_INPUT");
    if ((cc && cc[0] == '1') || (counters && *counters) || (v1 && *v1)) {
      if (v && v[0] == '2') return true;  // MOPT_AQL=2 forces it
      // said once, on stderr: an evidence run states its dispatch path itself (these variable names are a
      // guess at what the profiler exports; should a later ROCm rename them, the blocking wait's bound is
      // what is left — this line missing from a --pmc run's log is the sign)
      std::fprintf(stderr,
                   "libmoptimizer_hip: a per-dispatch counter profiler is attached (%s): blocking sweeps stay on "
                   "HIP streams, the direct AQL path is off (MOPT_AQL=2 forces it on)\n",
                   (cc && cc[0] == '1') ? "ROCPROF_COUNTER_COLLECTION" : (counters && *counters) ? "ROCPROF_COUNTERS" : "ROCP_INPUT");
      return false;
    }
    return true;
  }();
  return on;
}

bool initHsa() {  // g_mutex held
  if (g_hsa_tried) return g_hsa_ok;
  g_hsa_tried = true;
  if (hsa_init() != HSA_STATUS_SUCCESS) return false;  // reference-counted: the HIP runtime holds one too
  if (hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(g_loader), &g_loader) !=
          HSA_STATUS_SUCCESS ||
      !g_loader.hsa_ven_amd_loader_iterate_executables)
    return false;
  g_hsa_ok = true;
  return true;
}

struct AgentSearch {
  uint32_t want_domain = 0, want_bdf = 0;
  bool found_gpu = false, found_cpu = false;
  hsa_agent_t gpu{}, cpu{};
};

hsa_status_t onAgent(hsa_agent_t agent, void *data) {
  AgentSearch *s = static_cast<AgentSearch *>(data);
  hsa_device_type_t type;
  if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (type == HSA_DEVICE_TYPE_CPU) {
    if (!s->found_cpu) {
      s->cpu = agent;
      s->found_cpu = true;
    }
  } else if (type == HSA_DEVICE_TYPE_GPU && !s->found_gpu) {
    uint32_t bdf = 0, domain = 0;
    hsa_agent_get_info(agent, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_BDFID), &bdf);
    hsa_agent_get_info(agent, static_cast<hsa_agent_info_t>(HSA_AMD_AGENT_INFO_DOMAIN), &domain);
    if (bdf == s->want_bdf && domain == s->want_domain) {
      s->gpu = agent;
      s->found_gpu = true;
    }
  }
  return HSA_STATUS_SUCCESS;
}

hsa_status_t onGpuPool(hsa_amd_memory_pool_t pool, void *data) {
  auto *out = static_cast<std::pair<bool, hsa_amd_memory_pool_t> *>(data);
  if (out->first) return HSA_STATUS_SUCCESS;
  hsa_amd_segment_t segment;
  uint32_t flags = 0;
  bool allocatable = false;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &segment);
  if (segment != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &allocatable);
  if (allocatable && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) {
    out->first = true;
    out->second = pool;
  }
  return HSA_STATUS_SUCCESS;
}

void onQueueError(hsa_status_t status, hsa_queue_t *, void *data) {
  // a faulted kernel, a malformed packet: whoever waits for a result of this queue must give up
  static_cast<AqlQueue *>(data)->faulted.store(true, std::memory_order_release);
  const char *text = nullptr;
  hsa_status_string(status, &text);
  std::fprintf(stderr, "moptimizer_hip: direct-dispatch queue error: %s\n", text ? text : "?");
}

// HIP device ordinal -> its HSA agent (by PCI address: the visible-device variables may reorder
// one runtime's view and not the other's), the CPU agent, the device-local pool.
bool initDevice(int device, DeviceState &d) {
  char bus_id[64] = {0};
  if (hipDeviceGetPCIBusId(bus_id, sizeof bus_id, device) != hipSuccess) return false;
  unsigned domain = 0, bus = 0, dev = 0, fn = 0;
  if (std::sscanf(bus_id, "%x:%x:%x.%x", &domain, &bus, &dev, &fn) != 4) return false;
  AgentSearch search;
  search.want_domain = domain;
  search.want_bdf = (bus << 8) | (dev << 3) | fn;
  if (hsa_iterate_agents(onAgent, &search) != HSA_STATUS_SUCCESS || !search.found_gpu || !search.found_cpu)
    return false;
  d.gpu = search.gpu;
  d.cpu = search.cpu;
  std::pair<bool, hsa_amd_memory_pool_t> pool{false, {}};
  if (hsa_amd_agent_iterate_memory_pools(d.gpu, onGpuPool, &pool) != HSA_STATUS_SUCCESS || !pool.first)
    return false;
  d.local_pool = pool.second;
  return true;
}

bool createQueue(DeviceState &d, AqlQueue &q) {
  if (hsa_queue_create(d.gpu, kQueuePackets, HSA_QUEUE_TYPE_SINGLE, onQueueError, &q, UINT32_MAX, UINT32_MAX,
                       &q.queue) != HSA_STATUS_SUCCESS)
    return false;
  void *ring = nullptr;
  if (hsa_amd_memory_pool_allocate(d.local_pool, kArgSlotBytes * kArgSlots, 0, &ring) != HSA_STATUS_SUCCESS ||
      hsa_amd_agents_allow_access(1, &d.cpu, nullptr, ring) != HSA_STATUS_SUCCESS) {
    // (no host-visible device memory: no large BAR) — kernel arguments in host memory cost more than
    // the fences save; leave the direct path off
    if (ring) hsa_amd_memory_pool_free(ring);
    hsa_queue_destroy(q.queue);
    q.queue = nullptr;
    return false;
  }
  q.arg_ring = static_cast<unsigned char *>(ring);
  if (hsa_signal_create(0, 0, nullptr, &q.drained) != HSA_STATUS_SUCCESS) q.drained.handle = 0;
  q.agent = d.gpu;
  // (timestamps are written for dispatches that carry a completion signal only: the timed ones)
  if (hsa_signal_create(0, 0, nullptr, &q.stamped) != HSA_STATUS_SUCCESS ||
      hsa_amd_profiling_set_profiler_enabled(q.queue, 1) != HSA_STATUS_SUCCESS)
    q.stamped.handle = 0;
  return true;
}

DeviceState *deviceState(int device) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!enabled() || !initHsa()) return nullptr;
  DeviceState *&d = g_devices[device];
  if (!d) d = new DeviceState;
  if (!d->tried) {
    d->tried = true;
    d->ok = initDevice(device, *d);
  }
  return d->ok ? d : nullptr;
}

struct SymbolSearch {
  hsa_agent_t agent;
  std::string name;
  bool found = false;
  AqlKernel kernel;
};

hsa_status_t onExecutable(hsa_executable_t executable, void *data) {
  SymbolSearch *s = static_cast<SymbolSearch *>(data);
  if (s->found) return HSA_STATUS_SUCCESS;
  hsa_executable_symbol_t symbol;
  if (hsa_executable_get_symbol_by_name(executable, s->name.c_str(), &s->agent, &symbol) != HSA_STATUS_SUCCESS)
    return HSA_STATUS_SUCCESS;
  AqlKernel k;
  if (hsa_executable_symbol_get_info(symbol, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) !=
          HSA_STATUS_SUCCESS ||
      hsa_executable_symbol_get_info(symbol, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE,
                                     &k.kernarg_size) != HSA_STATUS_SUCCESS ||
      hsa_executable_symbol_get_info(symbol, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE,
                                     &k.group_size) != HSA_STATUS_SUCCESS ||
      hsa_executable_symbol_get_info(symbol, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE,
                                     &k.private_size) != HSA_STATUS_SUCCESS)
    return HSA_STATUS_SUCCESS;
  s->kernel = k;
  s->found = k.object != 0;
  return HSA_STATUS_SUCCESS;
}

}  // namespace

AqlQueue *aqlAcquireQueue(int device) {
  DeviceState *d = deviceState(device);
  if (!d) return nullptr;
  std::lock_guard<std::mutex> lock(d->mutex);
  const unsigned k = d->next.fetch_add(1) % kQueuesPerDevice;
  AqlQueue &q = d->queues[k];
  if (!q.queue && !q.faulted.load()) {
    if (!createQueue(*d, q)) {
      q.faulted.store(true);  // (not tried again)
      return nullptr;
    }
  }
  return q.queue ? &q : nullptr;
}

bool aqlRetain(int device) {
  DeviceState *d = deviceState(device);
  if (!d) return false;
  std::lock_guard<std::mutex> lock(d->mutex);
  // Counted only: a hardware queue (and its 1 MiB argument ring) is created when a cost that can use
  // the direct path first asks for one (aqlAcquireQueue) — costs with a correspondence search,
  // run-time compiled models and RCCL-combined shards never do, and processes that share a GPU must
  // not pin queues they will not use.  aqlTrim must still know that a cost lives here.
  ++d->users;
  return true;
}

void aqlWarm(int device) {
  const char *shared = std::getenv("MOPT_AQL_SHARDED");
  if (shared && shared[0] == '0') return;  // processes share the GPU: a queue only when a sweep asks
  DeviceState *d = deviceState(device);
  if (!d) return;
  std::lock_guard<std::mutex> lock(d->mutex);
  // the queues the next draws are going to hand out (aqlAcquireQueue: round robin from `next`), as many
  // as costs live on the device, up to the pool's size: a caller that creates its costs first and sweeps
  // afterwards — two reprojection costs of one problem — pays for no queue inside a sweep, and neither
  // does the cost that follows a short-lived one (whose draw moved `next` on)
  const int want = d->users < kQueuesPerDevice ? d->users : kQueuesPerDevice;
  const unsigned first = d->next.load();
  for (int j = 0; j < want; ++j) {
    AqlQueue &q = d->queues[(first + unsigned(j)) % kQueuesPerDevice];
    if (!q.queue && !q.faulted.load() && !createQueue(*d, q)) q.faulted.store(true);
  }
}

void aqlRelease(int device) {
  DeviceState *d = deviceState(device);
  if (!d) return;
  std::lock_guard<std::mutex> lock(d->mutex);
  if (d->users > 0) --d->users;
}

bool aqlTrim(int device) {
  DeviceState *d = deviceState(device);
  if (!d) return true;  // (nothing held)
  std::lock_guard<std::mutex> lock(d->mutex);
  if (d->users > 0) return false;
  // no cost of the device is alive (each drained its own sweeps in its destructor)
  for (AqlQueue &q : d->queues) {
    std::lock_guard<std::mutex> queue_lock(q.mutex);
    if (q.queue) {
      (void)hsa_queue_destroy(q.queue);
      q.queue = nullptr;
    }
    if (q.arg_ring) {
      (void)hsa_amd_memory_pool_free(q.arg_ring);
      q.arg_ring = nullptr;
    }
    if (q.drained.handle) (void)hsa_signal_destroy(q.drained);
    if (q.stamped.handle) (void)hsa_signal_destroy(q.stamped);
    q.drained.handle = q.stamped.handle = 0;
    q.stamp_owner = nullptr;
    q.dispatched = 0;
    q.faulted.store(false);
  }
  return true;
}

const AqlKernel *aqlLookup(int device, const void *host_function) {
  DeviceState *d = deviceState(device);
  if (!d) return nullptr;
  std::lock_guard<std::mutex> lock(d->mutex);
  auto it = d->kernels.find(host_function);
  if (it == d->kernels.end()) {
    AqlKernel kernel;  // object 0 = absent
    // the HIP runtime loads a code object when one of its kernels is first asked for
    hipFuncAttributes attributes;
    int current = -1;
    (void)hipGetDevice(&current);
    if (current != device) (void)hipSetDevice(device);
    if (hipFuncGetAttributes(&attributes, host_function) == hipSuccess) {
      const char *name = hipKernelNameRefByPtr(host_function, nullptr);
      if (name && *name) {
        SymbolSearch search;
        search.agent = d->gpu;
        search.name = std::string(name) + ".kd";
        g_loader.hsa_ven_amd_loader_iterate_executables(onExecutable, &search);
        // scratch would need the queue's scratch machinery; no sweep of this library uses any
        if (search.found && search.kernel.private_size == 0) kernel = search.kernel;
      }
    }
    if (current >= 0 && current != device) (void)hipSetDevice(current);
    (void)hipGetLastError();
    it = d->kernels.emplace(host_function, kernel).first;
  }
  return it->second.object ? &it->second : nullptr;
}

void aqlForgetStamp(AqlQueue *q, const void *owner) {
  if (!q || !owner) return;
  std::lock_guard<std::mutex> lock(q->mutex);
  if (q->stamp_owner == owner) q->stamp_owner = nullptr;  // (after a drain: its dispatch has completed)
}

bool aqlFaulted(const AqlQueue *queue) { return queue && queue->faulted.load(std::memory_order_acquire); }

bool aqlDrain(AqlQueue *q) {
  if (!q || !q->queue || !q->drained.handle) return false;
  std::lock_guard<std::mutex> lock(q->mutex);
  if (q->faulted.load(std::memory_order_acquire)) return false;
  hsa_queue_t *hq = q->queue;
  {
    const auto started = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (q->dispatched - hsa_queue_load_read_index_scacquire(hq) >= hq->size) {
      _mm_pause();
      if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - started > std::chrono::seconds(10)) return false;
    }
  }
  hsa_signal_store_relaxed(q->drained, 1);
  const uint64_t index = q->dispatched++;
  hsa_queue_store_write_index_relaxed(hq, index + 1);
  auto *packet = static_cast<hsa_barrier_and_packet_t *>(hq->base_address) + (index & (hq->size - 1));
  std::memset(reinterpret_cast<unsigned char *>(packet) + sizeof(uint16_t), 0,
              sizeof(*packet) - sizeof(uint16_t));
  packet->completion_signal = q->drained;
  constexpr uint16_t header = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) |
                              (1 << HSA_PACKET_HEADER_BARRIER) |
                              (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                              (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  __atomic_store_n(reinterpret_cast<uint16_t *>(packet), header, __ATOMIC_RELEASE);
  hsa_signal_store_screlease(hq->doorbell_signal, hsa_signal_value_t(index));
  // (the time-out of a signal wait is a hint and the wait may return early: loop; a wedged queue must
  // not wedge the caller: give up after 10 s)
  const auto started = std::chrono::steady_clock::now();
  while (hsa_signal_wait_scacquire(q->drained, HSA_SIGNAL_CONDITION_LT, 1, 1000000, HSA_WAIT_STATE_ACTIVE) >= 1) {
    if (q->faulted.load(std::memory_order_acquire)) return false;
    if (std::chrono::steady_clock::now() - started > std::chrono::seconds(10)) return false;
  }
  return true;
}

double aqlDispatchNanoseconds(AqlQueue *q, const void *timed_for) {
  if (!q || !q->stamped.handle || !timed_for) return -1.0;
  std::lock_guard<std::mutex> lock(q->mutex);
  if (q->stamp_owner != timed_for) return -1.0;
  const auto started = std::chrono::steady_clock::now();
  while (hsa_signal_wait_scacquire(q->stamped, HSA_SIGNAL_CONDITION_LT, 1, 1000000, HSA_WAIT_STATE_ACTIVE) >= 1) {
    if (q->faulted.load(std::memory_order_acquire)) return -1.0;
    if (std::chrono::steady_clock::now() - started > std::chrono::seconds(10)) return -1.0;
  }
  q->stamp_owner = nullptr;
  hsa_amd_profiling_dispatch_time_t t{};
  if (hsa_amd_profiling_get_dispatch_time(q->agent, q->stamped, &t) != HSA_STATUS_SUCCESS || t.end < t.start)
    return -1.0;
  static const double ns_per_tick = [] {
    uint64_t hz = 0;
    if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) != HSA_STATUS_SUCCESS || hz == 0) return 10.0;
    return 1e9 / double(hz);
  }();
  return double(t.end - t.start) * ns_per_tick;
}

bool aqlDispatch(AqlQueue *q, const AqlKernel *kernel, uint32_t grid, uint32_t block, const void *args,
                 size_t args_bytes, const void *timed_for) {
  if (!q || !kernel || q->faulted.load(std::memory_order_acquire)) return false;
  const size_t implicit_at = (args_bytes + 7) & ~size_t(7);
  if (args_bytes > kernel->kernarg_size || kernel->kernarg_size > kArgSlotBytes ||
      implicit_at + sizeof(ImplicitArgs) > kArgSlotBytes)
    return false;
  std::lock_guard<std::mutex> lock(q->mutex);
  hsa_queue_t *hq = q->queue;
  // Room in both rings.  Packets: the packet processor has taken everything up to the read index.
  // Argument blocks: every packet carries the barrier bit, so once the packet processor has taken
  // packet d + 2, packet d has completed and its block may be written again — with fewer than
  // kArgSlots - 2 packets beyond the read index the block about to be reused belongs to such a one.
  // (The blocking sweeps never get near either limit; a wedged queue must not wedge the caller.)
  {
    const uint64_t most = hq->size < uint64_t(kArgSlots - 2) ? hq->size : uint64_t(kArgSlots - 2);
    const auto started = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (q->dispatched - hsa_queue_load_read_index_scacquire(hq) >= most) {
      _mm_pause();
      if ((++spins & 0xfff) == 0 && (q->faulted.load(std::memory_order_acquire) ||
                                     std::chrono::steady_clock::now() - started > std::chrono::seconds(10)))
        return false;
    }
  }
  unsigned char *slot = q->arg_ring + size_t(q->dispatched % kArgSlots) * kArgSlotBytes;
  // kernel arguments: explicit, then the implicit block where the kernel has one
  std::memcpy(slot, args, args_bytes);
  size_t written = args_bytes;
  if (kernel->kernarg_size >= implicit_at + sizeof(ImplicitArgs)) {
    ImplicitArgs hidden;
    std::memset(&hidden, 0, sizeof hidden);
    hidden.block_count[0] = grid;
    hidden.block_count[1] = hidden.block_count[2] = 1;
    hidden.group_size[0] = uint16_t(block);
    hidden.group_size[1] = hidden.group_size[2] = 1;
    hidden.grid_dims = 1;
    if (implicit_at > args_bytes) std::memset(slot + args_bytes, 0, implicit_at - args_bytes);
    std::memcpy(slot + implicit_at, &hidden, sizeof hidden);
    written = implicit_at + sizeof hidden;
  }
  // device memory written through the BAR: posted, write-combined — fence, then read the last byte
  // back, which cannot complete before the writes have (what the HIP runtime does for arguments it
  // keeps in device memory)
  if (written > 0) {
    _mm_sfence();
    volatile unsigned char *last = slot + written - 1;
    _mm_mfence();
    (void)*last;
  }
  const uint64_t index = q->dispatched++;
  hsa_queue_store_write_index_relaxed(hq, index + 1);
  auto *packet = static_cast<hsa_kernel_dispatch_packet_t *>(hq->base_address) + (index & (hq->size - 1));
  packet->workgroup_size_x = uint16_t(block);
  packet->workgroup_size_y = 1;
  packet->workgroup_size_z = 1;
  packet->reserved0 = 0;
  packet->grid_size_x = grid * block;
  packet->grid_size_y = 1;
  packet->grid_size_z = 1;
  packet->private_segment_size = 0;
  packet->group_segment_size = kernel->group_size;
  packet->kernel_object = kernel->object;
  packet->kernarg_address = slot;
  packet->reserved2 = 0;
  packet->completion_signal.handle = 0;
  // One timed dispatch outstanding per queue.  The signal is armed for this one only when nothing is
  // outstanding: no owner, or an owner whose dispatch has completed (signal below 1) and was never read —
  // a cost that was destroyed, or a call that did not collect.  While an earlier timed dispatch is still in
  // flight — the same cost's superseded prefetch, say — this one goes untimed: re-arming would let the
  // earlier packet's completion release the wait and hand out the wrong timestamps.
  if (timed_for && q->stamped.handle &&
      (!q->stamp_owner || hsa_signal_load_relaxed(q->stamped) < 1)) {
    hsa_signal_store_relaxed(q->stamped, 1);
    packet->completion_signal = q->stamped;
    q->stamp_owner = timed_for;
  }
  constexpr uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                              (1 << HSA_PACKET_HEADER_BARRIER) |
                              (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                              (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  constexpr uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  __atomic_store_n(reinterpret_cast<uint32_t *>(packet), uint32_t(header) | (uint32_t(setup) << 16),
                   __ATOMIC_RELEASE);
  hsa_signal_store_screlease(hq->doorbell_signal, hsa_signal_value_t(index));
  return true;
}

}  // namespace mopt_detail
