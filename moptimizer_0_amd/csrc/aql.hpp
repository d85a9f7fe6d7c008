// Direct dispatch of the blocking sweeps: AQL packets written by this library into HSA queues of
// its own, instead of hipLaunchKernelGGL on a HIP stream.
//
// Why.  A blocking call (mopt_cost_linearize / mopt_cost_compute) is a chain of latencies — launch,
// sweep, kernel boundary, finalize, hand-over — and the HIP runtime's dispatch packets carry a
// SYSTEM-scope acquire fence: every launch invalidates the GPU's caches against host memory, which a
// sweep does not read (its inputs are resident tiles in HBM and kernel arguments; its results leave
// through explicit write-through stores).  scripts/probes/aql_probe.cpp, MI355X, a 256-workgroup
// kernel + a one-workgroup publishing kernel behind it + the host polling, per round trip:
//   hipModuleLaunchKernel x 2                               17.2 us
//   own queue, acquire SYSTEM / release AGENT on both       17.9 us
//   own queue, acquire AGENT  / release AGENT on both       13.3 us    <- what this file does
//   own queue, no fences (not correct across XCDs)           9.5 us
// (kernel arguments in host memory instead of device memory: 41 us.)
//
// What.  Per device a small pool of user-mode queues (an HSA queue carries a context-save area
// sized for the whole GPU: one per cost would be wasteful — HIP itself multiplexes its streams over
// four); a cost takes one round robin.  The kernel objects are those the HIP runtime has loaded:
// found through the loader extension by the kernel's mangled name (hipKernelNameRefByPtr), so there
// is one copy of the code and no file to ship.  Kernel arguments go into a ring of blocks in device
// memory the host can write (large BAR), explicit arguments packed as the compiler lays them out,
// followed by the code-object-v5 implicit block (block counts, group sizes), made visible with a
// store fence and a read-back of the last byte as the HIP runtime does for device-resident arguments.
// Packets: barrier bit (in order), acquire and release at AGENT scope (a kernel sees what the one
// before it wrote, on whichever XCD it ran), no completion signal (the finalize kernel publishes
// into mapped host memory itself).
//
// Scope.  The blocking sweeps of point2point, reprojection and built-in scalar-model costs, with no
// combine, the host-slot combine or the peer combine, and nothing pending on the cost's HIP stream;
// everything else (asynchronous calls, the device-resident loop, RCCL, costs with a correspondence
// search, run-time compiled models — their kernels live in hipRTC modules) stays on the HIP stream.  MOPT_AQL=0 switches the direct path off; it is off by itself under a profiler that collects
// hardware counters per dispatch (rocprofv3 --pmc), which hung with these packets — kernel tracing works.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace mopt_detail {

struct AqlQueue;  // one user-mode queue + its kernel-argument ring (aql.cpp)

struct AqlKernel {
  uint64_t object = 0;        // kernel descriptor address
  uint32_t kernarg_size = 0;  // explicit + implicit arguments
  uint32_t group_size = 0;    // static LDS
  uint32_t private_size = 0;  // scratch per work-item (must be 0 here)
};

// A queue of `device`'s pool (round robin), or nullptr when the direct path is unavailable.
AqlQueue *aqlAcquireQueue(int device);
// One more / one fewer cost lives on `device`.  Counting only: a queue is created when a cost that can
// use the direct path first asks for one (aqlAcquireQueue, a few milliseconds inside that cost's first
// blocking sweep) — costs that never take the path (a correspondence search, run-time compiled models,
// RCCL) pin no hardware queue.  Queues then stay for the process (a caller that builds one cost per
// outer iteration must not pay for them each time) unless aqlTrim gives them back.  aqlRetain returns
// whether it counted (false where the direct path is off): only then call aqlRelease.
bool aqlRetain(int device);
void aqlRelease(int device);
// Called where a cost of a kind that takes the direct path is created (point2point without a
// correspondence search, reprojection, built-in scalar models): makes sure the queue its first blocking
// sweep is going to draw exists, so that the few milliseconds of creating it fall into the construction
// (which copies and re-lays the data anyway) and not into the first solve.  Nothing under
// MOPT_AQL_SHARDED=0.
void aqlWarm(int device);
// `owner` is going away (after a drain): a timed dispatch of its that nobody read no longer holds the
// queue's profiling signal.
void aqlForgetStamp(AqlQueue *queue, const void *owner);
// Destroys the device's queues if no cost lives on it (true), else leaves them (false): every queue
// is a hardware queue of the GPU, and processes that share one run out of them (MOPT_AQL_SHARDED).
bool aqlTrim(int device);
// The loaded kernel behind a __global__ function of this library, or nullptr.
const AqlKernel *aqlLookup(int device, const void *host_function);
// One dispatch: `grid` workgroups of `block` threads; `args` are the explicit arguments as the
// compiler lays them out (packKernelArgs).  Returns false (nothing queued) when the arguments do
// not fit the kernel's segment or the queue has faulted.
// `timed_for` (a cost, or null): the dispatch carries the queue's profiling signal on that owner's
// behalf; aqlDispatchNanoseconds reads its start and end timestamps afterwards (one timed dispatch at a
// time per queue — the blocking sweeps' use; a request while another is outstanding is not timed).
bool aqlDispatch(AqlQueue *queue, const AqlKernel *kernel, uint32_t grid, uint32_t block,
                 const void *args, size_t args_bytes, const void *timed_for = nullptr);
// Duration of the queue's last timed dispatch as the packet processor stamped it (waits for it to
// complete); a negative value when there is none or the timestamps cannot be read.
double aqlDispatchNanoseconds(AqlQueue *queue, const void *timed_for);
// true once the queue has reported an error (a faulted kernel): waits on it must give up
bool aqlFaulted(const AqlQueue *queue);
// Waits until everything dispatched on the queue so far has completed and its writes are visible
// (a barrier packet with a completion signal): before memory its kernels used goes back to a pool.
bool aqlDrain(AqlQueue *queue);

// Explicit kernel arguments, each at its natural alignment, in declaration order.
constexpr size_t kAqlMaxExplicitArgs = 3072;
template <typename... Args>
inline size_t packKernelArgs(unsigned char (&buffer)[kAqlMaxExplicitArgs], const Args &...args) {
  size_t offset = 0;
  bool fits = true;
  auto put = [&](const void *value, size_t size, size_t align) {
    offset = (offset + align - 1) & ~(align - 1);
    if (offset + size > kAqlMaxExplicitArgs) {
      fits = false;
      return;
    }
    std::memcpy(buffer + offset, value, size);
    offset += size;
  };
  (put(&args, sizeof(Args), alignof(Args)), ...);
  return fits ? offset : 0;
}

// Where a launch helper of the device translation units sends its kernel: the direct path when
// `queue` is set, the HIP stream otherwise.
struct AqlSite {
  AqlQueue *queue = nullptr;
  int device = 0;
  const void *timed_for = nullptr;  // stamp this dispatch (profiling) for this owner
};

// hipLaunchKernelGGL's shape for the direct path; false = not dispatched (the caller falls back to
// the HIP stream).  The arguments are converted to the kernel's own parameter types before they
// are packed, as a call would convert them.
template <typename... Params, typename... Args>
inline bool aqlLaunch(const AqlSite &site, void (*kernel)(Params...), uint32_t grid, uint32_t block,
                      const Args &...args) {
  static_assert(sizeof...(Params) == sizeof...(Args), "one argument per kernel parameter");
  if (!site.queue) return false;
  // a few look-ups kept per signature and device for the life of the process (kernels that share a
  // signature — the moments and the cost sweep, the covariance forms of one sweep — share the
  // table): the map behind aqlLookup costs two mutexes and two searches
  constexpr int kDevices = 16, kWays = 8;
  struct Entry {
    std::atomic<void (*)(Params...)> kernel{nullptr};
    std::atomic<const AqlKernel *> found{nullptr};
  };
  static Entry table[kDevices][kWays];
  const AqlKernel *k = nullptr;
  if (site.device >= 0 && site.device < kDevices) {
    Entry *row = table[site.device];
    int way = 0;
    for (; way < kWays; ++way) {
      void (*have)(Params...) = row[way].kernel.load(std::memory_order_acquire);
      if (have == kernel) {
        k = row[way].found.load(std::memory_order_relaxed);
        break;
      }
      if (!have) break;
    }
    if (!k) {
      k = aqlLookup(site.device, reinterpret_cast<const void *>(kernel));
      if (!k) return false;
      static std::mutex fill;  // entries are filled a handful of times per process
      std::lock_guard<std::mutex> lock(fill);
      for (way = 0; way < kWays; ++way) {
        void (*have)(Params...) = row[way].kernel.load(std::memory_order_relaxed);
        if (have == kernel) break;
        if (!have) {
          row[way].found.store(k, std::memory_order_relaxed);
          row[way].kernel.store(kernel, std::memory_order_release);  // readers take `found` after this
          break;
        }
      }
    }
  } else {
    k = aqlLookup(site.device, reinterpret_cast<const void *>(kernel));
    if (!k) return false;
  }
  unsigned char buffer[kAqlMaxExplicitArgs];
  const size_t bytes =
      packKernelArgs(buffer, static_cast<typename std::decay<Params>::type>(args)...);
  if (bytes == 0 && sizeof...(Args) > 0) return false;
  return aqlDispatch(site.queue, k, grid, block, buffer, bytes, site.timed_for);
}

}  // namespace mopt_detail
