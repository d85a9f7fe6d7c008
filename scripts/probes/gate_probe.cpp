// How fast can a kernel that is ALREADY resident be released by the host, against launching it?
// (pre-queue the sweep gated on a word the host stores, instead of paying the
// launch path per blocking call.)  Measures, host store -> device sees it -> device's write-through
// flag -> host sees it, for a gate word in (a) mapped pinned host memory polled over PCIe and
// (b) fine-grained device memory stored to by the host through the BAR (if this system maps it),
// and the same round trip for a plain launch of a kernel that only raises the flag.
//   hipcc -O2 --offload-arch=gfx950 scripts/probes/gate_probe.cpp -o /tmp/gate_probe && /tmp/gate_probe
// Every device-side wait is bounded (50 ms of the 100 MHz wall clock).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(e)                                                                       \
  do {                                                                                 \
    hipError_t r_ = (e);                                                               \
    if (r_ != hipSuccess) {                                                            \
      std::printf("%s: %s\n", #e, hipGetErrorString(r_));                              \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

__global__ void gated(const unsigned long long *gate, unsigned long long expect,
                      unsigned long long *host_flag, unsigned long long timeout_ticks) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long seen = 0;
    while ((seen = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != expect) {
      if (wall_clock64() - t0 > timeout_ticks) break;
      __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(host_flag, seen == expect ? expect : ~0ull, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// A whole grid (one workgroup per CU, as the sweeps launch): every workgroup waits for the gate line
// itself — lanes 0..12 read the word and a 96-byte payload that sit in one 128-byte line of host
// memory — and the LAST workgroup to get going raises the host flag.
__global__ void gatedGrid(const unsigned long long *gate_line, unsigned long long expect,
                          unsigned int *arrivals, unsigned long long *host_flag,
                          unsigned long long timeout_ticks, double *sink) {
  __shared__ unsigned long long seen_s;
  if (threadIdx.x < 64) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long v = 0;
    bool ok = false;
    while (!ok) {
      if (threadIdx.x < 13)
        v = __hip_atomic_load(gate_line + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long word = __shfl(v, 0, 64);
      ok = word == expect;
      if (wall_clock64() - t0 > timeout_ticks) break;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) seen_s = ok ? expect : ~0ull;
    if (threadIdx.x >= 1 && threadIdx.x < 13 && sink) sink[blockIdx.x * 16 + threadIdx.x] = double(v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int n = atomicAdd(arrivals, 1u);
    if (n + 1 == gridDim.x) {
      *arrivals = 0;
      __hip_atomic_store(host_flag, seen_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The same with ONE poller: workgroup 0 waits for the gate line in host memory and relays word and
// payload through a line of uncached device memory that the other workgroups poll.
__global__ void relayedGrid(const unsigned long long *gate_line, unsigned long long *relay,
                            unsigned long long expect, unsigned int *arrivals,
                            unsigned long long *host_flag, unsigned long long timeout_ticks,
                            double *sink) {
  __shared__ unsigned long long seen_s;
  if (threadIdx.x < 64) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long v = 0;
    bool ok = false;
    if (blockIdx.x == 0) {
      while (!ok) {
        if (threadIdx.x < 13)
          v = __hip_atomic_load(gate_line + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = __shfl(v, 0, 64) == expect;
        if (wall_clock64() - t0 > timeout_ticks) break;
        if (!ok) __builtin_amdgcn_s_sleep(2);
      }
      if (threadIdx.x >= 1 && threadIdx.x < 13)
        __hip_atomic_store(relay + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0)
        __hip_atomic_store(relay, ok ? expect : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      unsigned long long word = 0;
      while (true) {
        if (threadIdx.x == 0)
          word = __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        word = __shfl(word, 0, 64);
        if (word == expect || word == ~0ull) break;
        if (wall_clock64() - t0 > 2 * timeout_ticks) break;
        __builtin_amdgcn_s_sleep(4);
      }
      ok = word == expect;
      if (ok && threadIdx.x >= 1 && threadIdx.x < 13)
        v = __hip_atomic_load(relay + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) seen_s = ok ? expect : ~0ull;
    if (threadIdx.x >= 1 && threadIdx.x < 13 && sink) sink[blockIdx.x * 16 + threadIdx.x] = double(v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int n = atomicAdd(arrivals, 1u);
    if (n + 1 == gridDim.x) {
      *arrivals = 0;
      __hip_atomic_store(host_flag, seen_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void plainGrid(unsigned long long value, unsigned int *arrivals,
                          unsigned long long *host_flag) {
  if (threadIdx.x == 0) {
    const unsigned int n = atomicAdd(arrivals, 1u);
    if (n + 1 == gridDim.x) {
      *arrivals = 0;
      __hip_atomic_store(host_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void plain(unsigned long long value, unsigned long long *host_flag) {
  if (threadIdx.x == 0)
    __hip_atomic_store(host_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static sigjmp_buf g_jump;
static void onSegv(int) { siglongjmp(g_jump, 1); }

static double now() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void spinUs(double us) {
  const double t0 = now();
  while (now() - t0 < us) {
  }
}
static void report(const char *what, std::vector<double> &v) {
  std::sort(v.begin(), v.end());
  std::printf("%-72s median %6.2f us  p10 %6.2f  p90 %6.2f\n", what, v[v.size() / 2], v[v.size() / 10],
              v[v.size() * 9 / 10]);
}

int main() {
  const int iters = 2000;
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long *flag = nullptr, *flag_dev = nullptr;
  CHECK(hipHostMalloc(reinterpret_cast<void **>(&flag), 128, hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&flag_dev), flag, 0));
  *flag = 0;
  const unsigned long long timeout = 100000ull * 50;  // 50 ms

  // ---- plain launch: launch call -> flag on the host
  std::vector<double> t_plain, t_plain_api;
  for (int i = 1; i <= iters; ++i) {
    spinUs(15.0);
    const double t0 = now();
    hipLaunchKernelGGL(plain, dim3(1), dim3(64), 0, s, (unsigned long long)i, flag_dev);
    const double t1 = now();
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned long long)i) {
    }
    t_plain.push_back(now() - t0);
    t_plain_api.push_back(t1 - t0);
  }
  report("plain launch: hipLaunchKernelGGL -> flag seen by the host", t_plain);
  report("   of which the launch call itself", t_plain_api);
  // two dependent launches, as a blocking sweep issues them (sweep + finalize)
  std::vector<double> t_two;
  for (int i = 1; i <= iters; ++i) {
    spinUs(15.0);
    const double t0 = now();
    hipLaunchKernelGGL(plain, dim3(256), dim3(256), 0, s, 0ull, flag_dev + 8);
    hipLaunchKernelGGL(plain, dim3(1), dim3(1024), 0, s, (unsigned long long)(iters + i), flag_dev);
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned long long)(iters + i)) {
    }
    t_two.push_back(now() - t0);
  }
  report("two dependent launches (256 x 256 then 1 x 1024) -> flag seen", t_two);

  // ---- gate in mapped host memory
  unsigned long long *gate_h = nullptr, *gate_h_dev = nullptr;
  CHECK(hipHostMalloc(reinterpret_cast<void **>(&gate_h), 128, hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&gate_h_dev), gate_h, 0));
  *gate_h = 0;
  std::vector<double> t_host;
  int timeouts = 0;
  for (int i = 1; i <= iters; ++i) {
    const unsigned long long v = 2ull * iters + i;
    hipLaunchKernelGGL(gated, dim3(1), dim3(64), 0, s, gate_h_dev, v, flag_dev, timeout);
    spinUs(25.0);  // resident and polling by now
    const double t0 = now();
    __atomic_store_n(gate_h, v, __ATOMIC_RELEASE);
    unsigned long long got;
    while ((got = __atomic_load_n(flag, __ATOMIC_ACQUIRE)) != v && got != ~0ull) {
    }
    t_host.push_back(now() - t0);
    if (got == ~0ull) ++timeouts;
  }
  report("resident kernel, gate word in mapped HOST memory (polled over PCIe)", t_host);
  if (timeouts) std::printf("   (%d timeouts)\n", timeouts);

  // ---- a whole grid: plain launch against a resident grid released by the gate line
  {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned int *arrivals = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&arrivals), 64));
    CHECK(hipMemset(arrivals, 0, 64));
    double *sink = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&sink), size_t(cus) * 16 * 8));
    CHECK(hipDeviceSynchronize());
    std::vector<double> t_pg, t_gg;
    for (int i = 1; i <= iters; ++i) {
      const unsigned long long v = 6ull * iters + i;
      spinUs(15.0);
      const double t0 = now();
      hipLaunchKernelGGL(plainGrid, dim3(cus), dim3(256), 0, s, v, arrivals, flag_dev);
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != v) {
      }
      t_pg.push_back(now() - t0);
    }
    report("plain launch of a CUs x 256 grid -> last workgroup's flag seen", t_pg);
    unsigned long long *line = nullptr, *line_dev = nullptr;
    CHECK(hipHostMalloc(reinterpret_cast<void **>(&line), 256, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&line_dev), line, 0));
    std::memset(line, 0, 256);
    int timeouts = 0;
    for (int i = 1; i <= iters; ++i) {
      const unsigned long long v = 8ull * iters + i;
      hipLaunchKernelGGL(gatedGrid, dim3(cus), dim3(256), 0, s, line_dev, v, arrivals, flag_dev, timeout, sink);
      spinUs(30.0);
      const double t0 = now();
      for (int k = 1; k < 13; ++k) line[k] = v + k;  // payload first, the word last
      __atomic_store_n(line, v, __ATOMIC_RELEASE);
      unsigned long long got;
      while ((got = __atomic_load_n(flag, __ATOMIC_ACQUIRE)) != v && got != ~0ull) {
      }
      t_gg.push_back(now() - t0);
      if (got == ~0ull) ++timeouts;
    }
    report("resident CUs x 256 grid, every workgroup polling the gate line in host memory", t_gg);
    if (timeouts) std::printf("   (%d timeouts)\n", timeouts);
    unsigned long long *relay = nullptr;
    hipError_t er = hipExtMallocWithFlags(reinterpret_cast<void **>(&relay), 256, hipDeviceMallocUncached);
    if (er == hipSuccess) {
      CHECK(hipMemset(relay, 0, 256));
      CHECK(hipDeviceSynchronize());
      std::vector<double> t_rg;
      timeouts = 0;
      for (int i = 1; i <= iters; ++i) {
        const unsigned long long v = 10ull * iters + i;
        hipLaunchKernelGGL(relayedGrid, dim3(cus), dim3(256), 0, s, line_dev, relay, v, arrivals, flag_dev,
                           timeout, sink);
        spinUs(30.0);
        const double t0 = now();
        for (int k = 1; k < 13; ++k) line[k] = v + k;
        __atomic_store_n(line, v, __ATOMIC_RELEASE);
        unsigned long long got;
        while ((got = __atomic_load_n(flag, __ATOMIC_ACQUIRE)) != v && got != ~0ull) {
        }
        t_rg.push_back(now() - t0);
        if (got == ~0ull) ++timeouts;
      }
      report("resident CUs x 256 grid, workgroup 0 polls the host line and relays through uncached HBM", t_rg);
      if (timeouts) std::printf("   (%d timeouts)\n", timeouts);
    }
  }

  // ---- gate in fine-grained DEVICE memory, stored to by the host
  unsigned long long *gate_d = nullptr;
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&gate_d), 128, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    std::printf("fine-grained device memory: %s\n", hipGetErrorString(e));
    return 0;
  }
  CHECK(hipMemset(gate_d, 0, 128));
  CHECK(hipDeviceSynchronize());
  struct sigaction sa, old_segv, old_bus;
  std::memset(&sa, 0, sizeof sa);
  sa.sa_handler = onSegv;
  sigaction(SIGSEGV, &sa, &old_segv);
  sigaction(SIGBUS, &sa, &old_bus);
  bool host_can_store = false;
  if (sigsetjmp(g_jump, 1) == 0) {
    __atomic_store_n(gate_d, 0ull, __ATOMIC_RELEASE);
    host_can_store = true;
  }
  sigaction(SIGSEGV, &old_segv, nullptr);
  sigaction(SIGBUS, &old_bus, nullptr);
  if (!host_can_store) {
    std::printf("fine-grained device memory is not mapped for host stores on this system\n");
    return 0;
  }
  std::vector<double> t_dev;
  timeouts = 0;
  for (int i = 1; i <= 200; ++i) {
    const unsigned long long v = 4ull * iters + i;
    hipLaunchKernelGGL(gated, dim3(1), dim3(64), 0, s, gate_d, v, flag_dev, timeout);
    spinUs(25.0);
    const double t0 = now();
    __atomic_store_n(gate_d, v, __ATOMIC_RELEASE);
    unsigned long long got;
    while ((got = __atomic_load_n(flag, __ATOMIC_ACQUIRE)) != v && got != ~0ull) {
    }
    t_dev.push_back(now() - t0);
    if (got == ~0ull) ++timeouts;
  }
  report("resident kernel, gate word in fine-grained DEVICE memory (host stores through the BAR)", t_dev);
  if (timeouts) std::printf("   (%d timeouts)\n", timeouts);
  CHECK(hipDeviceSynchronize());
  return 0;
}
