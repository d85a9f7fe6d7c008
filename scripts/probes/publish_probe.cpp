// What does handing a finalize kernel's 43 results to a polling host thread cost, and would values
// that validate themselves be cheaper than payload + flag?
//   flag    the library's form (cdna_hip_programming.md guideline 16, R1): write-through stores of the
//           43 doubles, s_waitcnt vmcnt(0) in every storing wave, barrier, then the sequence word;
//           the host polls the word
//   tagged  every value goes out as one aligned 16-byte store {value, sequence}: no wait, no barrier,
//           no flag; the host polls the 43 tags (a tag that shows the new sequence arrived in the same
//           16-byte write as its value)
//   none    results to device memory only (what the kernel costs without the hand-over)
//   fused   ONE launch: every writer workgroup stores its row write-through (agent scope), drains it,
//           takes a ticket (agent-scope atomic); the workgroup that draws the last ticket adds the rows
//           (agent-scope loads) and hands over in the flag form — the sweep's last-arriving workgroup as
//           the finalize kernel
// Each iteration: a 256-workgroup writer grid (a sweep's epilogue), then the one-workgroup kernel that
// adds the 256 rows and hands over; the host measures launch -> results seen.  Run plain for the host
// times and under rocprofv3 --kernel-trace --stats for the kernel durations.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/publish_probe.cpp -o /tmp/publish_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(e)                                                  \
  do {                                                            \
    hipError_t r_ = (e);                                          \
    if (r_ != hipSuccess) {                                       \
      std::printf("%s: %s\n", #e, hipGetErrorString(r_));         \
      return 1;                                                   \
    }                                                             \
  } while (0)

constexpr int kRows = 256, kCols = 23, kResults = 43, kThreads = 1024;

struct Tagged {
  double value;
  unsigned long long sequence;
};

__global__ void writerKernel(double *rows, double seed) {
  if (threadIdx.x < kCols) rows[blockIdx.x * kCols + threadIdx.x] = seed + blockIdx.x + threadIdx.x;
}

__device__ __forceinline__ double columnSum(const double *rows, double (&scratch)[kThreads],
                                            double (&total)[kCols]) {
  constexpr int per_col = kThreads / kCols, stride = per_col * kCols;
  constexpr int kLoads = (kRows * kCols + stride - 1) / stride;
  const int t = threadIdx.x;
  double s = 0.0;
  if (t < stride) {
    double v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int idx = t + k * stride;
      v[k] = rows[idx < kRows * kCols ? idx : t];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kLoads; ++k) s += (t + k * stride) < kRows * kCols ? v[k] : 0.0;
  }
  scratch[t] = s;
  __syncthreads();
  if (t < kCols) {
    double sum = 0.0;
#pragma unroll
    for (int g = 0; g < per_col; ++g) sum += scratch[t + g * kCols];
    total[t] = sum;
  }
  __syncthreads();
  return t < kResults ? total[t % kCols] + double(t / kCols) : 0.0;  // 43 "results"
}

template <int MODE>  // 0 none, 1 flag, 2 tagged
__global__ __launch_bounds__(kThreads) void handOverKernel(const double *rows, double *device_out,
                                                           double *host_values,
                                                           unsigned long long *host_flag,
                                                           Tagged *host_tagged,
                                                           unsigned long long sequence) {
  __shared__ double scratch[kThreads];
  __shared__ double total[kCols];
  const double v = columnSum(rows, scratch, total);
  const int t = threadIdx.x;
  if (t < kResults) device_out[t] = v;
  if constexpr (MODE == 1) {
    if (t < kResults)
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_values + t),
                         static_cast<unsigned long long>(__double_as_longlong(v)), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0)
      __hip_atomic_store(host_flag, sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if constexpr (MODE == 2) {
    if (t < kResults) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(v));
      const u32x4 data = {static_cast<unsigned int>(bits), static_cast<unsigned int>(bits >> 32),
                          static_cast<unsigned int>(sequence), static_cast<unsigned int>(sequence >> 32)};
      Tagged *dst = host_tagged + t;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(data) : "memory");
    }
  }
}


// the fused form: writer grid of 256-thread workgroups, the last one to arrive finalizes
__global__ __launch_bounds__(256) void fusedKernel(double *rows, double seed, unsigned int *ticket,
                                                   double *device_out, double *host_values,
                                                   unsigned long long *host_flag,
                                                   unsigned long long sequence) {
  __shared__ unsigned int drawn;
  const int t = threadIdx.x;
  if (t < kCols)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(rows + blockIdx.x * kCols + t),
                       static_cast<unsigned long long>(__double_as_longlong(seed + blockIdx.x + t)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (t == 0) drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (drawn != gridDim.x - 1) return;
  if (t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
  // 256 threads: 11 per column, 253 active, 24 rows each, all loads issued together
  constexpr int per_col = 256 / kCols, stride = per_col * kCols, kLoads = (kRows * kCols + stride - 1) / stride;
  __shared__ double scratch[256];
  __shared__ double total[kCols];
  double s = 0.0;
  if (t < stride) {
    unsigned long long v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const int idx = t + k * stride;
      v[k] = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(rows + (idx < kRows * kCols ? idx : t)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kLoads; ++k)
      s += (t + k * stride) < kRows * kCols ? __longlong_as_double(static_cast<long long>(v[k])) : 0.0;
  }
  scratch[t] = s;
  __syncthreads();
  if (t < kCols) {
    double sum = 0.0;
#pragma unroll
    for (int g = 0; g < per_col; ++g) sum += scratch[t + g * kCols];
    total[t] = sum;
  }
  __syncthreads();
  const double v = t < kResults ? total[t % kCols] + double(t / kCols) : 0.0;
  if (t < kResults) {
    device_out[t] = v;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_values + t),
                       static_cast<unsigned long long>(__double_as_longlong(v)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) __hip_atomic_store(host_flag, sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double median(std::vector<double> &v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  double *rows, *device_out;
  CHECK(hipMalloc(&rows, sizeof(double) * kRows * kCols));
  CHECK(hipMalloc(&device_out, sizeof(double) * 64));
  void *host_block = nullptr;
  CHECK(hipHostMalloc(&host_block, 4096, hipHostMallocMapped | hipHostMallocCoherent));
  double *host_values = static_cast<double *>(host_block);                                    // 43 doubles
  unsigned long long *host_flag = reinterpret_cast<unsigned long long *>(host_values + 64);   // own line
  Tagged *host_tagged = reinterpret_cast<Tagged *>(static_cast<char *>(host_block) + 1024);   // 43 x 16 B
  for (int k = 0; k < 64; ++k) host_values[k] = 0.0;
  *host_flag = 0;
  for (int k = 0; k < kResults; ++k) host_tagged[k] = Tagged{0.0, 0ull};
  void *dev_values, *dev_flag, *dev_tagged;
  CHECK(hipHostGetDevicePointer(&dev_values, host_values, 0));
  CHECK(hipHostGetDevicePointer(&dev_flag, host_flag, 0));
  CHECK(hipHostGetDevicePointer(&dev_tagged, host_tagged, 0));

  unsigned int *ticket;
  CHECK(hipMalloc(&ticket, sizeof(unsigned int)));
  CHECK(hipMemset(ticket, 0, sizeof(unsigned int)));
  const int reps = 3000;
  unsigned long long sequence = 0;
  int bad = 0;
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<double> us;
    for (int i = 0; i < reps; ++i) {
      ++sequence;
      const double seed = double(i);
      const auto t0 = std::chrono::steady_clock::now();
      if (mode != 3) writerKernel<<<kRows, 64, 0, stream>>>(rows, seed);
      double got0 = 0.0, got42 = 0.0;
      if (mode == 3) {
        fusedKernel<<<kRows, 256, 0, stream>>>(rows, seed, ticket, device_out, static_cast<double *>(dev_values),
                                               static_cast<unsigned long long *>(dev_flag), sequence);
        while (__atomic_load_n(host_flag, __ATOMIC_ACQUIRE) != sequence) {
        }
        got0 = host_values[0];
        got42 = host_values[42];
      } else if (mode == 0) {
        handOverKernel<0><<<1, kThreads, 0, stream>>>(rows, device_out, nullptr, nullptr, nullptr, sequence);
        CHECK(hipStreamSynchronize(stream));
      } else if (mode == 1) {
        handOverKernel<1><<<1, kThreads, 0, stream>>>(rows, device_out, static_cast<double *>(dev_values),
                                                      static_cast<unsigned long long *>(dev_flag), nullptr,
                                                      sequence);
        while (__atomic_load_n(host_flag, __ATOMIC_ACQUIRE) != sequence) {
        }
        got0 = host_values[0];
        got42 = host_values[42];
      } else {
        handOverKernel<2><<<1, kThreads, 0, stream>>>(rows, device_out, nullptr, nullptr,
                                                      static_cast<Tagged *>(dev_tagged), sequence);
        for (int k = kResults - 1; k >= 0; --k)
          while (__atomic_load_n(&host_tagged[k].sequence, __ATOMIC_ACQUIRE) != sequence) {
          }
        got0 = host_tagged[0].value;
        got42 = host_tagged[42].value;
      }
      const auto t1 = std::chrono::steady_clock::now();
      us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
      if (mode != 0) {
        const double expect0 = 256.0 * seed + 255.0 * 128.0;         // column 0
        const double expect42 = expect0 + 256.0 * (42 % kCols) + 1.0;  // column 19, + 42 / 23
        if (got0 != expect0 || got42 != expect42) ++bad;
      }
      if (mode != 0) CHECK(hipStreamSynchronize(stream));  // next iteration starts from an idle stream
    }
    std::vector<double> tail(us.begin() + reps / 10, us.end());
    std::printf("%-6s launch -> results seen by the host: median %.2f us (min %.2f)\n",
                mode == 0 ? "none*" : (mode == 1 ? "flag" : (mode == 2 ? "tagged" : "fused")), median(tail),
                *std::min_element(tail.begin(), tail.end()));
  }
  std::printf("(* none: launch -> hipStreamSynchronize returns)\nwrong values seen: %d\n", bad);
  return bad ? 1 : 0;
}
