// Kernels of scripts/probes/aql_probe.cpp, compiled to a gfx950 code object of their own
// (hipcc --genco) so that both the HIP module API and the HSA loader can load the same code.
#include <hip/hip_runtime.h>

// a sweep's stand-in: 256 workgroups, each leaves a row
extern "C" __global__ __launch_bounds__(256) void probeWriter(double *rows, double seed) {
  if (threadIdx.x < 23) rows[blockIdx.x * 23 + threadIdx.x] = seed + blockIdx.x + threadIdx.x;
}

// a finalize's stand-in: one workgroup adds the rows and hands the sum over to mapped host memory
// (write-through stores, drained, then the sequence word: the library's publish form)
extern "C" __global__ __launch_bounds__(256) void probePublish(const double *rows, int num_rows,
                                                                double *host_values,
                                                                unsigned long long *host_flag,
                                                                unsigned long long sequence) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < num_rows * 23; i += 256) s += rows[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(host_values, part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(host_flag, sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the same rows stored write-through at agent scope and drained before the wave ends: no release
// fence needed behind the kernel for a reader on another XCD (with an acquire in front of it)
extern "C" __global__ __launch_bounds__(256) void probeWriterThrough(double *rows, double seed) {
  if (threadIdx.x < 23) {
    __hip_atomic_store(rows + blockIdx.x * 23 + threadIdx.x, seed + blockIdx.x + threadIdx.x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);  // every counter: the stores have been acknowledged
  }
}

// No second kernel: every workgroup hands its row straight to the host — 23 values written through to
// mapped host memory, drained, then the row's tag (the sequence number) — and the host adds the rows.
extern "C" __global__ __launch_bounds__(256) void probeWriterToHost(double *host_rows, double seed,
                                                                    unsigned long long sequence) {
  double *row = host_rows + blockIdx.x * 24;  // 23 values + tag, 192 bytes = 3 cache lines
  if (threadIdx.x < 23) {
    __hip_atomic_store(row + threadIdx.x, seed + blockIdx.x + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(row + 23), sequence, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
