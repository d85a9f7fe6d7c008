// What does the width of the one-workgroup finalize cost?  A writer grid (256 workgroups, one row of
// 23 doubles each, like a sweep's epilogue) followed by a one-workgroup kernel that adds the 256 rows
// per column, with T = 64 ... 1024 threads, and by an empty kernel of the same widths.  Run under
//   rocprofv3 --kernel-trace --stats -- /tmp/finalize_width_probe
// and read the average duration per kernel name.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/finalize_width_probe.cpp -o /tmp/finalize_width_probe
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(e)                                                  \
  do {                                                            \
    hipError_t r_ = (e);                                          \
    if (r_ != hipSuccess) {                                       \
      std::printf("%s: %s\n", #e, hipGetErrorString(r_));         \
      return 1;                                                   \
    }                                                             \
  } while (0)

constexpr int kRows = 256, kCols = 23;

__global__ void writerKernel(double *rows, double seed) {
  if (threadIdx.x < kCols) rows[blockIdx.x * kCols + threadIdx.x] = seed + blockIdx.x + threadIdx.x;
}

template <int T>
__global__ __launch_bounds__(T) void emptyKernel(double *out) {
  if (threadIdx.x == 0 && out == nullptr) out[0] = 1.0;
}

// per_col threads per column, each adding rows r, r + per_col, ...; all loads of a thread issued
// before the first use; partial sums through LDS; thread c < kCols adds its column's per_col values
template <int T>
__global__ __launch_bounds__(T) void columnSumKernel(const double *rows, double *out) {
  constexpr int per_col = T / kCols;
  constexpr int stride = per_col * kCols;
  constexpr int kLoads = (kRows * kCols + stride - 1) / stride;
  __shared__ double scratch[T];
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
    double total = 0.0;
#pragma unroll
    for (int g = 0; g < per_col; ++g) total += scratch[t + g * kCols];
    out[t] = total;
  }
}


// the same sums with a fixed batch of 16 loads per thread where 6 are needed: FENCED = loads from
// clamped indices issued together (ten of them redundant), else each load under its own branch
template <bool FENCED>
__global__ __launch_bounds__(1024) void columnSumBatch16Kernel(const double *rows, double *out) {
  constexpr int T = 1024, per_col = T / kCols, stride = per_col * kCols, kBatch = 16;
  constexpr int total_elems = kRows * kCols;
  __shared__ double scratch[T];
  const int t = threadIdx.x;
  double s = 0.0;
  if (t < stride) {
    double v[kBatch];
    if constexpr (FENCED) {
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int idx = t + k * stride;
        v[k] = rows[idx < total_elems ? idx : t];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < kBatch; ++k) v[k] = (t + k * stride) < total_elems ? v[k] : 0.0;
    } else {
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int idx = t + k * stride;
        v[k] = idx < total_elems ? rows[idx] : 0.0;
      }
    }
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int k = 0; k < kBatch; k += 4) {
      s0 += v[k];
      s1 += v[k + 1];
      s2 += v[k + 2];
      s3 += v[k + 3];
    }
    s = (s0 + s1) + (s2 + s3);
  }
  scratch[t] = s;
  __syncthreads();
  if (t < kCols) {
    double total = 0.0;
#pragma unroll
    for (int g = 0; g < per_col; ++g) total += scratch[t + g * kCols];
    out[t] = total;
  }
}

// one wave per group of columns, no LDS, no barrier: lane l of wave w adds rows l, l + 64, ... of
// column c and the wave reduces with DPP/shuffles
template <int T>
__global__ __launch_bounds__(T) void columnSumShuffleKernel(const double *rows, double *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int waves = T / 64;
  for (int c = wave; c < kCols; c += waves) {
    double v[kRows / 64];
#pragma unroll
    for (int k = 0; k < kRows / 64; ++k) v[k] = rows[(lane + 64 * k) * kCols + c];
    double s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[c] = s;
  }
}

template <int T>
int run(hipStream_t stream, double *rows, double *out, int reps) {
  for (int i = 0; i < reps; ++i) {
    writerKernel<<<kRows, 64, 0, stream>>>(rows, double(i));
    columnSumKernel<T><<<1, T, 0, stream>>>(rows, out);
    writerKernel<<<kRows, 64, 0, stream>>>(rows, double(i));
    columnSumShuffleKernel<T><<<1, T, 0, stream>>>(rows, out);
    writerKernel<<<kRows, 64, 0, stream>>>(rows, double(i));
    emptyKernel<T><<<1, T, 0, stream>>>(out);
  }
  return hipStreamSynchronize(stream) == hipSuccess ? 0 : 1;
}

int main() {
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  double *rows, *out;
  CHECK(hipMalloc(&rows, sizeof(double) * kRows * kCols));
  CHECK(hipMalloc(&out, sizeof(double) * 64));
  const int reps = 2000;
  for (int i = 0; i < reps; ++i) {
    writerKernel<<<kRows, 64, 0, stream>>>(rows, double(i));
    columnSumBatch16Kernel<true><<<1, 1024, 0, stream>>>(rows, out);
    writerKernel<<<kRows, 64, 0, stream>>>(rows, double(i));
    columnSumBatch16Kernel<false><<<1, 1024, 0, stream>>>(rows, out);
  }
  if (run<64>(stream, rows, out, reps) || run<256>(stream, rows, out, reps) ||
      run<512>(stream, rows, out, reps) || run<1024>(stream, rows, out, reps)) {
    std::printf("a launch failed\n");
    return 1;
  }
  double h[kCols];
  CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  // rows hold seed + row + col with seed = reps - 1: column c sums to 256 (seed + c) + 255 * 128
  const double expect0 = 256.0 * (reps - 1) + 255.0 * 128.0;
  std::printf("column 0: %.1f (expected %.1f), column 22: %.1f (expected %.1f)\n", h[0], expect0, h[22],
              expect0 + 256.0 * 22);
  return 0;
}
