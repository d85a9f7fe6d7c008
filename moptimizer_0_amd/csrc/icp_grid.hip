// Uniform-grid construction for the correspondence search, on the GPU.
//
// The grid that icpMatchKernel walks (sweep_kernels.hip) is built once per cloud pair:
//   bounding box of the targets -> cell ids -> stable sort of (cell id, index) -> cell offsets ->
//   targets gathered cell by cell (32-byte padded), sources gathered in cell order.
// Everything O(points) runs on the device; the host only picks the grid resolution from the six
// bounding-box numbers.  The sort is rocPRIM's radix sort (through hipCUB): it is stable, so
// inside a cell the points keep their original order, which is the tie-break rule the search
// documents.  (The reference has no correspondence search to follow - model.h:24-26 leaves it
// to the user's update().)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <vector>

#include "sweep.hpp"

namespace mopt {
namespace {

constexpr int kBoxBlocks = 512;

template <typename S>
__global__ __launch_bounds__(kBlockThreads) void boundingBoxKernel(const S *xyz, long long m,
                                                                   double *block_lohi) {
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x; i < m;
       i += (long long)gridDim.x * kBlockThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = double(xyz[3 * i + a]);
      lo[a] = v < lo[a] ? v : lo[a];
      hi[a] = v > hi[a] ? v : hi[a];
    }
  }
  __shared__ double lds[kBlockThreads / 64][6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) {
      const double l = __shfl_down(lo[a], off, 64), h = __shfl_down(hi[a], off, 64);
      lo[a] = l < lo[a] ? l : lo[a];
      hi[a] = h > hi[a] ? h : hi[a];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lds[wave][a] = lo[a];
      lds[wave][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = lds[0][threadIdx.x];
    for (int w = 1; w < kBlockThreads / 64; ++w) {
      const double o = lds[w][threadIdx.x];
      v = threadIdx.x < 3 ? (o < v ? o : v) : (o > v ? o : v);
    }
    block_lohi[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

struct GridShape {
  double origin[3];
  double cell;
  int dims[3];
};

// cell id of every point (clamped into the grid), and the identity permutation to carry along
template <typename S>
__global__ __launch_bounds__(kBlockThreads) void cellKeyKernel(const S *xyz, long long m,
                                                               const GridShape g,
                                                               unsigned int *keys, int *index) {
  const long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= m) return;
  long long id = 0, stride = 1;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double c = floor((double(xyz[3 * i + a]) - g.origin[a]) / g.cell);
    c = c < 0.0 ? 0.0 : (c >= double(g.dims[a]) ? double(g.dims[a] - 1) : c);
    id += (long long)c * stride;
    stride *= g.dims[a];
  }
  keys[i] = (unsigned int)id;
  index[i] = int(i);
}

// sorted keys -> offsets: cell_start[c] = first position whose key is >= c (c = 0 .. ncells, so the
// last entry is m).  One binary search per cell: uniform work whatever the occupancy pattern.
__global__ __launch_bounds__(kBlockThreads) void cellStartKernel(const unsigned int *sorted_keys,
                                                                 long long m, int ncells,
                                                                 int *cell_start) {
  const long long c = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (c > ncells) return;
  long long lo = 0, hi = m;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)sorted_keys[mid] < c) lo = mid + 1;
    else hi = mid;
  }
  cell_start[c] = int(lo);
}

template <typename S, int OUT_STRIDE>
__global__ __launch_bounds__(kBlockThreads) void gatherPointsKernel(const S *xyz, const int *perm,
                                                                    long long m, S *out) {
  const long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= m) return;
  const long long from = perm[i];
#pragma unroll
  for (int a = 0; a < 3; ++a) out[i * OUT_STRIDE + a] = xyz[3 * from + a];
  if (OUT_STRIDE == 4) out[i * OUT_STRIDE + 3] = S(0);
}

inline int blocksFor(long long m) { return int((m + kBlockThreads - 1) / kBlockThreads); }

}  // namespace

template <typename S>
hipError_t icpBoundingBox(const S *d_xyz, long long m, double lo[3], double hi[3],
                          hipStream_t stream) {
  for (int a = 0; a < 3; ++a) lo[a] = hi[a] = 0.0;
  if (m <= 0) return hipSuccess;
  const int grid = blocksFor(m) < kBoxBlocks ? blocksFor(m) : kBoxBlocks;
  double *d_part = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_part), size_t(grid) * 6 * sizeof(double));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(boundingBoxKernel<S>, dim3(grid), dim3(kBlockThreads), 0, stream, d_xyz, m,
                     d_part);
  std::vector<double> part(size_t(grid) * 6);
  e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost,
                       stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d_part);
  if (e != hipSuccess) return e;
  for (int a = 0; a < 3; ++a) {
    lo[a] = part[a];
    hi[a] = part[3 + a];
  }
  for (int b = 1; b < grid; ++b)
    for (int a = 0; a < 3; ++a) {
      if (part[size_t(b) * 6 + a] < lo[a]) lo[a] = part[size_t(b) * 6 + a];
      if (part[size_t(b) * 6 + 3 + a] > hi[a]) hi[a] = part[size_t(b) * 6 + 3 + a];
    }
  return hipSuccess;
}

template <typename S>
hipError_t icpSortByCell(const S *d_xyz, long long m, const double origin[3], double cell,
                         const int dims[3], int *d_perm, int *d_cell_start, hipStream_t stream) {
  long long ncells = 1;
  GridShape g;
  for (int a = 0; a < 3; ++a) {
    g.origin[a] = origin[a];
    g.dims[a] = dims[a];
    ncells *= dims[a];
  }
  g.cell = cell;
  if (m <= 0) {
    if (d_cell_start)
      return hipMemsetAsync(d_cell_start, 0, size_t(ncells + 1) * sizeof(int), stream);
    return hipSuccess;
  }
  unsigned int *keys = nullptr, *keys_sorted = nullptr;
  int *index = nullptr;
  void *temp = nullptr;
  auto release = [&]() {
    if (keys) (void)hipFree(keys);
    if (keys_sorted) (void)hipFree(keys_sorted);
    if (index) (void)hipFree(index);
    if (temp) (void)hipFree(temp);
  };
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&keys), size_t(m) * sizeof(unsigned int));
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void **>(&keys_sorted), size_t(m) * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&index), size_t(m) * sizeof(int));
  if (e != hipSuccess) {
    release();
    return e;
  }
  hipLaunchKernelGGL(cellKeyKernel<S>, dim3(blocksFor(m)), dim3(kBlockThreads), 0, stream, d_xyz, m,
                     g, keys, index);
  int bits = 1;
  while ((1ll << bits) < ncells) ++bits;
  size_t temp_bytes = 0;
  e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, keys_sorted, index, d_perm,
                                         int(m), 0, bits, stream);
  if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_sorted, index, d_perm,
                                           int(m), 0, bits, stream);
  if (e == hipSuccess && d_cell_start) {
    hipLaunchKernelGGL(cellStartKernel, dim3(blocksFor(ncells + 1)), dim3(kBlockThreads), 0, stream,
                       keys_sorted, m, int(ncells), d_cell_start);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);  // the scratch arrays die here
  release();
  return e;
}

template <typename S>
hipError_t icpGatherPoints(const S *d_xyz, const int *d_perm, long long m, S *d_out, bool padded,
                           hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  if (padded)
    hipLaunchKernelGGL((gatherPointsKernel<S, 4>), dim3(blocksFor(m)), dim3(kBlockThreads), 0,
                       stream, d_xyz, d_perm, m, d_out);
  else
    hipLaunchKernelGGL((gatherPointsKernel<S, 3>), dim3(blocksFor(m)), dim3(kBlockThreads), 0,
                       stream, d_xyz, d_perm, m, d_out);
  return hipGetLastError();
}

#define MOPT_INSTANTIATE_GRID(S)                                                                  \
  template hipError_t icpBoundingBox<S>(const S *, long long, double[3], double[3], hipStream_t); \
  template hipError_t icpSortByCell<S>(const S *, long long, const double[3], double,             \
                                       const int[3], int *, int *, hipStream_t);                  \
  template hipError_t icpGatherPoints<S>(const S *, const int *, long long, S *, bool, hipStream_t);
MOPT_INSTANTIATE_GRID(double)
MOPT_INSTANTIATE_GRID(float)

}  // namespace mopt
