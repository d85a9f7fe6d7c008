// Uniform-grid construction for the correspondence search, on the GPU.
//
// The grid that icpMatchKernel walks (sweep_kernels.hip) is built once per cloud pair:
//   bounding box of the targets -> cell ids -> stable sort of (cell id, index) -> cell offsets ->
//   targets gathered cell by cell (32-byte padded), sources gathered in cell order.
// Everything O(points) runs on the device; the host only picks the grid resolution from the six
// bounding-box numbers.  The sort is a least-significant-digit radix sort written here for wave64
// (8-bit digits, as many passes as the cell count needs: 2-3 in practice): per pass a per-workgroup
// digit histogram, an exclusive scan over the (digit, workgroup) table, and a scatter in which
// every wavefront ranks its 64 keys among equal digits with eight ballots — no atomics on the
// output, so the sort is stable: inside a cell the points keep their original order, which is the
// tie-break rule the search documents.  (The reference has no correspondence search to follow -
// model.h:24-26 leaves it to the user's update().)
#include <hip/hip_runtime.h>

#include <vector>

#include "sweep.hpp"

namespace mopt_detail {  // the size-class cache in front of hipMalloc / hipFree (device_pool.cpp):
hipError_t deviceAlloc(void **out, size_t bytes);  // a grid is built per cloud pair, and hipFree
void deviceRelease(void *p);                       // synchronises the device every time
}  // namespace mopt_detail

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
      const bool finite = fabs(v) <= 1e300;  // (NaN and +-Inf coordinates do not shape the grid)
      lo[a] = finite && v < lo[a] ? v : lo[a];
      hi[a] = finite && v > hi[a] ? v : hi[a];
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

// cell id of every point (clamped into the grid); with parked_key >= 0 a point with a NaN or
// infinite coordinate gets that key instead — one past the last cell, so it sorts behind all others
template <typename S>
__global__ __launch_bounds__(kBlockThreads) void cellKeyKernel(const S *xyz, long long m,
                                                               const GridShape g,
                                                               long long parked_key,
                                                               unsigned int *keys) {
  const long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= m) return;
  long long id = 0, stride = 1;
  bool finite = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double v = double(xyz[3 * i + a]);
    finite = finite && fabs(v) <= 1e300;
    double c = floor((v - g.origin[a]) / g.cell);
    c = c >= 0.0 ? c : 0.0;  // (also a NaN coordinate: such a point is never anyone's nearest)
    c = c >= double(g.dims[a]) ? double(g.dims[a] - 1) : c;
    id += (long long)c * stride;
    stride *= g.dims[a];
  }
  keys[i] = (unsigned int)((parked_key >= 0 && !finite) ? parked_key : id);
}

// how many of the sorted keys are the parked key (they are the last ones)
__global__ void countParkedKernel(const unsigned int *sorted_keys, long long m,
                                  unsigned int parked_key, long long *out) {
  long long lo = 0, hi = m;  // first position whose key is >= parked_key
  while (lo < hi) {
    const long long mid = (lo + hi) / 2;
    if (sorted_keys[mid] < parked_key) lo = mid + 1;
    else hi = mid;
  }
  *out = m - lo;
}

// ---- stable LSD radix sort of (cell id, point index) pairs, 8 bits per pass --------------------
constexpr int kSortRounds = 8;                            // keys per thread
constexpr int kSortTile = kBlockThreads * kSortRounds;    // keys per workgroup
constexpr int kDigits = 256;
static_assert(kBlockThreads == kDigits, "one thread per digit in the tables below");

// hist[digit][workgroup]: how many keys of this workgroup's tile carry the digit
__global__ __launch_bounds__(kBlockThreads) void radixHistogramKernel(
    const unsigned int *__restrict__ keys, long long m, int shift, unsigned int *__restrict__ hist,
    int num_groups) {
  __shared__ unsigned int h[kDigits];
  h[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const long long i = base + r * kBlockThreads + threadIdx.x;
    if (i < m) atomicAdd(&h[(keys[i] >> shift) & (kDigits - 1)], 1u);  // LDS atomic: counts only
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * num_groups + blockIdx.x] = h[threadIdx.x];
}

// In-place exclusive scan of `length` counters: in the digit-major table the result is the first
// output slot of every (digit, workgroup).  Three coalesced steps — every workgroup scans its own
// 1024 counters (16-byte loads, four per thread) and leaves their total; one workgroup scans the
// totals; every counter then takes its workgroup's offset.  (A single workgroup walking the whole
// table, thread by thread in strided chunks, took 0.33 ms per pass at 1 M points and 0.8 ms at 4 M.)
constexpr int kScanPerThread = 4;
constexpr int kScanTile = kBlockThreads * kScanPerThread;

__device__ __forceinline__ unsigned int blockExclusiveScan(unsigned int value, unsigned int *total) {
  __shared__ unsigned int part[kBlockThreads];
  part[threadIdx.x] = value;
  __syncthreads();
  for (int off = 1; off < kBlockThreads; off <<= 1) {  // Hillis-Steele, inclusive
    const unsigned int add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  *total = part[kBlockThreads - 1];
  return part[threadIdx.x] - value;
}

__global__ __launch_bounds__(kBlockThreads) void scanTilesKernel(unsigned int *data, long long length,
                                                                 unsigned int *tile_totals) {
  const long long base = (long long)blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
  unsigned int v[kScanPerThread];
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k) v[k] = base + k < length ? data[base + k] : 0u;
  const unsigned int mine = (v[0] + v[1]) + (v[2] + v[3]);
  unsigned int total;
  unsigned int run = blockExclusiveScan(mine, &total);
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k) {
    if (base + k < length) data[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) tile_totals[blockIdx.x] = total;
}

// exclusive scan of the tile totals by one workgroup (chunked: any number of tiles)
__global__ __launch_bounds__(kBlockThreads) void scanTotalsKernel(unsigned int *totals, int count) {
  const int chunk = (count + kBlockThreads - 1) / kBlockThreads;
  const int lo = threadIdx.x * chunk;
  const int hi = lo + chunk < count ? lo + chunk : count;
  unsigned int sum = 0;
  for (int i = lo; i < hi; ++i) sum += totals[i];
  unsigned int all;
  unsigned int run = blockExclusiveScan(sum, &all);
  for (int i = lo; i < hi; ++i) {
    const unsigned int v = totals[i];
    totals[i] = run;
    run += v;
  }
}

__global__ __launch_bounds__(kBlockThreads) void addTileOffsetsKernel(
    unsigned int *data, long long length, const unsigned int *tile_offsets) {
  const unsigned int offset = tile_offsets[blockIdx.x];
  const long long base = (long long)blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
#pragma unroll
  for (int k = 0; k < kScanPerThread; ++k)
    if (base + k < length) data[base + k] += offset;
}

// values_in == nullptr: the value of key i is i (first pass).
__global__ __launch_bounds__(kBlockThreads) void radixScatterKernel(
    const unsigned int *__restrict__ keys_in, const int *__restrict__ values_in, long long m,
    int shift, const unsigned int *__restrict__ offsets, int num_groups,
    unsigned int *__restrict__ keys_out, int *__restrict__ values_out) {
  __shared__ unsigned int next_slot[kDigits];                      // per digit, for this workgroup
  __shared__ unsigned int wave_count[kBlockThreads / 64][kDigits];  // keys per (wave, digit), per round
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  next_slot[threadIdx.x] = offsets[(size_t)threadIdx.x * num_groups + blockIdx.x];
  const long long base = (long long)blockIdx.x * kSortTile;
  for (int r = 0; r < kSortRounds; ++r) {
#pragma unroll
    for (int w = 0; w < kBlockThreads / 64; ++w) wave_count[w][threadIdx.x] = 0;
    __syncthreads();
    const long long i = base + r * kBlockThreads + threadIdx.x;
    const bool valid = i < m;
    const unsigned int key = valid ? keys_in[i] : 0u;
    const unsigned int digit = (key >> shift) & (kDigits - 1);
    // lanes of this wavefront holding the same digit: eight ballots, one per bit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long with_bit = __ballot(bit);
      peers &= bit ? with_bit : ~with_bit;
    }
    const unsigned int rank = (unsigned int)__popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wave_count[wave][digit] = (unsigned int)__popcll(peers);
    __syncthreads();
    if (valid) {
      unsigned int before = 0;
      for (int w = 0; w < wave; ++w) before += wave_count[w][digit];
      const unsigned int pos = next_slot[digit] + before + rank;
      keys_out[pos] = key;
      values_out[pos] = values_in ? values_in[i] : int(i);
    }
    __syncthreads();
    unsigned int used = 0;
#pragma unroll
    for (int w = 0; w < kBlockThreads / 64; ++w) used += wave_count[w][threadIdx.x];
    next_slot[threadIdx.x] += used;  // column threadIdx.x is touched by this thread only from here on
  }
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
  hipError_t e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&d_part), size_t(grid) * 6 * sizeof(double));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(boundingBoxKernel<S>, dim3(grid), dim3(kBlockThreads), 0, stream, d_xyz, m,
                     d_part);
  std::vector<double> part(size_t(grid) * 6);
  e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost,
                       stream);
  const hipError_t synced = hipStreamSynchronize(stream);  // (also before the block goes back)
  if (e == hipSuccess) e = synced;
  mopt_detail::deviceRelease(d_part);
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
  for (int a = 0; a < 3; ++a)
    if (!(lo[a] <= hi[a])) lo[a] = hi[a] = 0.0;  // no finite coordinate on this axis
  return hipSuccess;
}

template <typename S>
hipError_t icpSortByCell(const S *d_xyz, long long m, const double origin[3], double cell,
                         const int dims[3], int *d_perm, int *d_cell_start, hipStream_t stream,
                         long long *num_parked) {
  if (num_parked) *num_parked = 0;
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
  // scratch: two key buffers, one value buffer (the other one is d_perm) and the histogram table
  const int num_groups = int((m + kSortTile - 1) / kSortTile);
  int bits = 1;
  while ((1ll << bits) < ncells + (num_parked ? 1 : 0)) ++bits;
  const int passes = (bits + 7) / 8;
  const long long table = (long long)kDigits * num_groups;  // (digit, workgroup) counters
  const int scan_tiles = int((table + kScanTile - 1) / kScanTile);
  unsigned int *keys = nullptr, *keys_alt = nullptr, *hist = nullptr, *tile_totals = nullptr;
  int *values_alt = nullptr;
  auto release = [&]() {
    (void)hipStreamSynchronize(stream);  // nothing queued may still use a block that goes back
    mopt_detail::deviceRelease(keys);
    mopt_detail::deviceRelease(keys_alt);
    mopt_detail::deviceRelease(values_alt);
    mopt_detail::deviceRelease(hist);
    mopt_detail::deviceRelease(tile_totals);
  };
  hipError_t e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&keys), size_t(m) * sizeof(unsigned int));
  if (e == hipSuccess)
    e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&keys_alt), size_t(m) * sizeof(unsigned int));
  if (e == hipSuccess)
    e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&values_alt), size_t(m) * sizeof(int));
  if (e == hipSuccess)
    e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&hist), size_t(table) * sizeof(unsigned int));
  if (e == hipSuccess)
    e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&tile_totals), size_t(scan_tiles) * sizeof(unsigned int));
  if (e != hipSuccess) {
    release();
    return e;
  }
  hipLaunchKernelGGL(cellKeyKernel<S>, dim3(blocksFor(m)), dim3(kBlockThreads), 0, stream, d_xyz, m,
                     g, num_parked ? ncells : -1ll, keys);
  const unsigned int *keys_in = keys;
  unsigned int *keys_out = keys_alt;
  const int *values_in = nullptr;  // pass 1: value i = index i
  for (int p = 0; p < passes; ++p) {
    // ping-pong so that the last pass writes the permutation into d_perm
    int *values_out = ((passes - 1 - p) % 2 == 0) ? d_perm : values_alt;
    hipLaunchKernelGGL(radixHistogramKernel, dim3(num_groups), dim3(kBlockThreads), 0, stream,
                       keys_in, m, 8 * p, hist, num_groups);
    hipLaunchKernelGGL(scanTilesKernel, dim3(scan_tiles), dim3(kBlockThreads), 0, stream, hist,
                       table, tile_totals);
    hipLaunchKernelGGL(scanTotalsKernel, dim3(1), dim3(kBlockThreads), 0, stream, tile_totals,
                       scan_tiles);
    hipLaunchKernelGGL(addTileOffsetsKernel, dim3(scan_tiles), dim3(kBlockThreads), 0, stream, hist,
                       table, tile_totals);
    hipLaunchKernelGGL(radixScatterKernel, dim3(num_groups), dim3(kBlockThreads), 0, stream, keys_in,
                       values_in, m, 8 * p, hist, num_groups, keys_out, values_out);
    values_in = values_out;
    unsigned int *recycled = const_cast<unsigned int *>(keys_in);
    keys_in = keys_out;
    keys_out = recycled;
  }
  const unsigned int *keys_sorted = keys_in;
  e = hipGetLastError();
  if (e == hipSuccess && d_cell_start) {
    hipLaunchKernelGGL(cellStartKernel, dim3(blocksFor(ncells + 1)), dim3(kBlockThreads), 0, stream,
                       keys_sorted, m, int(ncells), d_cell_start);
    e = hipGetLastError();
  }
  long long *d_parked = nullptr;
  if (e == hipSuccess && num_parked) {
    e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&d_parked), sizeof(long long));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(countParkedKernel, dim3(1), dim3(1), 0, stream, keys_sorted, m,
                         (unsigned int)ncells, d_parked);
      e = hipMemcpyAsync(num_parked, d_parked, sizeof(long long), hipMemcpyDeviceToHost, stream);
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);  // the scratch arrays die here
  mopt_detail::deviceRelease(d_parked);
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

// how many cells hold at least one target: one count per workgroup, added up on the host (this is
// part of building the grid, which ends in a stream synchronisation anyway)
__global__ __launch_bounds__(kBlockThreads) void occupiedCellsKernel(const int *cell_start,
                                                                     long long ncells,
                                                                     unsigned int *per_block) {
  const long long c = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  const bool occupied = c < ncells && cell_start[c + 1] > cell_start[c];
  const int in_block = __syncthreads_count(occupied ? 1 : 0);
  if (threadIdx.x == 0) per_block[blockIdx.x] = (unsigned int)in_block;
}

hipError_t icpCountOccupiedCells(const int *d_cell_start, long long ncells, long long *occupied,
                                 hipStream_t stream) {
  *occupied = 0;
  if (ncells <= 0) return hipSuccess;
  const unsigned blocks = unsigned((ncells + kBlockThreads - 1) / kBlockThreads);
  unsigned int *d_counts = nullptr;
  hipError_t e = mopt_detail::deviceAlloc(reinterpret_cast<void **>(&d_counts), size_t(blocks) * sizeof(unsigned int));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(occupiedCellsKernel, dim3(blocks), dim3(kBlockThreads), 0, stream, d_cell_start,
                     ncells, d_counts);
  std::vector<unsigned int> counts(blocks);
  e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(counts.data(), d_counts, size_t(blocks) * sizeof(unsigned int),
                       hipMemcpyDeviceToHost, stream);
  const hipError_t synced = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = synced;
  mopt_detail::deviceRelease(d_counts);
  if (e != hipSuccess) return e;
  for (unsigned int v : counts) *occupied += v;
  return hipSuccess;
}

#define MOPT_INSTANTIATE_GRID(S)                                                                  \
  template hipError_t icpBoundingBox<S>(const S *, long long, double[3], double[3], hipStream_t); \
  template hipError_t icpSortByCell<S>(const S *, long long, const double[3], double,             \
                                       const int[3], int *, int *, hipStream_t, long long *);     \
  template hipError_t icpGatherPoints<S>(const S *, const int *, long long, S *, bool, hipStream_t);
MOPT_INSTANTIATE_GRID(double)
MOPT_INSTANTIATE_GRID(float)

}  // namespace mopt
