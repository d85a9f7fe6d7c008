// Device-side building blocks shared by the sweep kernels (sweep_kernels.hip, fd_kernels.hip):
// 16-byte lane loads, the ping-pong tile walk and the LDS-transpose workgroup reduction.  Everything
// is in an anonymous namespace of the including translation unit.
#pragma once

#include "sweep.hpp"

#include <hip/hip_ext.h>

#include <type_traits>

namespace mopt {
namespace {

// In-kernel time stamps for the diagnostic builds under scripts/probes/ (they define
// MOPT_STAMP_WORDS and include a kernel source whole): wave 0 of a workgroup stores the 100 MHz
// wall clock at point k into a buffer of its own that nothing else reads.  The library's build has
// no stamp in it.
#ifdef MOPT_STAMP_WORDS
__device__ unsigned long long *g_stamp_buffer;
#define MOPT_STAMP(k)                                                                              \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (threadIdx.x == 0)                                                                          \
      g_stamp_buffer[size_t(blockIdx.x) * MOPT_STAMP_WORDS + (k)] = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define MOPT_STAMP(k) ((void)0)
#endif

template <typename S>
struct alignas(16) Pack {
  S v[16 / sizeof(S)];
};

template <typename S>
__device__ __forceinline__ Pack<S> loadPack(const S *p) {
  return *reinterpret_cast<const Pack<S> *>(p);
}

// Streaming (non-temporal) 16-byte load: the line is not kept in L2 / Infinity Cache.  Used when
// the data set is larger than the 256 MiB Infinity Cache, where a sweep can never re-use a line
// of the previous sweep anyway and allocating them only costs fill bandwidth.
template <typename S>
__device__ __forceinline__ Pack<S> loadPackStreaming(const S *p) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
  Pack<S> out;
  __builtin_memcpy(&out, &raw, sizeof out);
  return out;
}

template <typename S>
__device__ __forceinline__ S lossWeight(int kind, S param, S s) {
  if (kind == kLossGemanMcClure) {
    const S d = s + param;
    return (param * param) / (d * d);
  }
  return S(1);
}

// r = (R p + t) - q with the association of a 4x4 * [p;1] product followed by the subtraction
// (tst/point2point.cpp:42-45).
template <typename S>
__device__ __forceinline__ void p2pResidual(const S (&T)[12], const S (&p)[3], const S (&q)[3],
                                            S (&r)[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const S warped = ((T[a * 4 + 0] * p[0] + T[a * 4 + 1] * p[1]) + T[a * 4 + 2] * p[2]) +
                     T[a * 4 + 3];
    r[a] = warped - q[a];
  }
}

// A slot of the tile layout holds a correspondence when its index is inside the data set and
// its target is not the NaN marker — the device form of the model's `f` returning false for an
// index (model.h:32, linearization.h:102,144): padding, and source points the correspondence
// search (icpMatchKernel) left unmatched.
template <typename S>
__device__ __forceinline__ bool isCorrespondence(long long index, long long count, S target_x) {
  return index < count && target_x == target_x;
}

// Walks this workgroup's tiles (blockIdx.x, + gridDim.x, ...) with two register sets used in
// ping-pong: the six 16-byte loads of the NEXT tile are issued into the idle set before the
// arithmetic of the current one starts, and nothing is ever copied between the sets.
//
// Two details matter to the generated waits (checked in the ISA):
//  * a copy `cur = nxt` at the loop end forces `s_waitcnt vmcnt(0)` there, leaving one tile in
//    flight per wave;
//  * a *conditional* prefetch (`if (next < n) load`) makes the wait at the join conservative
//    (vmcnt(0) again), because on the not-taken path the needed loads are the youngest.  So the
//    prefetch is unconditional and, past the end, re-reads this workgroup's last tile (an L2 hit).
// body(packs, first): packs[plane].v[e] is coordinate `plane` of correspondence first + e.
// block / num_blocks: this workgroup's place among those sweeping this cost (a launch may carry the
// workgroups of several costs: the *ResidentSetKernel forms); by default the whole grid.
template <typename S, bool STREAMING, typename Body>
__device__ __forceinline__ void sweepTiles(const S *tiles, int num_tiles, Body &&body, int block,
                                           int num_blocks) {
  constexpr int V = TileShape<S>::kVec;
  constexpr int TP = TileShape<S>::kPoints;
  const int stride = num_blocks;
  const int first_tile = block;
  if (first_tile >= num_tiles) return;
  const int mine = (num_tiles - first_tile + stride - 1) / stride;  // tiles of this workgroup
  const S *lane_base = tiles + threadIdx.x * V;
  auto tileOf = [&](int i) { return first_tile + (i < mine ? i : mine - 1) * stride; };
  auto load = [&](Pack<S>(&dst)[6], int tile) {
    const S *base = lane_base + size_t(tile) * TileShape<S>::kP2PScalars;
#pragma unroll
    for (int pl = 0; pl < 6; ++pl)
      dst[pl] = STREAMING ? loadPackStreaming<S>(base + pl * TP) : loadPack<S>(base + pl * TP);
  };
  Pack<S> a[6], b[6];
  load(a, tileOf(0));
  for (int i = 0; i < mine; i += 2) {
    load(b, tileOf(i + 1));
    body(a, (long long)tileOf(i) * TP + threadIdx.x * V);
    if (i + 1 >= mine) break;
    load(a, tileOf(i + 2));
    body(b, (long long)tileOf(i + 1) * TP + threadIdx.x * V);
  }
}
template <typename S, bool STREAMING, typename Body>
__device__ __forceinline__ void sweepTiles(const S *tiles, int num_tiles, Body &&body) {
  sweepTiles<S, STREAMING>(tiles, num_tiles, static_cast<Body &&>(body), int(blockIdx.x),
                           int(gridDim.x));
}

// The moments of the V correspondences one lane holds of a tile, added to `acc`.
template <typename S>
__device__ __forceinline__ void momentsOfPack(double (&acc)[kAccMoments], const Pack<S> (&cur)[6],
                                              long long first, const P2PSweepArgs<S> &A) {
  constexpr int V = TileShape<S>::kVec;
  // fp64: the moment updates contract into one v_fma_f64 each, straight into the accumulators.
  // fp32: the four correspondences a lane holds per tile are first summed in fp32 (fp32 FMAs),
  // then promoted once — 23 conversions + fp64 adds per tile instead of 92, which otherwise makes
  // the fp32 sweep VALU-bound.
  using Local = typename std::conditional<sizeof(S) == 8, double, S>::type;
  Local loc[kAccMoments];
  if constexpr (sizeof(S) == 4) {
#pragma unroll
    for (int k = 0; k < kAccMoments; ++k) loc[k] = Local(0);
  }
  auto add = [&](int k, S v) {
    if constexpr (sizeof(S) == 8)
      acc[k] += double(v);
    else
      loc[k] += v;
  };
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const S p[3] = {cur[0].v[e], cur[1].v[e], cur[2].v[e]};
    const bool valid = isCorrespondence(first + e, A.count, cur[3].v[e]);
    const S q[3] = {valid ? cur[3].v[e] : S(0), valid ? cur[4].v[e] : S(0),
                    valid ? cur[5].v[e] : S(0)};
    S r[3];
    p2pResidual<S>(A.T[0], p, q, r);
    S rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    S w = lossWeight<S>(A.loss_kind, A.loss_param, rr);
    w = valid ? w : S(0);
    rr = valid ? rr : S(0);
    const S wp[3] = {w * p[0], w * p[1], w * p[2]};
    const S wr[3] = {w * r[0], w * r[1], w * r[2]};
    add(0, w);
    add(1, wp[0]);
    add(2, wp[1]);
    add(3, wp[2]);
    add(4, wp[0] * p[0]);
    add(5, wp[0] * p[1]);
    add(6, wp[0] * p[2]);
    add(7, wp[1] * p[1]);
    add(8, wp[1] * p[2]);
    add(9, wp[2] * p[2]);
    add(10, wr[0]);
    add(11, wr[1]);
    add(12, wr[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) add(13 + 3 * k + c, p[k] * wr[c]);
    add(22, rr);
  }
  if constexpr (sizeof(S) == 4) {
#pragma unroll
    for (int k = 0; k < kAccMoments; ++k) acc[k] += double(loc[k]);
  }
}

__device__ __forceinline__ double waveSum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Per-thread accumulators -> one row of `NACC` doubles per workgroup, through an LDS transpose.
//
// Cross-lane shuffles (ds_bpermute) go through the CU's single LDS crossbar: 6 steps x 2 dwords
// x NACC values x 4 waves of them cost microseconds per workgroup (measured: the 23-value
// epilogue took as long as ~4 tiles of streaming).  Instead every lane stores its values once
// (conflict-free ds_write_b64, row of 64 lanes per value, rows padded to 72 doubles so that four
// consecutive value-rows tile the 64 banks), then thread (k, part) adds the 4 waves x 8 lanes of
// value k whose lane index is = part (mod 8) and the 8 parts are combined with three xor
// shuffles.  Order of additions is fixed, so the row is reproducible bit for bit.  Rows of up to 32
// values (one per 8 threads of a 256-thread workgroup) go through in one pass — the 28 of a
// symmetric linearization in 64.5 KB of LDS, one pair of barriers instead of two —, longer ones in
// chunks of 23 (53 KB per workgroup).
constexpr int kReduceChunk = 23;
constexpr int kReduceOnePass = 32;
constexpr int kReduceRow = 72;

template <int NACC, int THREADS = kBlockThreads>
__device__ __forceinline__ void blockReduceStore(double (&acc)[NACC], double *out_row) {
  constexpr int kWaves = THREADS / 64;
  constexpr int kChunk = NACC <= kReduceOnePass ? NACC : kReduceChunk;
  constexpr int kPasses = (NACC + kChunk - 1) / kChunk;
  __shared__ double lds[kWaves][kChunk][kReduceRow];
  // two workgroups of the VALU-heavy sweeps share a CU's 160 KB (gfx950): a row length that takes
  // the one-pass form past half of it would silently drop them to one — or not link at all on a
  // part with 64 KB
  static_assert(sizeof(double) * kWaves * kChunk * kReduceRow <= 80 * 1024,
                "blockReduceStore: more than half of a gfx950 CU's LDS for one workgroup");
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int k_read = threadIdx.x >> 3;  // value handled in the read phase (0..31)
  const int part = threadIdx.x & 7;
#pragma unroll
  for (int pass = 0; pass < kPasses; ++pass) {
    if (pass > 0) __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kChunk; ++kk) {
      const int k = pass * kChunk + kk;
      if (k < NACC) lds[wave][kk][lane] = acc[k];
    }
    __syncthreads();
    const int k_out = pass * kChunk + k_read;
    if (k_read < kChunk && k_out < NACC) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w)
#pragma unroll
        for (int j = 0; j < 8; ++j) v += lds[w][k_read][j * 8 + part];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      if (part == 0) out_row[k_out] = v;
    }
  }
}


// Which cost of a ResidentSweepSet this workgroup sweeps: workgroups [first_block[k],
// first_block[k + 1]) take cost k.
__device__ __forceinline__ int costOfBlock(const ResidentSweepSet &set) {
  int k = 0;
  while (k + 1 < set.num_costs && int(blockIdx.x) >= set.first_block[k + 1]) ++k;
  return k;
}

// moments / cost / forward-difference sweeps: (tiles, num_tiles, args) signature (the leading scalar
// arguments are preloaded into SGPRs at wave launch); optionally a timestamped dispatch
template <typename Kernel, typename S>
hipError_t launchTiled(Kernel kernel, int grid, const LaunchSite &site, const P2PSweepArgs<S> &args) {
  if (site.aql.queue && site.aql_used && !site.time_start &&
      mopt_detail::aqlLaunch(site.aql, kernel, uint32_t(grid), uint32_t(kBlockThreads), args.tiles,
                             args.num_tiles, args)) {
    *site.aql_used = true;
    return hipSuccess;
  }
  if (site.time_start && site.time_stop)
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kBlockThreads), 0, site.stream, site.time_start,
                          site.time_stop, 0, args.tiles, args.num_tiles, args);
  else
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlockThreads), 0, site.stream, args.tiles,
                       args.num_tiles, args);
  return hipGetLastError();
}

}  // namespace
}  // namespace mopt
