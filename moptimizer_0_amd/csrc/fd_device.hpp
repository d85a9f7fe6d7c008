// The forward-difference linearization of the point-to-point model evaluated as the reference does
// (linearization.h:65-124 over tst/point2point.cpp:32-51): the arithmetic of one correspondence — seven
// residuals, eighteen quotients, w J^T S J and w J^T S r entry by entry — and the accumulation over the
// packs a lane is handed.  Shared by fd_kernels.hip (the streaming sweeps; its header comment says why the
// arithmetic is spelled the way it is) and sweep_kernels.hip (the one-launch solve of small problems,
// which holds its correspondences in registers).  Anonymous namespace of the including translation unit.
#pragma once

#include "sweep_device.hpp"

namespace mopt {
namespace {

// Two fp32 points side by side: every value of the per-point arithmetic below can be a pair, and the
// compiler turns the element-wise operations into v_pk_{mul,add,fma}_f32 — one instruction for two
// points.  (Scalar fp32 instructions issue at the rate of fp64 ones on this machine; the fp32 rate
// is the packed rate.)  The constants stay scalars: a packed instruction broadcasts an SGPR operand.
typedef float PointPair __attribute__((ext_vector_type(2)));
struct PairValid {
  bool a, b;
};
__device__ __forceinline__ bool validOf(long long index, long long count, float tx) {
  return isCorrespondence(index, count, tx);
}
__device__ __forceinline__ bool validOf(long long index, long long count, double tx) {
  return isCorrespondence(index, count, tx);
}
__device__ __forceinline__ PairValid validOf(long long index, long long count, PointPair tx) {
  return PairValid{isCorrespondence(index, count, tx.x), isCorrespondence(index + 1, count, tx.y)};
}
template <typename X>
__device__ __forceinline__ X keepValid(bool valid, X x) {
  return valid ? x : X(0);
}
__device__ __forceinline__ PointPair keepValid(PairValid valid, PointPair x) {
  return PointPair{valid.a ? x.x : 0.0f, valid.b ? x.y : 0.0f};
}
// c * a + b with a scalar constant c
__device__ __forceinline__ double fmaConst(double c, double a, double b) {
  return __builtin_fma(c, a, b);
}
__device__ __forceinline__ float fmaConst(float c, float a, float b) {
  return __builtin_fmaf(c, a, b);
}
__device__ __forceinline__ PointPair fmaConst(float c, PointPair a, PointPair b) {
  return __builtin_elementwise_fma(PointPair{c, c}, a, b);
}

// acc += a * b: one v_fma_f64 in fp64; in fp32 the product is rounded to fp32 first, as the
// reference's float instantiation does, and the running sum is fp64 — or, where the sums are formed
// in fp32 first (a `float` or PointPair accumulator, below), one fp32 FMA
template <typename Acc, typename X>
constexpr bool kPairInto = __is_same(Acc, PointPair) && __is_same(X, PointPair);

template <typename Acc, typename X>
__device__ __forceinline__ void accFma(Acc &acc, X a, X b) {
  if constexpr (kPairInto<Acc, X>)
    acc = __builtin_elementwise_fma(a, b, acc);
  else if constexpr (sizeof(Acc) == 4)
    acc = __builtin_fmaf(a, b, acc);
  else if constexpr (sizeof(X) == 8)
    acc = __builtin_fma(a, b, acc);
  else
    acc += double(a * b);
}
template <typename Acc, typename X>
__device__ __forceinline__ void accDot3(Acc &acc, const X (&a)[3], const X (&b)[3]) {
  if constexpr (kPairInto<Acc, X>) {
    acc = __builtin_elementwise_fma(
        a[2], b[2], __builtin_elementwise_fma(a[1], b[1], __builtin_elementwise_fma(a[0], b[0], acc)));
  } else if constexpr (sizeof(Acc) == 4) {
    acc = __builtin_fmaf(a[2], b[2], __builtin_fmaf(a[1], b[1], __builtin_fmaf(a[0], b[0], acc)));
  } else if constexpr (sizeof(X) == 8) {
    acc = __builtin_fma(a[2], b[2], __builtin_fma(a[1], b[1], __builtin_fma(a[0], b[0], acc)));
  } else {
    acc += double((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]);
  }
}

// Where the 27 rotation entries at x + h_j e_j (j = 3..5) live during the sweep.  Everything else a
// point needs — [R | t] at x, t_j + h_j, 1 / h_j, the covariance: 30-39 scalars — is read from the
// kernel arguments and stays in scalar registers (a VALU instruction takes one scalar operand), and
// x + h e_j for a rotation parameter leaves t untouched, so the perturbed transforms need no
// translation column.  What is left does not fit the scalar file next to that (54 more SGPRs):
//   kFdRotationLds        re-read from LDS per point (same-address reads, 14 ds_read_b128; every
//                         one of them still returns 1 KiB to the wave, and with all seven transforms
//                         in LDS — 30 reads per point — that return path, not the VALU, set the pace
//                         at 88-97 us for 10 M points)
//   kFdRotationRegisters  54 VGPRs per lane, loaded once
//   kFdRotationMixed      two of the three perturbed rotations in registers (36 VGPRs), the third
//                         re-read from LDS
//   kFdRotationMixedPlus  + six of the third one's nine entries: 256 VGPRs, what fits next to the 43
//                         accumulators of the general form at two waves per SIMD
enum FdRotationHome : int {
  kFdRotationLds = 0,
  kFdRotationRegisters = 1,
  kFdRotationMixed = 2,
  kFdRotationMixedPlus = 3
};

// for_each_pack(body): calls body(packs, first) for every 16-byte pack of correspondences this lane is to
// evaluate — the ping-pong walk over this workgroup's tiles (p2pForwardDiffBody below), or the tiles a
// lane holds in registers (the one-launch solve).  out_row: where the workgroup's NACC sums go.
template <typename S, int COV, int HOME, typename ForEachPack>
__device__ __forceinline__ void p2pForwardDiffRow(const P2PSweepArgs<S> &A,
                                                  const P2PSweepArgs<S> &in_memory,
                                                  ForEachPack &&for_each_pack, double *out_row) {
  constexpr int V = TileShape<S>::kVec;
  constexpr int NACC = (COV == kCovGeneral) ? kAccFull : kAccSym;
  __shared__ S Rlds[3][12];  // [c][a * 3 + k] = R(x + h_{3+c} e_{3+c})(a, k); 9 of 12 used
  if (threadIdx.x < 27) {
    // the one access indexed by lane: from the arguments where they lie in memory (kernel
    // arguments, or the resident forms' block in HBM) — indexing a by-value copy by lane puts the
    // whole 840-byte struct into every lane's scratch
    const int c = threadIdx.x / 9, ak = threadIdx.x % 9;
    Rlds[c][ak] = in_memory.T[4 + c][(ak / 3) * 4 + ak % 3];
  }
  __syncthreads();
  // kFdRotationMixed: the first two perturbed rotations in registers (36 VGPRs), the third re-read
  // from LDS — for the general covariance form, whose 43 accumulators leave no room for all three
  constexpr int kInRegs = HOME == kFdRotationRegisters ? 3 : (HOME >= kFdRotationMixed ? 2 : 0);
  // kFdRotationMixedPlus: + the first kExtra entries of the third rotation
  constexpr int kExtra = HOME == kFdRotationMixedPlus ? 6 : 0;
  S Rreg[kInRegs ? kInRegs : 1][9];
  S Rextra[kExtra ? kExtra : 1];
#pragma unroll
  for (int c = 0; c < kInRegs; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) Rreg[c][k] = Rlds[c][k];
#pragma unroll
  for (int k = 0; k < kExtra; ++k) Rextra[k] = Rlds[2][k];

  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  const long long count = A.count;
  // H(i, j) in the row layout of the finalize kernel: upper triangle column-wise, or full column-major
  auto at = [](int i, int j) { return COV == kCovGeneral ? j * kNumParams + i : j * (j + 1) / 2 + i; };
  constexpr int kB = (COV == kCovGeneral) ? 36 : 21;  // first entry of b

  // `robust`: the loss kind is taken out of the point (a branch inside would split the basic block
  // and let the compiler sink one point's accumulation below the next point's Jacobian)
  // X: the value of one point (S) or of two fp32 points side by side (PointPair)
  auto point = [&](auto robust, const auto (&in)[6], long long index, auto &dst) {
    using X = std::remove_cv_t<std::remove_reference_t<decltype(in[0])>>;
    const X p[3] = {in[0], in[1], in[2]};
    const auto valid = validOf(index, count, in[3]);
    const X q[3] = {keepValid(valid, in[3]), keepValid(valid, in[4]), keepValid(valid, in[5])};
    X s[3], r[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      s[a] = fmaConst(A.T[0][a * 4 + 2], p[2],
                      fmaConst(A.T[0][a * 4 + 1], p[1], A.T[0][a * 4 + 0] * p[0]));
      r[a] = (s[a] + A.T[0][a * 4 + 3]) - q[a];
    }
    // translation columns: only entry (j, j) moves (t_j + h_j is entry (j, 3) of the transform at
    // x + h_j e_j)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      d[j] = (((s[j] + A.T[1 + j][j * 4 + 3]) - q[j]) - r[j]) * A.inv_h[j];
    // rotation columns: a transformed point each
    X Acol[3][3];  // Acol[c][a] = J[a][3 + c]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      S Rc[9];
      if (c < kInRegs) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rc[k] = Rreg[c < kInRegs ? c : 0][k];
      } else {
        asm volatile("" ::: "memory");  // from LDS here, not from registers kept alive
#pragma unroll
        for (int k = 0; k < 9; ++k) Rc[k] = (k < kExtra) ? Rextra[k < kExtra ? k : 0] : Rlds[c][k];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const X sp = fmaConst(Rc[a * 3 + 2], p[2], fmaConst(Rc[a * 3 + 1], p[1], Rc[a * 3 + 0] * p[0]));
        const X rp = (sp + A.T[0][a * 4 + 3]) - q[a];
        Acol[c][a] = (rp - r[a]) * A.inv_h[3 + c];
      }
    }
    X rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    X w = X(1);
    if constexpr (decltype(robust)::value) w = lossWeight<X>(kLossGemanMcClure, X(A.loss_param), rr);
    w = keepValid(valid, w);
    rr = keepValid(valid, rr);
    const X wd[3] = {w * d[0], w * d[1], w * d[2]};
    if constexpr (COV == kCovIdentity) {
      // w J^T J, w J^T r with J = [diag(d) | A]: entries (0,1), (0,2), (1,2) are sums of exact zeros
      X wA[3][3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int a = 0; a < 3; ++a) wA[c][a] = w * Acol[c][a];
#pragma unroll
      for (int i = 0; i < 3; ++i) accFma(dst[at(i, i)], wd[i], d[i]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int i = 0; i < 3; ++i) accFma(dst[at(i, 3 + c)], wd[i], Acol[c][i]);
#pragma unroll
        for (int c2 = 0; c2 <= c; ++c2) accDot3(dst[at(3 + c2, 3 + c)], wA[c2], Acol[c]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) accFma(dst[kB + i], wd[i], r[i]);
#pragma unroll
      for (int c = 0; c < 3; ++c) accDot3(dst[kB + 3 + c], wA[c], r);
    } else {
      // w S A (column c: wSA[c][a] = w sum_b S(a, b) A(b, c)), S (w d) and S (w r)
      X wSA[3][3], Swd[3][3], wSr[3];
      const X wr[3] = {w * r[0], w * r[1], w * r[2]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          wSA[c][a] =
              w * fmaConst(A.cov[a * 3 + 2], Acol[c][2],
                           fmaConst(A.cov[a * 3 + 1], Acol[c][1], A.cov[a * 3 + 0] * Acol[c][0]));
#pragma unroll
        for (int j = 0; j < 3; ++j) Swd[j][a] = A.cov[a * 3 + j] * wd[j];  // S(a, j) w d_j
        wSr[a] = fmaConst(A.cov[a * 3 + 2], wr[2],
                          fmaConst(A.cov[a * 3 + 1], wr[1], A.cov[a * 3 + 0] * wr[0]));
      }
      // translation x translation: d_i S(i, j) w d_j
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (COV == kCovGeneral || i <= j) accFma(dst[at(i, j)], d[i], Swd[j][i]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // translation x rotation: d_i (w S A)(i, c); and its transpose side under a general S:
        // sum_a A(a, c) S(a, j) w d_j
#pragma unroll
        for (int i = 0; i < 3; ++i) accFma(dst[at(i, 3 + c)], d[i], wSA[c][i]);
        if constexpr (COV == kCovGeneral) {
#pragma unroll
          for (int j = 0; j < 3; ++j) accDot3(dst[at(3 + c, j)], Acol[c], Swd[j]);
        }
        // rotation x rotation: sum_a A(a, c2) (w S A)(a, c)
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2)
          if (COV == kCovGeneral || c2 <= c) accDot3(dst[at(3 + c2, 3 + c)], Acol[c2], wSA[c]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) accFma(dst[kB + i], d[i], wSr[i]);
#pragma unroll
      for (int c = 0; c < 3; ++c) accDot3(dst[kB + 3 + c], Acol[c], wSr);
    }
    dst[NACC - 1] += rr;
  };

  // fp32 under the identity or a symmetric covariance: pairs of points in packed arithmetic, their
  // sums kept in fp32 per lane and pair slot and promoted to fp64 once, after the sweep (a slot adds
  // 1 / (2 x lanes of the grid) of the points — 38 of 10 M; the reference's float instantiation adds
  // all of them in fp32; promoting in between would keep the 28 fp64 accumulators alive in the loop,
  // 56 registers that push the kernel past 256)
  constexpr bool kPairs = sizeof(S) == 4 && COV != kCovGeneral;
  PointPair pair_sums[kPairs ? NACC : 1];
#pragma unroll
  for (int k = 0; k < (kPairs ? NACC : 1); ++k) pair_sums[k] = PointPair(0);
  auto sweep = [&](auto robust) {
    for_each_pack([&](const Pack<S>(&cur)[6], long long first) {
      if constexpr (V == 2) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          __builtin_amdgcn_sched_barrier(0);  // one point after the other
          const S in[6] = {cur[0].v[e], cur[1].v[e], cur[2].v[e], cur[3].v[e], cur[4].v[e], cur[5].v[e]};
          point(robust, in, first + e, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        if constexpr (kPairs) {
          // four fp32 points per pack, two at a time as PointPairs (identity / symmetric covariance)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            __builtin_amdgcn_sched_barrier(0);  // one pair after the other
            PointPair in[6];
#pragma unroll
            for (int pl = 0; pl < 6; ++pl)
              in[pl] = PointPair{cur[pl].v[2 * half], cur[pl].v[2 * half + 1]};
            point(robust, in, first + 2 * half, pair_sums);
          }
          __builtin_amdgcn_sched_barrier(0);
        } else {
          // four fp32 points per pack: as a real loop (unrolled they need > 256 registers whatever
          // the barriers say), the pack element picked by selects
#pragma unroll 1
          for (int e = 0; e < V; ++e) {
            S in[6];
#pragma unroll
            for (int pl = 0; pl < 6; ++pl) {
              in[pl] = cur[pl].v[0];
#pragma unroll
              for (int k = 1; k < V; ++k) in[pl] = (e == k) ? cur[pl].v[k] : in[pl];
            }
            point(robust, in, first + e, acc);
          }
        }
      }
    });
  };
  if (A.loss_kind == kLossGemanMcClure)
    sweep(std::true_type());
  else
    sweep(std::false_type());
  if constexpr (kPairs) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = double(pair_sums[k].x) + double(pair_sums[k].y);
  }
  blockReduceStore<NACC>(acc, out_row);
}

template <typename S, bool STREAMING, int COV, int HOME>
__device__ __forceinline__ void p2pForwardDiffBody(const S *tiles, int num_tiles,
                                                   const P2PSweepArgs<S> &A, int block,
                                                   int num_blocks,
                                                   const P2PSweepArgs<S> &in_memory) {
  constexpr int NACC = (COV == kCovGeneral) ? kAccFull : kAccSym;
  p2pForwardDiffRow<S, COV, HOME>(
      A, in_memory,
      [&](auto &&body) { sweepTiles<S, STREAMING>(tiles, num_tiles, body, block, num_blocks); },
      A.partials + size_t(block) * NACC);
}


}  // namespace
}  // namespace mopt
