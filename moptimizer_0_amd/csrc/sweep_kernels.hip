// gfx950 (CDNA4, MI355X) kernels of the per-residual linearization sweep.
//
// What is computed follows /root/reference/include/moptimizer/linearization.h:
//   for every index i:  r_i, J_i  ->  w_i = loss(|r_i|^2)
//                       H += w_i J_i^T S J_i ;  b += w_i J_i^T S r_i ;  sum += r_i^T r_i
// (:101-117 forward-difference form, :143-154 analytic form, :36-47 cost only), with the
// point-to-point model of tst/point2point.cpp:32-78 and the reprojection model of
// tst/camera_calibration.cpp:35-41 built in as device code in place of the per-point virtual
// calls.  How it is computed is laid out for the hardware:
//
//   * HBM-bound streaming reduction (48 B in, O(1) out per correspondence): no MFMA, no LDS
//     staging of inputs — every lane issues 16-byte loads straight to VGPRs, six per tile, all
//     six landing in one contiguous 24 KiB tile (layout in sweep.hpp), with the next tile's loads
//     issued before the current tile's arithmetic so each wave keeps 2 x 96 B per lane in flight.
//   * 64-wide wavefronts: per-thread register accumulators -> one LDS transpose per workgroup
//     (every lane stores each value once, 32 x 8 threads add them up, three xor-shuffles finish;
//     blockReduceStore) -> one partial row per workgroup in HBM.
//   * A fixed grid (workgroups per CU x CUs) strides over tiles, so the summation tree — and
//     therefore the result — is deterministic for a given device and count.
//   * A second tiny kernel adds the workgroup partials in a fixed order and writes
//     H (column-major) | b | sum_sq; no atomics anywhere.
#include "sweep_device.hpp"
#include "fd_device.hpp"
#include "lm_device.hpp"

namespace mopt {
namespace {

// Which entries of a model's Jacobian are the literal zero, whatever the data: products with them
// add nothing to any sum (for finite data not even in the last bit), so the accumulation below
// leaves them out — the point-to-point patterns are half zeros, and an H entry none of whose terms
// survives is never touched.  (A product with a literal 1 the compiler folds by itself; x * 0 it may
// not.)
struct DensePattern {
  static constexpr bool zero(int, int) { return false; }
};
template <int JAC>
struct P2PJacobianPattern {
  static constexpr bool zero(int a, int j) {
    if (JAC == kJacAnalyticTst) {
      // {1,0,0,0,1,0}, {0,0,1,0,-z,y}, {z,0,-x,-y,x,0}  (tst/point2point.cpp:71-75 read row-major)
      return (a == 0 && (j == 1 || j == 2 || j == 3 || j == 5)) ||
             (a == 1 && (j == 0 || j == 1 || j == 3)) || (a == 2 && (j == 1 || j == 5));
    }
    if (j < 3) return a != j;                   // [ I3 | . ]
    if (JAC == kJacAnalyticRight) return false;  // -R skew(p): dense
    return a == j - 3;                           // -skew(.): zero diagonal
  }
};

// acc += w J^T S J (upper triangle, or all n*n entries when S is not symmetric), w J^T S r, r^T r.
// J is m x n (row index = output), cov row-major m x m.  NACC selects the n*n form.  In fp64 every
// sum is a chain of fused multiply-adds straight into its accumulator.
// `Acc`: double, or float where the caller adds the points of a pack in fp32 first and promotes
// their sum (the fp32 point-to-point sweep: a conversion and an fp64 add per entry and point
// otherwise).
template <typename S, int M, int N, int COV, typename Pattern = DensePattern, typename Acc, int NACC>
__device__ __forceinline__ void accumulateDense(const S (&J)[M][N], const S (&r)[M], S w, S rr,
                                                const S *cov, Acc (&acc)[NACC]) {
  S SJ[M][N];
  S Sr[M];
  if (COV == kCovIdentity) {
#pragma unroll
    for (int a = 0; a < M; ++a) {
#pragma unroll
      for (int j = 0; j < N; ++j) SJ[a][j] = J[a][j];
      Sr[a] = r[a];
    }
  } else {
#pragma unroll
    for (int a = 0; a < M; ++a) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        S v = 0;
#pragma unroll
        for (int c = 0; c < M; ++c)
          if (!Pattern::zero(c, j)) v += cov[a * M + c] * J[c][j];
        SJ[a][j] = v;
      }
      S v = 0;
#pragma unroll
      for (int c = 0; c < M; ++c) v += cov[a * M + c] * r[c];
      Sr[a] = v;
    }
  }
  S wJ[M][N];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int i = 0; i < N; ++i) wJ[a][i] = Pattern::zero(a, i) ? S(0) : w * J[a][i];

  constexpr bool kFull = (NACC == N * N + N + 1);
  constexpr int kNH = kFull ? N * N : N * (N + 1) / 2;
  static_assert(NACC == kNH + N + 1, "accumulator count does not match the matrix form");
  // (S J)(a, j) is structurally zero only under the identity covariance
  constexpr bool kDirect = sizeof(S) == 8 || sizeof(Acc) == 4;  // one FMA per term into acc
  auto term = [&](Acc &dst, S &partial, S x, S y) {
    if constexpr (sizeof(Acc) == 4)
      dst = __builtin_fmaf(x, y, dst);
    else if constexpr (sizeof(S) == 8)
      dst = __builtin_fma(double(x), double(y), dst);
    else
      partial += x * y;
  };
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (!kFull && i > j) continue;
      const int k = kFull ? (j * N + i) : (j * (j + 1) / 2 + i);
      S partial = 0;
      bool any = false;
#pragma unroll
      for (int a = 0; a < M; ++a) {
        if (Pattern::zero(a, i) || (COV == kCovIdentity && Pattern::zero(a, j))) continue;
        term(acc[k], partial, wJ[a][i], SJ[a][j]);
        any = true;
      }
      if (!kDirect && any) acc[k] += double(partial);
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    S partial = 0;
#pragma unroll
    for (int a = 0; a < M; ++a) {
      if (Pattern::zero(a, i)) continue;
      term(acc[kNH + i], partial, wJ[a][i], Sr[a]);
    }
    if (!kDirect) acc[kNH + i] += double(partial);
  }
  acc[kNH + N] += Acc(rr);
}

// ---- point-to-point, literal evaluation ------------------------------------------------------
template <typename S, int JAC, int COV, typename Acc>
__device__ __forceinline__ void p2pPointLiteral(
    const P2PSweepArgs<S> &A, const S (&p)[3], const S (&q)[3], bool valid,
    Acc (&acc)[(COV == kCovGeneral) ? kAccFull : kAccSym]) {
  S r[3];
  p2pResidual<S>(A.T[0], p, q, r);
  S J[3][6];
  if (JAC == kJacAnalytic) {
    // [ I3 | -skew(p) ], row-major (model.h:35-42)
    const S z = S(0), o = S(1);
    const S Ja[3][6] = {{o, z, z, z, p[2], -p[1]}, {z, o, z, -p[2], z, p[0]}, {z, z, o, p[1], -p[0], z}};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 6; ++j) J[a][j] = Ja[a][j];
  } else if (JAC == kJacAnalyticTst) {
    // the same 18 numbers written column-major and read row-major (tst/point2point.cpp:71-75
    // against linearization.h:17-18)
    const S z = S(0), o = S(1);
    const S Jt[3][6] = {{o, z, z, z, o, z}, {z, z, o, z, -p[2], p[1]}, {p[2], z, -p[0], -p[1], p[0], z}};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 6; ++j) J[a][j] = Jt[a][j];
  } else if (JAC == kJacAnalyticLeft) {
    // derivative with respect to a left perturbation of the pose: [ I3 | -skew(R p + t) ], with
    // R p + t = r + q (levenberg_marquadt_dyn.cpp:82-83 "TODO Manifold operation")
    const S z = S(0), o = S(1);
    const S w0 = r[0] + q[0], w1 = r[1] + q[1], w2 = r[2] + q[2];
    const S Jl[3][6] = {{o, z, z, z, w2, -w1}, {z, o, z, -w2, z, w0}, {z, z, o, w1, -w0, z}};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 6; ++j) J[a][j] = Jl[a][j];
  } else if (JAC == kJacAnalyticRight) {
    // derivative with respect to R <- R Exp(phi), t <- t + rho (tst/manifold.cpp:47,
    // tst/state_model.cpp:28-34): [ I3 | -R skew(p) ]
    const S z = S(0), o = S(1);
    S Jr[3][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const S R0 = A.T[0][a * 4 + 0], R1 = A.T[0][a * 4 + 1], R2 = A.T[0][a * 4 + 2];
      Jr[a][0] = a == 0 ? o : z;
      Jr[a][1] = a == 1 ? o : z;
      Jr[a][2] = a == 2 ? o : z;
      Jr[a][3] = -(R1 * p[2] - R2 * p[1]);
      Jr[a][4] = -(R2 * p[0] - R0 * p[2]);
      Jr[a][5] = -(R0 * p[1] - R1 * p[0]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 6; ++j) J[a][j] = Jr[a][j];
  } else {
    // forward differences are p2pForwardDiffKernel's (7 transforms do not fit the scalar registers)
    static_assert(JAC != kJacNumeric, "forward differences: p2pForwardDiffKernel");
  }
  const S rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const S w = lossWeight<S>(A.loss_kind, A.loss_param, rr);
  accumulateDense<S, 3, 6, COV, P2PJacobianPattern<JAC>>(J, r, valid ? w : S(0), valid ? rr : S(0),
                                                          A.cov, acc);
}

template <typename S, int JAC, int COV, bool STREAMING = false>
__device__ __forceinline__ void p2pLinearizeLiteralBody(const P2PSweepArgs<S> &A, int block,
                                                        int num_blocks) {
  constexpr int NACC = (COV == kCovGeneral) ? kAccFull : kAccSym;
  constexpr int V = TileShape<S>::kVec;
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

  // fp32: the four points of a pack are added in fp32 and their sums promoted once per pack
  using Local = typename std::conditional<sizeof(S) == 8, double, float>::type;
  sweepTiles<S, STREAMING>(A.tiles, A.num_tiles, [&](const Pack<S>(&cur)[6], long long first) {
    Local loc[sizeof(S) == 8 ? 1 : NACC];
    if constexpr (sizeof(S) == 4) {
#pragma unroll
      for (int k = 0; k < NACC; ++k) loc[k] = 0.0f;
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const S p[3] = {cur[0].v[e], cur[1].v[e], cur[2].v[e]};
      const bool ok = isCorrespondence(first + e, A.count, cur[3].v[e]);
      const S q[3] = {ok ? cur[3].v[e] : S(0), ok ? cur[4].v[e] : S(0), ok ? cur[5].v[e] : S(0)};
      if constexpr (sizeof(S) == 8)
        p2pPointLiteral<S, JAC, COV>(A, p, q, ok, acc);
      else
        p2pPointLiteral<S, JAC, COV>(A, p, q, ok, loc);
    }
    if constexpr (sizeof(S) == 4) {
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] += double(loc[k]);
    }
  }, block, num_blocks);
  blockReduceStore<NACC>(acc, A.partials + size_t(block) * NACC);
}

// (tiles / num_tiles repeat the first members of A as preloaded leading arguments, as in the moments
// sweep; STREAMING: non-temporal loads once the data exceed the aggregate L2)
template <typename S, int JAC, int COV, bool STREAMING>
__global__ __launch_bounds__(kBlockThreads) void p2pLinearizeLiteralKernel(
    const S *tiles, int num_tiles, const P2PSweepArgs<S> A) {
  P2PSweepArgs<S> B = A;
  B.tiles = tiles;
  B.num_tiles = num_tiles;
  p2pLinearizeLiteralBody<S, JAC, COV, STREAMING>(B, blockIdx.x, gridDim.x);
}

// Resident form (device-resident LM, sweep.hpp): the per-x constants come from HBM, where the
// step kernel of the previous trial left them; a kernel queued past the end of the minimisation
// finds control->done set and returns.
template <typename S, int JAC, int COV>
__global__ __launch_bounds__(kBlockThreads) void p2pLinearizeLiteralResidentKernel(
    const P2PSweepArgs<S> *__restrict__ d_args, const LmControl *__restrict__ control) {
  if (control->done) return;
  const P2PSweepArgs<S> A = *d_args;
  p2pLinearizeLiteralBody<S, JAC, COV>(A, blockIdx.x, gridDim.x);
}

// The sweeps of several literally evaluated point2point costs of one problem in one launch (as
// reprojResidentSetKernel below): the loop of levenberg_marquadt_dyn.cpp:48-60 pays one launch per
// evaluated point, not one per cost.
template <typename S, int JAC, int COV>
__global__ __launch_bounds__(kBlockThreads) void p2pLinearizeLiteralResidentSetKernel(
    const ResidentSweepSet set, const LmControl *__restrict__ control) {
  if (control->done) return;
  const int k = costOfBlock(set);
  const P2PSweepArgs<S> A = *static_cast<const P2PSweepArgs<S> *>(set.args[k]);
  p2pLinearizeLiteralBody<S, JAC, COV>(A, int(blockIdx.x) - set.first_block[k],
                                       set.first_block[k + 1] - set.first_block[k]);
}

// ---- point-to-point, weighted moments --------------------------------------------------------
// Every Jacobian this model produces is affine in the source point p (analytic: [I | -skew(p)];
// forward differences: ((R_j - R) p + (t_j - t)) / h_j), so
//   sum_i w_i J_i^T S J_i  and  sum_i w_i J_i^T S r_i
// are linear in the 22 moments  sum w, sum w p, sum w p p^T, sum w r, sum w p r^T.  The sweep
// accumulates those (plus sum r^T r); finalizeMomentsKernel contracts them with the basis.
// `tiles` / `num_tiles` repeat the first members of A as leading scalar arguments: gfx950 can
// preload those into SGPRs at wave launch (-amdgpu-kernarg-preload-count), so the first tile's
// loads go out without waiting for a kernel-argument fetch.
// (momentsOfPack — the moments of the V correspondences one lane holds of a tile — is in sweep_device.hpp:
// the forward-difference translation unit sweeps with it too)
template <typename S, bool STREAMING>
__device__ __forceinline__ void p2pMomentsBody(const S *tiles, int num_tiles,
                                               const P2PSweepArgs<S> &A) {
  double acc[kAccMoments];
#pragma unroll
  for (int k = 0; k < kAccMoments; ++k) acc[k] = 0.0;
  sweepTiles<S, STREAMING>(tiles, num_tiles, [&](const Pack<S>(&cur)[6], long long first) {
    momentsOfPack<S>(acc, cur, first, A);
  });
  blockReduceStore<kAccMoments>(acc, A.partials + size_t(blockIdx.x) * kAccMoments);
}

template <typename S, bool STREAMING>
__global__ __launch_bounds__(kBlockThreads) void p2pMomentsKernel(const S *tiles, int num_tiles,
                                                                  const P2PSweepArgs<S> A) {
  p2pMomentsBody<S, STREAMING>(tiles, num_tiles, A);
}

template <typename S, bool STREAMING>
__global__ __launch_bounds__(kBlockThreads) void p2pMomentsResidentKernel(
    const S *tiles, int num_tiles, const P2PSweepArgs<S> *__restrict__ d_args,
    const LmControl *__restrict__ control) {
  if (control->done) return;
  const P2PSweepArgs<S> A = *d_args;
  p2pMomentsBody<S, STREAMING>(tiles, num_tiles, A);
}

// ---- point-to-point, cost only (linearization.h:36-63) ----------------------------------------
template <typename S, bool STREAMING>
__global__ __launch_bounds__(kBlockThreads) void p2pCostKernel(const S *tiles, int num_tiles,
                                                               const P2PSweepArgs<S> A) {
  constexpr int V = TileShape<S>::kVec;
  double acc[1] = {0.0};
  sweepTiles<S, STREAMING>(tiles, num_tiles, [&](const Pack<S>(&cur)[6], long long first) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const S p[3] = {cur[0].v[e], cur[1].v[e], cur[2].v[e]};
      const bool valid = isCorrespondence(first + e, A.count, cur[3].v[e]);
      const S q[3] = {valid ? cur[3].v[e] : S(0), valid ? cur[4].v[e] : S(0),
                      valid ? cur[5].v[e] : S(0)};
      S r[3];
      p2pResidual<S>(A.T[0], p, q, r);
      const S rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      acc[0] += valid ? double(rr) : 0.0;
    }
  });
  blockReduceStore<1>(acc, A.partials + blockIdx.x);
}

// ---- reprojection (camera calibration), fp64, forward differences ------------------------------
__device__ __forceinline__ void reprojResidual(const double (&Mx)[12], const double (&P)[4],
                                               double u, double v, double (&r)[2]) {
  // as written, no fused multiply-adds: the CPU restatement evaluates the same four products and
  // three sums, and forward differences amplify a last-bit difference of a residual by eps / h_j
#pragma clang fp contract(off)
  double o[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    o[a] = ((Mx[a * 4 + 0] * P[0] + Mx[a * 4 + 1] * P[1]) + Mx[a * 4 + 2] * P[2]) +
           Mx[a * 4 + 3] * P[3];
  r[0] = u - (o[0] / o[2]);  // tst/camera_calibration.cpp:38
  r[1] = v - (o[1] / o[2]);  // :39
}

// Seven 3x4 projections at x and x + h_j e_j are 84 fp64 constants: as kernel arguments they asked
// for 180 scalar registers against ~100, and every reproj* linearize kernel spilled 128-148 of them
// (v_readlane / v_writelane around each use; rounds 1-3).  As in fd_kernels.hip the six perturbed
// projections live in LDS: 72 lanes copy them once per workgroup from where the arguments lie in
// memory (`in_memory`: the kernel-argument segment, or the resident forms' block in HBM), and each is
// read (same-address reads: broadcasts) while the one before it is being used.  The projection at x,
// the step reciprocals, the covariance and the loss stay in scalar registers (~50).  The tile's loads
// are issued before that copy, so the two latencies overlap.  Per element the arithmetic is the
// reference's: seven residuals with both divisions each, twelve quotients (linearization.h:101-107),
// contraction off.
//
// One element per lane, 256 per tile: the arithmetic of an element is one long dependent chain
// (≈ 400 fp64 instructions, 14 divisions), so what a sweep of BASELINE config 5 costs is that chain
// once per workgroup — with two elements per lane (tiles of 512, rounds 1-3) it ran twice, or
// interleaved in > 300 registers; measured in one process (scripts/probes/reproj_stamps.cpp,
// profiles/r4_reproj_forms.txt): 4.8 us per 40 k / 60 k elements against 5.7 (interleaved) and 6.1
// (one after the other) where an empty launch of the same grid takes 4.2, and 134 against 138 and
// 151 us at 4 M elements.
// block / num_blocks: this workgroup's place among those sweeping this cost (a launch may carry
// the workgroups of several costs: reprojResidentSetKernel); the cost's partial rows are num_blocks.
template <int COV, bool COST_ONLY>
__device__ __forceinline__ void reprojBody(const ReprojSweepArgs &A, int block, int num_blocks,
                                           const ReprojSweepArgs &in_memory) {
  constexpr int NACC = COST_ONLY ? 1 : ((COV == kCovGeneral) ? kAccFull : kAccSym);
  MOPT_STAMP(0);
  __shared__ double Mlds[kNumParams][12];
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

  struct Element {
    double P[4], u, v;
  };
  auto loadElement = [&](int tile) {
    const unsigned char *tb = A.tiles + size_t(tile) * kReprojTileBytes;
    const double *planes = reinterpret_cast<const double *>(tb) + threadIdx.x;
    const int2 px = reinterpret_cast<const int2 *>(tb + size_t(4) * kReprojTilePoints * 8)[threadIdx.x];
    Element el;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) el.P[pl] = planes[pl * kReprojTilePoints];
    el.u = double(px.x);
    el.v = double(px.y);
    return el;
  };

  // this workgroup's tiles: block, block + num_blocks, ...; the next one is loaded before the
  // arithmetic of the current one (unconditionally — past the end the last one again, an L2 hit:
  // a conditional prefetch makes the compiler wait for everything at the join, sweep_device.hpp)
  int tile = block;
  const int last = tile < A.num_tiles ? tile + ((A.num_tiles - 1 - tile) / num_blocks) * num_blocks : tile;
  Element el{};
  if (tile < A.num_tiles) el = loadElement(tile);
  if constexpr (!COST_ONLY) {
    // the one access indexed by lane goes to memory, not to a by-value copy (which would put the
    // whole struct into every lane's scratch)
    if (threadIdx.x < kNumParams * 12) (&Mlds[0][0])[threadIdx.x] = (&in_memory.M[1][0])[threadIdx.x];
    __syncthreads();
  }
  MOPT_STAMP(1);
  // `robust`: the loss kind is taken out of the element (a branch inside splits the basic block, and
  // the compiler then sinks the seven residuals below it, behind the LDS reads of all six projections)
  auto sweep = [&](auto robust) {
    for (; tile < A.num_tiles; tile += num_blocks) {
      const bool valid = (long long)tile * kReprojTilePoints + threadIdx.x < A.count;
      const Element nx = loadElement(tile + num_blocks <= last ? tile + num_blocks : last);
      double r[2];
      reprojResidual(A.M[0], el.P, el.u, el.v, r);
      const double rr = r[0] * r[0] + r[1] * r[1];
      MOPT_STAMP(2);
      if constexpr (COST_ONLY) {
        acc[0] += valid ? rr : 0.0;
      } else {
        double J[2][6];
        // one perturbed projection after the other (all seven residuals interleaved need > 256
        // registers), each read from LDS while the one before it is being used
        double Mj[2][12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Mj[0][k] = Mlds[0][k];
#pragma unroll
        for (int j = 0; j < kNumParams; ++j) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" ::: "memory");  // read here, not hoisted into 144 registers
          if (j + 1 < kNumParams) {
#pragma unroll
            for (int k = 0; k < 12; ++k) Mj[(j + 1) & 1][k] = Mlds[j + 1][k];
          }
          double rp[2];
          reprojResidual(Mj[j & 1], el.P, el.u, el.v, rp);
          J[0][j] = (rp[0] - r[0]) * A.inv_h[j];
          J[1][j] = (rp[1] - r[1]) * A.inv_h[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        MOPT_STAMP(3);
        // padded slots repeat the last element (relayoutReprojKernel): finite, left out by w = 0 —
        // a branch around the accumulation lets the compiler sink all seven residuals into it, behind
        // the LDS reads of all six projections (144 registers)
        double w = 1.0;
        if constexpr (decltype(robust)::value) w = lossWeight<double>(kLossGemanMcClure, A.loss_param, rr);
        accumulateDense<double, 2, 6, COV>(J, r, valid ? w : 0.0, valid ? rr : 0.0, A.cov, acc);
      }
      MOPT_STAMP(4);
      el = nx;
    }
  };
  if (!COST_ONLY && A.loss_kind == kLossGemanMcClure)
    sweep(std::true_type());
  else
    sweep(std::false_type());
  MOPT_STAMP(5);
  blockReduceStore<NACC>(acc, A.partials + size_t(block) * NACC);
  MOPT_STAMP(6);
}

template <int COV, bool COST_ONLY>
__global__ __launch_bounds__(kBlockThreads) void reprojKernel(const ReprojSweepArgs A) {
  reprojBody<COV, COST_ONLY>(A, blockIdx.x, gridDim.x, A);
}

template <int COV>
__global__ __launch_bounds__(kBlockThreads) void reprojResidentKernel(
    const ReprojSweepArgs *__restrict__ d_args, const LmControl *__restrict__ control) {
  if (control->done) return;
  const ReprojSweepArgs A = *d_args;
  reprojBody<COV, false>(A, blockIdx.x, gridDim.x, *d_args);
}

// The sweeps of several reprojection costs of one problem in one launch (device-resident LM):
// workgroups [first_block[k], first_block[k + 1]) sweep cost k.  Each cost alone fills part of the
// chip, behind one another they would also pay a launch boundary each.
template <int COV>
__global__ __launch_bounds__(kBlockThreads) void reprojResidentSetKernel(
    const ResidentSweepSet set, const LmControl *__restrict__ control) {
  if (control->done) return;
  const int k = costOfBlock(set);
  const ReprojSweepArgs *d_args = static_cast<const ReprojSweepArgs *>(set.args[k]);
  const ReprojSweepArgs A = *d_args;
  reprojBody<COV, false>(A, int(blockIdx.x) - set.first_block[k],
                         set.first_block[k + 1] - set.first_block[k], *d_args);
}

// ---- small parametric models over per-element scalar data ------------------------------------
// The other models the reference's tests run through the same cost classes (n != 6):
//   ExpCurve   y - exp(x0 t + x1)            tst/curve_fitting.cpp:81-98, multiple_objectives.cpp:81-98
//   Rational   y - x0 t / (x1 + t)           tst/test_models.h:7-20; Jacobian tst/differentiation.cpp:26-38
//   Powell     Powell's singular function    tst/powell.cpp:21-60 (4 outputs, one element)
// Same skeleton: per-element residual, analytic or forward-difference Jacobian
// (linearization.h:101-117 / :143-154), loss weight, dense accumulation, workgroup partial row.
// accept() is what IBaseModel::f / f_df return (model.h:32,43): false = "this index is not a residual",
// and the sweep skips it (linearization.h:102,144).  The *Marked forms of the models over observations
// (t, y) say so for an observation whose y is NaN — the marker the point2point sweeps use for a source
// without a correspondence — and replace that y by a harmless one, so that everything computed for the
// skipped index stays finite and a weight of zero is enough to leave it out (one select on the way
// in, not one per residual and Jacobian entry).  The check costs 5 % of a sweep that is bound by its
// arithmetic (10 M elements: 40.3 against 38.4 us), so data without markers run the forms without it.
template <typename S>
__device__ __forceinline__ bool acceptUnlessMarked(S (&d)[2]) {
  const bool ok = d[1] == d[1];
  d[1] = ok ? d[1] : S(0);
  return ok;
}

template <typename S>
struct ExpCurve {
  static constexpr int N = 2, M = 1, D = 2;
  static constexpr bool kHasJacobian = false;
  __device__ static bool accept(S (&)[D]) { return true; }
  __device__ static void residual(const S *x, const S *d, S (&r)[M]) {
#pragma clang fp contract(off)  // as written: the checker's arithmetic (forward differences amplify an ulp)
    r[0] = d[1] - exp(x[0] * d[0] + x[1]);
  }
  __device__ static void jacobian(const S *, const S *, S (&)[M][N]) {}
};
template <typename S>
struct ExpCurveMarked : ExpCurve<S> {
  __device__ static bool accept(S (&d)[2]) { return acceptUnlessMarked<S>(d); }
};

template <typename S>
struct Rational {
  static constexpr int N = 2, M = 1, D = 2;
  static constexpr bool kHasJacobian = true;
  __device__ static bool accept(S (&)[D]) { return true; }
  __device__ static void residual(const S *x, const S *d, S (&r)[M]) {
#pragma clang fp contract(off)  // as written: the checker's arithmetic (forward differences amplify an ulp)
    r[0] = d[1] - (x[0] * d[0]) / (x[1] + d[0]);
  }
  __device__ static void jacobian(const S *x, const S *d, S (&J)[M][N]) {
    const S denominator = x[1] + d[0];
    J[0][0] = -d[0] / denominator;
    J[0][1] = (x[0] * d[0]) / (denominator * denominator);
  }
};
template <typename S>
struct RationalMarked : Rational<S> {
  __device__ static bool accept(S (&d)[2]) { return acceptUnlessMarked<S>(d); }
};

template <typename S>
struct Powell {
  static constexpr int N = 4, M = 4, D = 0;
  static constexpr bool kHasJacobian = true;
  __device__ static bool accept(S (&)[1]) { return true; }
  __device__ static void residual(const S *x, const S *, S (&r)[M]) {
#pragma clang fp contract(off)  // as written: the checker's arithmetic (forward differences amplify an ulp)
    r[0] = x[0] + 10 * x[1];
    r[1] = sqrt(S(5)) * (x[2] - x[3]);
    r[2] = (x[1] - 2 * x[2]) * (x[1] - 2 * x[2]);
    r[3] = sqrt(S(10)) * (x[0] - x[3]) * (x[0] - x[3]);
  }
  // exactly the entries the reference's test model writes (tst/powell.cpp:31-57), including its
  // (x1 + 2 x2) in rows 2 — the test's own Jacobian, not the textbook one
  __device__ static void jacobian(const S *x, const S *, S (&J)[M][N]) {
    const S s5 = sqrt(S(5)), s10 = sqrt(S(10));
    J[0][0] = 1;  J[1][0] = 0;   J[2][0] = 0;                              J[3][0] = s10 * 2 * (x[0] - x[3]);
    J[0][1] = 10; J[1][1] = 0;   J[2][1] = 2 * (x[1] + 2 * x[2]);          J[3][1] = 0;
    J[0][2] = 0;  J[1][2] = s5;  J[2][2] = 2 * (x[1] + 2 * x[2]) * (-2);   J[3][2] = 0;
    J[0][3] = 0;  J[1][3] = -s5; J[2][3] = 0;                              J[3][3] = s10 * 2 * (x[0] - x[3]) * (-1);
  }
};

// The skeleton of the tiled sweeps (and of the run-time compiled twin, jit_model.cpp): a lane takes
// the V = 16 / sizeof(S) consecutive elements of one 16-byte pack per data plane and step, the next
// step's packs are requested before this one's arithmetic starts, the loss kind is taken out of the
// element, products go into the sums as fused multiply-adds (accumulateDense).  Rounds 1-3 loaded
// 8 bytes per lane and plane in a plain grid-stride loop: 52 us for the forward-difference sweep of
// the exp curve over 10 M elements against 37 us for the same model compiled at run time.
template <typename S, template <typename> class ModelT, int JAC, int COV, bool COST_ONLY>
__device__ __forceinline__ void scalarModelBody(const ScalarSweepArgs<S> &A, int block,
                                                int num_blocks) {
  using Model = ModelT<S>;
  constexpr int N = Model::N, M = Model::M, D = Model::D;
  constexpr int V = TileShape<S>::kVec;
  constexpr bool kNumeric = JAC == kJacNumeric || !Model::kHasJacobian;
  constexpr int NACC =
      COST_ONLY ? 1 : ((COV == kCovGeneral) ? N * N + N + 1 : N * (N + 1) / 2 + N + 1);
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  // (r+ - r) / h_j (linearization.h:105) as a product with 1 / h_j formed once per sweep, like the
  // hand-written point2point sweeps: an fp64 division per Jacobian entry and element otherwise
  S inv_h[N];
#pragma unroll
  for (int j = 0; j < N; ++j) inv_h[j] = S(1) / A.h[j];
  // `robust`: the loss kind is a property of the sweep (a branch inside the element splits its basic
  // block and lets the compiler sink one element's accumulation below the next one's residuals)
  auto sweep = [&](auto robust) {
    auto element = [&](S (&d)[D > 0 ? D : 1], bool valid) {
      // an index the model rejects is skipped like a slot past the end: weight zero (its data made
      // harmless by accept())
      valid = Model::accept(d) && valid;
      S r[M];
      Model::residual(A.x, d, r);
      S rr = 0;
#pragma unroll
      for (int a = 0; a < M; ++a) rr += r[a] * r[a];
      rr = valid ? rr : S(0);
      if constexpr (COST_ONLY) {
        acc[0] += double(rr);
      } else {
        S J[M][N];
        if constexpr (kNumeric) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            S xp[N];
#pragma unroll
            for (int k = 0; k < N; ++k) xp[k] = A.x[k];
            xp[j] += A.h[j];  // linearization.h:89
            S rp[M];
            Model::residual(xp, d, rp);
#pragma unroll
            for (int a = 0; a < M; ++a) J[a][j] = (rp[a] - r[a]) * inv_h[j];  // :105
          }
        } else {
          Model::jacobian(A.x, d, J);
        }
        S w = S(1);
        if constexpr (decltype(robust)::value) w = lossWeight<S>(kLossGemanMcClure, A.loss_param, rr);
        accumulateDense<S, M, N, COV>(J, r, valid ? w : S(0), rr, A.cov, acc);
      }
    };
    const long long step = (long long)num_blocks * kBlockThreads * V;
    long long i = ((long long)block * kBlockThreads + threadIdx.x) * V;
    Pack<S> cur[D > 0 ? D : 1], nxt[D > 0 ? D : 1];
    // (a run-time choice between these and non-temporal loads — one uniform branch per pack — cost
    // 6 us of the 38 at 10 M elements, whichever way it went: profiles/r4_jit_timing.txt)
    auto load = [&](const S *from) { return loadPack<S>(from); };
    if (i < A.count) {
#pragma unroll
      for (int p = 0; p < D; ++p) nxt[p] = load(A.data + p * A.stride + i);
    }
    for (; i < A.count; i += step) {
#pragma unroll
      for (int p = 0; p < D; ++p) cur[p] = nxt[p];
      // the next step's packs are requested before this one's arithmetic starts (past the end: this
      // step's again — an unconditional load, see sweepTiles)
      const long long ahead = i + step < A.count ? i + step : i;
#pragma unroll
      for (int p = 0; p < D; ++p) nxt[p] = load(A.data + p * A.stride + ahead);
      // a slot past the end (the zero-padded tail of the last pack) is evaluated on the pack's first
      // element, which is in range, and enters every sum with weight zero
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if constexpr (kNumeric && !COST_ONLY) __builtin_amdgcn_sched_barrier(0);  // one element after the other
        const bool valid = i + e < A.count;
        S d[D > 0 ? D : 1];
#pragma unroll
        for (int p = 0; p < D; ++p) d[p] = valid ? cur[p].v[e] : cur[p].v[0];
        element(d, valid);
      }
      if constexpr (kNumeric && !COST_ONLY) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (!COST_ONLY && A.loss_kind == kLossGemanMcClure)
    sweep(std::true_type());
  else
    sweep(std::false_type());
  blockReduceStore<NACC>(acc, A.partials + size_t(block) * NACC);
}

template <typename S, template <typename> class ModelT, int JAC, int COV, bool COST_ONLY>
__global__ __launch_bounds__(kBlockThreads) void scalarModelKernel(const ScalarSweepArgs<S> A) {
  scalarModelBody<S, ModelT, JAC, COV, COST_ONLY>(A, blockIdx.x, gridDim.x);
}

template <typename S, template <typename> class ModelT, int JAC, int COV>
__global__ __launch_bounds__(kBlockThreads) void scalarModelResidentKernel(
    const ScalarSweepArgs<S> *__restrict__ d_args, const LmControl *__restrict__ control) {
  if (control->done) return;
  const ScalarSweepArgs<S> A = *d_args;
  scalarModelBody<S, ModelT, JAC, COV, false>(A, blockIdx.x, gridDim.x);
}

// several costs over the same built-in model in one launch (tst/multiple_objectives.cpp:102-132
// splits the curve fit's 67 observations over two costs)
template <typename S, template <typename> class ModelT, int JAC, int COV>
__global__ __launch_bounds__(kBlockThreads) void scalarModelResidentSetKernel(
    const ResidentSweepSet set, const LmControl *__restrict__ control) {
  if (control->done) return;
  const int k = costOfBlock(set);
  const ScalarSweepArgs<S> A = *static_cast<const ScalarSweepArgs<S> *>(set.args[k]);
  scalarModelBody<S, ModelT, JAC, COV, false>(A, int(blockIdx.x) - set.first_block[k],
                                              set.first_block[k + 1] - set.first_block[k]);
}

// ---- correspondence search (ICP update step) ---------------------------------------------------
// One thread per source point: warp it with the current pose, look through the grid cells around
// it — the 2 x 2 x 2 nearest first, the wave in lock step, then what of the (2 reach + 1)^3 block
// the best so far still admits —, keep the nearest target within the maximum distance, and write
// that target into the target planes of the point's slot (or the NaN marker).  The reference leaves
// this step to the user model's update(x) (model.h:24-26; "setup can be i.e nearest neighboor
// search", docs/Cost.puml:14-17) and ships no implementation, so semantics are defined here: exact
// nearest neighbour in the Euclidean metric, ties resolved to the first candidate in (cell z, y, x;
// original index) order.

// candidates per lane and trip of the first round (4, 6 and 8 measure the same; 2 is slower)
constexpr int kIcpTrip = 4;

template <typename S, int TRIP>
__device__ __forceinline__ void icpMatchBody(const IcpMatchArgs<S> &A, const S (&T)[12]) {
  constexpr int TP = TileShape<S>::kPoints;
  const long long i = (long long)blockIdx.x * kBlockThreads + threadIdx.x;
  if (i >= (long long)A.num_tiles * TP) return;  // (whole workgroups: TP is a multiple of their size)
  S *slot = A.tiles + (i / TP) * TileShape<S>::kP2PScalars + (i % TP);
  // No lane leaves before the end: the first round below runs in lock step over the wave, every
  // load from an address that exists whatever the lane holds (slots past the count are padding).
  const S p[3] = {slot[0 * TP], slot[1 * TP], slot[2 * TP]};
  S w[3], g[3];
  bool inside = i < A.count;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    w[a] = ((T[a * 4 + 0] * p[0] + T[a * 4 + 1] * p[1]) + T[a * 4 + 2] * p[2]) + T[a * 4 + 3];
    g[a] = floor((w[a] - A.origin[a]) * A.inv_cell);
    inside = inside && g[a] >= S(-A.reach) && g[a] <= S(A.dims[a] - 1 + A.reach);
  }
  const int c[3] = {inside ? int(g[0]) : 0, inside ? int(g[1]) : 0, inside ? int(g[2]) : 0};
  bool found = false;
  S best_d = A.max_dist2;
  // Ties go to the candidate stored first — (cell z, y, x; original index) order — whatever order
  // the cells are visited in, and however often: the position k is part of the comparison.
  int best_k = 0x7fffffff;
  // In the first round the winner's coordinates travel with it (reading them at the end would be
  // one more dependent round trip for every wave); the second round tracks distance and position
  // only and reads the coordinates of a winner it found once, at its end: six conditional moves
  // less per candidate where candidates are many.
  S best[3] = {S(0), S(0), S(0)};
  auto consider = [&](const S (&q)[3], int k, bool present, auto keep_coordinates) {
    const S d0 = w[0] - q[0], d1 = w[1] - q[1], d2 = w[2] - q[2];
    const S dist = d0 * d0 + d1 * d1 + d2 * d2;
    const bool better = present && dist <= A.max_dist2 &&
                        (!found || dist < best_d || (dist == best_d && k < best_k));
    found = found || better;
    best_d = better ? dist : best_d;
    best_k = better ? k : best_k;
    if constexpr (decltype(keep_coordinates)::value) {
#pragma unroll
      for (int a = 0; a < 3; ++a) best[a] = better ? q[a] : best[a];
    }
  };
  auto fetch = [&](int k, S (&q)[3]) {
    const Pack<S> *cand = reinterpret_cast<const Pack<S> *>(A.sorted + size_t(k) * 4);
    if (sizeof(S) == 8) {
      const Pack<S> lo = cand[0], hi = cand[1];
      q[0] = lo.v[0]; q[1] = lo.v[1]; q[2] = hi.v[0];
    } else {
      const Pack<S> all = cand[0];
      q[0] = all.v[0]; q[1] = all.v[1]; q[2] = all.v[2];
    }
  };
  // Offsets inside the own cell, [0, cell).  They are recomputed from the cell index and can
  // disagree with the binning's floor((w - origin) / cell) by a few ulps OF THE COORDINATE (not of
  // the offset): every gap to a face is shortened by that much before it is squared, in the scalar
  // type's own epsilon (targets were binned with floor()).
  constexpr S kUlps = S(16) * std::numeric_limits<S>::epsilon();
  const S fx = w[0] - (A.origin[0] + S(c[0]) * A.cell);
  const S fy = w[1] - (A.origin[1] + S(c[1]) * A.cell);
  const S fz = w[2] - (A.origin[2] + S(c[2]) * A.cell);
  const S ex = kUlps * (fabs(w[0]) + fabs(A.origin[0]) + A.cell);
  const S ey = kUlps * (fabs(w[1]) + fabs(A.origin[1]) + A.cell);
  const S ez = kUlps * (fabs(w[2]) + fabs(A.origin[2]) + A.cell);
  auto within = [&](S bound) {
    return bound * (S(1) - kUlps) <= (found ? best_d : A.max_dist2);
  };
  auto sq = [](S v) { return v > S(0) ? v * v : S(0); };

  // First round, the wave in lock step: the 2 x 2 x 2 cells nearest to the point (per axis the own
  // cell and the neighbour across the nearer face) hold every target closer than half a cell.
  // They are four runs of two consecutive cells: eight range bounds fetched together, then the
  // lane's candidates out of the four runs as ONE list, four per trip — every trip is eight (fp32:
  // four) loads in flight and one wait, and the wave makes as many trips as its longest list
  // needs.  Walking the rows one after the other instead, each lane under its own pruning, the
  // wave executed the union of its lanes' visits: ~116 vector loads per wave with an eighth of the
  // lanes active in each, ~20 dependent waits.
  const S half = S(0.5) * A.cell;
  const int sx = fx < half ? -1 : 0, sy = fy < half ? -1 : 0, sz = fz < half ? -1 : 0;
  int first[4], ends[4];
  {
    int xa = c[0] + sx, xb = xa + 1;
    xa = xa < 0 ? 0 : xa;
    xb = xb >= A.dims[0] ? A.dims[0] - 1 : xb;
    int at_a[4], at_b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = c[1] + sy + (r & 1), z = c[2] + sz + (r >> 1);
      const bool ok = inside && xa <= xb && y >= 0 && y < A.dims[1] && z >= 0 && z < A.dims[2];
      const int row = (z * A.dims[1] + y) * A.dims[0];  // the grid has at most 2^28 cells (icp.cpp)
      at_a[r] = ok ? row + xa : 0;  // (entry 0 twice: an empty run)
      at_b[r] = ok ? row + xb + 1 : 0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      first[r] = A.cell_start[at_a[r]];
      ends[r] = A.cell_start[at_b[r]];
    }
  }
  const int n0 = ends[0] - first[0];
  const int n1 = n0 + (ends[1] - first[1]);
  const int n2 = n1 + (ends[2] - first[2]);
  const bool first_round = TRIP > 0;
  const int total = first_round ? n2 + (ends[3] - first[3]) : 0;
  constexpr int kSlots = TRIP > 0 ? TRIP : 1;
  for (int j = 0; __any(j < total); j += kSlots) {
    int k[kSlots];
    bool present[kSlots];
    S q[kSlots][3];
#pragma unroll
    for (int u = 0; u < kSlots; ++u) {
      const int jj = j + u;
      present[u] = jj < total;
      // position jj of the four runs laid end to end (selects, no branches)
      int at = first[3] + (jj - n2);
      at = jj < n2 ? first[2] + (jj - n1) : at;
      at = jj < n1 ? first[1] + (jj - n0) : at;
      at = jj < n0 ? first[0] + jj : at;
      k[u] = present[u] ? at : 0;  // (a trip is only made when some lane has a candidate)
    }
#pragma unroll
    for (int u = 0; u < kSlots; ++u) fetch(k[u], q[u]);
#pragma unroll
    for (int u = 0; u < kSlots; ++u) consider(q[u], k[u], present[u], std::true_type());
  }

  // Everything else lies across a far face of that block: only a lane whose best so far (or the
  // maximum distance, while it has none) reaches the nearest of the three goes on — at about one
  // target per cell and a source within half a cell of its target, none.
  const S far_x = (sx < 0 ? A.cell - fx : fx) - ex;
  const S far_y = (sy < 0 ? A.cell - fy : fy) - ey;
  const S far_z = (sz < 0 ? A.cell - fz : fz) - ez;
  const bool go = inside && (!first_round || within(sq(fmin(far_x, fmin(far_y, far_z)))));
  const int first_round_k = best_k;
  // Second round, in lock step too (round 5; rounds 3-4 walked the rows one after the other, each
  // lane under its running bound: fewer candidates per lane, but two dependent waits of the whole
  // wave for every row ANY of its lanes visited).  The rows of the (2 reach + 1)^2 block around the
  // own one are taken nine (then eight) at a time: every lane that goes on lists, under the bound
  // it has now, the stretch of cells of each row it still has to see — a stretch is consecutive in
  // storage: one candidate range, two bounds —, the 18 bounds are fetched together, and the wave
  // walks the stretches laid end to end four candidates per trip like the first round: one wait
  // for the bounds and one per trip, every lane that has something active in every trip.  A cell is
  // listed only if its box can hold something at least as close as what has been found (or within
  // the maximum distance while nothing has); the bound is taken from the own cell's faces and
  // tightens from one group of rows to the next.
  const S gx[2] = {fx - ex, (A.cell - fx) - ex};
  const S gy[2] = {fy - ey, (A.cell - fy) - ey};
  const S gz[2] = {fz - ez, (A.cell - fz) - ez};
  const int R = A.reach;
  auto gapAlong = [&](const S (&gap)[2], int d) {  // to the slab of cells d steps away
    return d == 0 ? S(0) : gap[d < 0 ? 0 : 1] + S((d < 0 ? -d : d) - 1) * A.cell;
  };
  // rows(r, dy, dz) names row r < 9 of the group (the same for every lane of the wave) or says there
  // is none; skip_block: the rows of the 2 x 2 x 2 block have only their far cell left (reach 1)
  auto walkRows = [&](bool active, auto rows, auto skip_block) {
    int at_a[9], at_b[9];
    bool some = false;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      int dy = 0, dz = 0;
      const bool valid = rows(r, dy, dz);
      const int y = c[1] + dy, z = c[2] + dz;
      const S yz = sq(gapAlong(gy, dy)) + sq(gapAlong(gz, dz));
      int left = 0, right = 0;
      for (int step = 0; step < R; ++step) {
        left += (left == step && within(yz + sq(gx[0] + S(step) * A.cell))) ? 1 : 0;
        right += (right == step && within(yz + sq(gx[1] + S(step) * A.cell))) ? 1 : 0;
      }
      bool ok = valid && active && within(yz) && y >= 0 && y < A.dims[1] && z >= 0 && z < A.dims[2];
      int xa = c[0] - left, xb = c[0] + right;
      if constexpr (decltype(skip_block)::value) {
        // cells c0 + sx and c0 + sx + 1 of a row of the block were candidates of the first round
        const bool in_block = first_round && (dy == sy || dy == sy + 1) && (dz == sz || dz == sz + 1);
        xa = in_block ? (sx < 0 ? c[0] + 1 : c[0] - 1) : xa;
        xb = in_block ? xa : xb;
        ok = ok && (!in_block || (sx < 0 ? right > 0 : left > 0));
      }
      xa = xa < 0 ? 0 : xa;
      xb = xb >= A.dims[0] ? A.dims[0] - 1 : xb;
      ok = ok && xa <= xb;
      some = some || ok;
      const int row = (z * A.dims[1] + y) * A.dims[0];
      at_a[r] = ok ? row + xa : 0;  // (entry 0 twice: an empty stretch)
      at_b[r] = ok ? row + xb + 1 : 0;
    }
    if (!__any(some)) return;
    int lo[9], upto[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      lo[r] = A.cell_start[at_a[r]];
      upto[r] = A.cell_start[at_b[r]];
    }
    upto[0] -= lo[0];  // upto[r]: the list position one past row r's last candidate
#pragma unroll
    for (int r = 1; r < 9; ++r) upto[r] = upto[r - 1] + (upto[r] - lo[r]);
    const int listed = upto[8];
    for (int j = 0; __any(j < listed); j += kIcpTrip) {
      int k[kIcpTrip];
      bool present[kIcpTrip];
      S q[kIcpTrip][3];
#pragma unroll
      for (int u = 0; u < kIcpTrip; ++u) {
        const int jj = j + u;
        present[u] = jj < listed;
        int at = lo[8] + (jj - upto[7]);
#pragma unroll
        for (int r = 7; r >= 1; --r) at = jj < upto[r] ? lo[r] + (jj - upto[r - 1]) : at;
        at = jj < upto[0] ? lo[0] + jj : at;
        k[u] = present[u] ? at : 0;
      }
#pragma unroll
      for (int u = 0; u < kIcpTrip; ++u) fetch(k[u], q[u]);
      // (distance and position only: the coordinates of a winner found here are read once, at the end)
#pragma unroll
      for (int u = 0; u < kIcpTrip; ++u) consider(q[u], k[u], present[u], std::false_type());
    }
  };
  if (__any(go)) {
    // the own row and the ring of eight around it
    auto inner = [](int r, int &dy, int &dz) {
      dy = r % 3 - 1;
      dz = r / 3 - 1;
      return true;
    };
    if (R == 1) {
      walkRows(go, inner, std::true_type());
    } else {
      // Cells finer than the radius (icp.cpp picks them so where the targets are dense): after the
      // inner nine rows the rings further out, eight rows at a time, counter-clockwise from each
      // ring's (-ring, -ring) corner.  A lane is done once a ring's nearest face is out of its reach;
      // the walk ends when every lane is.  (Ring and row counters are the same for all lanes.)
      walkRows(go, inner, std::false_type());
      // (further out fewer and fewer lanes admit anything, and what they admit depends on what the
      // rows before have found: past ring A.lock_rings the rows are walked one after the other under
      // the running bound, the form of rounds 3-4 — a wave skips a row none of its lanes admits)
      auto visit = [&](int y, int z, int xa, int xb) {
        xa = xa < 0 ? 0 : xa;
        xb = xb >= A.dims[0] ? A.dims[0] - 1 : xb;
        if (z < 0 || z >= A.dims[2] || y < 0 || y >= A.dims[1] || xa > xb) return;
        const int row = (z * A.dims[1] + y) * A.dims[0];
        const int lo = A.cell_start[row + xa], hi = A.cell_start[row + xb + 1];
        for (int k = lo; k < hi; k += 2) {  // two candidates per step, both loads issued before either is used
          S qa[3], qb[3];
          const bool pair = k + 1 < hi;
          fetch(k, qa);
          fetch(pair ? k + 1 : k, qb);
          consider(qa, k, true, std::false_type());
          consider(qb, k + 1, pair, std::false_type());
        }
      };
      bool walking = go;
      for (int ring = 2; ring <= R; ++ring) {
        const S nearest = fmin(fmin(gapAlong(gy, -ring), gapAlong(gy, ring)),
                               fmin(gapAlong(gz, -ring), gapAlong(gz, ring)));
        walking = walking && within(sq(nearest));
        if (!__any(walking)) break;
        if (ring > A.lock_rings) {
          for (int u = 0; u < 8 * ring; ++u) {
            const int side = u / (2 * ring), along = u % (2 * ring);
            const int dy = side == 0 ? -ring + along : side == 1 ? ring : side == 2 ? ring - along : -ring;
            const int dz = side == 0 ? -ring : side == 1 ? -ring + along : side == 2 ? ring : ring - along;
            const S yz = sq(gapAlong(gy, dy)) + sq(gapAlong(gz, dz));
            if (!(walking && within(yz))) continue;
            int left = 0, right = 0;
            for (int step = 0; step < R; ++step) {
              left += (left == step && within(yz + sq(gx[0] + S(step) * A.cell))) ? 1 : 0;
              right += (right == step && within(yz + sq(gx[1] + S(step) * A.cell))) ? 1 : 0;
            }
            visit(c[1] + dy, c[2] + dz, c[0] - left, c[0] + right);
          }
          continue;
        }
        for (int base = 0; base < 8 * ring; base += 8) {
          auto outer = [&](int r, int &dy, int &dz) {
            const int u = base + r;
            const int side = u / (2 * ring), along = u % (2 * ring);
            dy = side == 0 ? -ring + along : side == 1 ? ring : side == 2 ? ring - along : -ring;
            dz = side == 0 ? -ring : side == 1 ? -ring + along : side == 2 ? ring : ring - along;
            return r < 8;
          };
          walkRows(walking, outer, std::false_type());
        }
      }
    }
  }
  if (go && best_k != first_round_k) fetch(best_k, best);
  const S nan = S(__builtin_nan(""));
  slot[3 * TP] = found ? best[0] : nan;
  slot[4 * TP] = found ? best[1] : nan;
  slot[5 * TP] = found ? best[2] : nan;
  // How many sources found a target: every wave leaves its count in a slot of its own and the
  // publishing kernel adds the slots — no atomics (15.6 k of them on one address, one per wave,
  // take 178 us at 1 M; one per workgroup behind a barrier made the search 57.6 us instead of 44.4).
  if (A.matched) {
    const unsigned long long in_wave = __ballot(found);
    if ((threadIdx.x & 63) == 0)
      A.matched[(size_t)blockIdx.x * (kBlockThreads / 64) + threadIdx.x / 64] =
          (unsigned int)__popcll(in_wave);
  }
}

template <typename S, int TRIP>
__global__ __launch_bounds__(kBlockThreads) void icpMatchKernel(const IcpMatchArgs<S> A) {
  icpMatchBody<S, TRIP>(A, A.T);
}

// Resident form for the device-resident LM: the pose comes from the cost's sweep constants in HBM
// (T at the point the step kernel has just proposed), and the search only runs when the step
// kernel asked for it — the model's update(x) at the top of an outer iteration
// (levenberg_marquadt_dyn.cpp:54) — or not at all once the loop has stopped.
template <typename S, int TRIP>
__global__ __launch_bounds__(kBlockThreads) void icpMatchResidentKernel(
    const IcpMatchArgs<S> A, const P2PSweepArgs<S> *__restrict__ d_args,
    const LmControl *__restrict__ control) {
  if (control->done || !control->pad[1]) return;
  S T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = d_args->T[0][k];
  icpMatchBody<S, TRIP>(A, T);
}

// the target of slot i as a packed triple: to position i, or to position order[i] (the caller's
// index of the source in that slot: an ICP cost keeps its sources in cell order)
template <typename S>
__global__ void gatherTargetsKernel(const S *__restrict__ tiles, long long count,
                                    const int *__restrict__ order, S *__restrict__ out_xyz) {
  constexpr int TP = TileShape<S>::kPoints;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const S *slot = tiles + (i / TP) * TileShape<S>::kP2PScalars + (i % TP);
  const long long to = order ? order[i] : i;
#pragma unroll
  for (int k = 0; k < 3; ++k) out_xyz[3 * to + k] = slot[(3 + k) * TP];
}

// ---- layout conversion (once per data set) -----------------------------------------------------
template <typename S>
__global__ void relayoutP2PKernel(const S *__restrict__ src, const S *__restrict__ tgt,
                                  long long count, S *__restrict__ tiles, long long padded) {
  constexpr int TP = TileShape<S>::kPoints;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  const long long tile = i / TP;
  const int within = int(i % TP);
  S *dst = tiles + tile * TileShape<S>::kP2PScalars + within;
  const bool valid = i < count;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dst[k * TP] = valid ? src[3 * i + k] : S(0);
    dst[(3 + k) * TP] = valid ? tgt[3 * i + k] : S(0);
  }
}

__global__ void relayoutReprojKernel(const double *__restrict__ pts, const int *__restrict__ pix,
                                     long long count, unsigned char *__restrict__ tiles,
                                     long long padded) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  const long long tile = i / kReprojTilePoints;
  const int within = int(i % kReprojTilePoints);
  unsigned char *tb = tiles + tile * kReprojTileBytes;
  double *planes = reinterpret_cast<double *>(tb);
  int *pixels = reinterpret_cast<int *>(tb + size_t(4) * kReprojTilePoints * 8);
  // the tail of the last tile repeats the last element: whatever the sweep computes for a padded slot
  // is then as finite as for a real one, and a weight of zero is enough to leave it out (an all-zero
  // point projects to 0 / 0)
  const long long from = i < count ? i : count - 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) planes[k * kReprojTilePoints + within] = pts[4 * from + k];
  pixels[2 * within + 0] = pix[2 * from + 0];
  pixels[2 * within + 1] = pix[2 * from + 1];
}

// ---- finalisation: workgroup partials -> H | b | sum_sq -----------------------------------------
// One workgroup of 1024 threads.  partials[grid][nacc] is read as a flat array with a stride that
// is a multiple of nacc, so each thread stays in one column and all of its (few) loads are
// independent and in flight together — the kernel costs about one memory round trip, not one
// per row.  Column totals are then formed in a fixed order (bitwise reproducible).
constexpr int kFinalThreads = 1024;
// The resident finalize kernels that run the LM step in the same launch: 512 threads, so that a
// lane has 256 VGPRs — under the 128 of a 1024-thread workgroup the step's register-held solve
// spilled 434 VGPRs to scratch on its one busy lane.  (The width of a one-workgroup kernel costs
// nothing measurable: scripts/probes/finalize_width_probe.cpp.)  At least n^2 + n + 1 = 273
// threads: one per published value of a 16-parameter problem.
constexpr int kStepThreads = 512;
constexpr int kMaxAccumulators = 288;  // n <= 16 (run-time compiled wide models): n*n + n + 1 <= 273

// Write-through store at system scope (sc0 sc1): straight to mapped host memory, nothing left
// dirty in L2, so publishing needs no L2 write-back (a system-scope release fence costs two).
__device__ __forceinline__ void storeSystem(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p),
                     static_cast<unsigned long long>(__double_as_longlong(v)), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

// Called by every thread of the (single) finalize workgroup; threads k < count hold result k in
// `v`.  Payload goes out write-through, every storing wave drains it (s_waitcnt vmcnt(0)), the
// workgroup meets at a barrier, then one lane stores the sequence word — the payload is complete
// before the flag is issued (cdna_hip_programming.md Guideline 16, form R1, at system scope).
__device__ __forceinline__ void publishToHost(const HostPublish &pub, int count, double v,
                                              unsigned long long status = 0) {
  if (pub.host_flag == nullptr) return;
  if (int(threadIdx.x) < count && pub.host_result) storeSystem(pub.host_result + threadIdx.x, v);
  if (threadIdx.x == 0 && pub.host_status)
    __hip_atomic_store(pub.host_status, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(pub.host_flag, pub.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double loadSystem(const double *p) {
  return __longlong_as_double(static_cast<long long>(
      __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_SYSTEM)));
}

// Sum of `v` over the ranks (sweep.hpp, PeerCombine), by the single finalize workgroup; threads
// k < count hold value k.  Same producer form as publishToHost (write-through payload, every
// storing wave drained, barrier, then the sequence word), here into every rank's slot block;
// the consumer side polls the G flags of the own block with system-scope loads from one lane
// each, sleeps between polls, gives up after timeout_ticks (the wait must end on every path: a
// peer that never arrives turns into kStatusPeerTimeout and NaN sums, not into a hung GPU), and
// then adds the G slots in rank order.
// (`sequence_add`: the resident kernels' trial count on top of pc.sequence — a separate argument,
// because a modified copy of `pc` is a private array indexed by lane: scratch for every launch)
__device__ __forceinline__ double peerCombine(const PeerCombine &pc, int count, double v,
                                              unsigned long long *status,
                                              unsigned long long sequence_add = 0) {
  *status = 0;
  if (pc.num_ranks <= 0) return v;
  const unsigned long long sequence = pc.sequence + sequence_add;
  __shared__ double *peer_block[kMaxPeers];
  __shared__ int timed_out;
  const int tid = threadIdx.x;
  const int G = pc.num_ranks;
  if (tid < kMaxPeers) peer_block[tid] = pc.blocks[tid < G ? tid : 0];
  if (tid == 0) timed_out = 0;
  __syncthreads();
  const size_t mine = slotIndex(sequence, G, pc.rank);
  if (tid < count)
    for (int p = 0; p < G; ++p) storeSystem(peer_block[p] + mine + pc.offset + tid, v);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  double *own = peer_block[pc.rank];
  if (tid < G) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(peer_block[tid] + mine + kSlotFlag),
                       sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(
        own + slotIndex(sequence, G, tid) + kSlotFlag);
    const unsigned long long started = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < sequence) {
      if (wall_clock64() - started > pc.timeout_ticks) {
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  double total = 0.0;
  if (tid < count)
    for (int k = 0; k < G; ++k)
      total += loadSystem(own + slotIndex(sequence, G, k) + pc.offset + tid);
  if (timed_out) {
    *status = kStatusPeerTimeout;
    total = __builtin_nan("");
  }
  return total;
}

// K rows of one thread (t, t + stride, ...; zeros past the end) added up.  The K loads are issued
// together from clamped indices and fenced from the selects that zero the rows past the end: a load
// under a branch gets its own wait, and left alone the scheduler interleaves loads and selects
// (reusing vcc) — either way a batch became 3-4 dependent round trips.
template <int K>
__device__ __forceinline__ double columnBatch(const double *partials, int t, int stride,
                                              int total_elems) {
  int at[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int idx = t + k * stride;
    at[k] = idx < total_elems ? idx : 0;  // (row 0 exists whatever the grid)
  }
  __builtin_amdgcn_sched_barrier(0);
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = partials[at[k]];
  __builtin_amdgcn_sched_barrier(0);
  // (the comparisons again from a laundered copy of t: left to itself the compiler keeps the K lane
  // masks of the first round alive across the loads — 2 K scalar registers, spilled from K = 32 on)
  int t_again = t;
  asm volatile("" : "+v"(t_again));
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < K; ++k) s[k & 3] += (t_again + k * stride) < total_elems ? v[k] : 0.0;
  return (s[0] + s[1]) + (s[2] + s[3]);
}

struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// Column sums of the `grid` partial rows of `nacc` values -> total[0 .. nacc).  `parked` runs once
// per thread after the thread's rows have been requested and added and before the first barrier:
// the place where a resident kernel puts values it requested *before* this call (basis, LM state)
// into LDS, so that their round trip overlaps the rows'.
template <int T, typename Parked = NoHook>
__device__ __forceinline__ void columnTotals(const double *partials, int grid, int nacc,
                                             double (&scratch)[T],
                                             double (&total)[kMaxAccumulators],
                                             Parked parked = Parked()) {
  const int per_col = T / nacc;  // threads per column
  const int stride = per_col * nacc;
  const int total_elems = grid * nacc;
  const int t = threadIdx.x;
  // rows per thread, the same for all — 1024 threads: 256 rows x 23 values: 6; the two-workgroups-
  // per-CU sweeps, 512 rows x 28: 15, x 43: 23; 512 threads: 12, 29, 47 — each a single batch, one
  // memory round trip; larger grids continue in batches of four
  const int rows = (total_elems + stride - 1) / stride;
  constexpr bool kRoomy = T <= 512;  // 256 VGPRs a lane: batches of 32 and 48 fit
  double sum = 0.0;
  if (t < stride) {
    if (rows <= 6) {
      sum = columnBatch<6>(partials, t, stride, total_elems);
    } else if (rows <= 12) {
      sum = columnBatch<12>(partials, t, stride, total_elems);
    } else if (rows <= 16) {
      sum = columnBatch<16>(partials, t, stride, total_elems);
    } else if (rows <= 24) {
      sum = columnBatch<24>(partials, t, stride, total_elems);
    } else if (kRoomy && rows <= 32) {
      sum = columnBatch<kRoomy ? 32 : 4>(partials, t, stride, total_elems);
    } else if (kRoomy && rows <= 48) {
      sum = columnBatch<kRoomy ? 48 : 4>(partials, t, stride, total_elems);
    } else {
      sum = columnBatch<24>(partials, t, stride, total_elems);
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int idx = t + 24 * stride;
      for (; idx + 3 * stride < total_elems; idx += 4 * stride) {
        const double a = partials[idx], b = partials[idx + stride], c = partials[idx + 2 * stride],
                     d = partials[idx + 3 * stride];
        s0 += a;
        s1 += b;
        s2 += c;
        s3 += d;
      }
      for (; idx < total_elems; idx += stride) s0 += partials[idx];
      sum += (s0 + s1) + (s2 + s3);
    }
  }
  parked();
  scratch[t] = sum;
  __syncthreads();
  if (t < nacc) {
    double v = 0.0;
    for (int g = 0; g < per_col; ++g) v += scratch[t + g * nacc];
    total[t] = v;
  }
  __syncthreads();
}

// rows of [upper triangle or full H | b | sum_sq] over n parameters -> H (n x n column-major) | b |
// sum_sq.  nacc tells the form: n(n+1)/2 + n + 1 (symmetric) or n*n + n + 1 (full).
template <int T, typename Parked = NoHook>
__device__ __forceinline__ unsigned long long finalizeDenseBody(const double *partials, int grid,
                                                                int nacc, int n, double *result,
                                                                const HostPublish &pub,
                                                                const PeerCombine &pc,
                                                                double *result_lds = nullptr,
                                                                Parked parked = Parked(),
                                                                unsigned long long sequence_add = 0) {
  static_assert(T >= kMaxWideParams * kMaxWideParams + kMaxWideParams + 1, "one thread per value");
  __shared__ double scratch[T];
  __shared__ double total[kMaxAccumulators];
  columnTotals<T>(partials, grid, nacc, scratch, total, parked);
  const int k = threadIdx.x;
  const int count = n * n + n + 1;
  double v = 0.0;
  if (k < count) {
    const bool full = (nacc == count);
    const int nh = full ? n * n : n * (n + 1) / 2;
    if (k < n * n) {
      const int i = k % n, j = k / n;  // column-major H(i, j)
      if (full) {
        v = total[j * n + i];
      } else {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        v = total[hi * (hi + 1) / 2 + lo];
      }
    } else {
      v = total[nh + (k - n * n)];  // b (n) then sum_sq
    }
  }
  unsigned long long status;
  v = peerCombine(pc, count, v, &status, sequence_add);
  if (k < count) {
    result[k] = v;
    if (result_lds) result_lds[k] = v;
  }
  publishToHost(pub, count, v, status);
  return status;
}

__global__ __launch_bounds__(kFinalThreads) void finalizeDenseKernel(const double *partials,
                                                                      int grid, int nacc, int n,
                                                                      double *result,
                                                                      const HostPublish pub,
                                                                      const PeerCombine pc) {
  finalizeDenseBody<kFinalThreads>(partials, grid, nacc, n, result, pub, pc);
}

// Resident forms (device-resident LM): nothing goes to the host, the peer-combine sequence number
// is the base the host assigned plus the number of trials the step kernel has counted.  STEP = 4 or
// 8: this cost is the last of its problem and the LM step (lm_device.hpp) runs right here, on the
// result still in LDS, as LevenbergMarquadtDynamic<float> or <double> — one launch and one memory
// round trip less per evaluated point; the stored LM state is requested before the finalize work so
// that its latency hides behind it.
template <int STEP>
struct StepScalar {
  using type = double;
};
template <>
struct StepScalar<4> {
  using type = float;
};

template <int STEP>
__global__ __launch_bounds__(kStepThreads) void finalizeDenseResidentKernel(
    const double *partials, int grid, int nacc, int n, double *result, LmControl *control,
    PeerCombine pc, const LmProblem P, int own_index) {
  if (control->done) return;
  using S = typename StepScalar<STEP>::type;
  LmStateWords state_words = {0u, 0u};
  if constexpr (STEP != 0) state_words = lmPrefetchState<S>(P);
  __shared__ double own[kSlotData];
  const unsigned long long status = finalizeDenseBody<kStepThreads>(
      partials, grid, nacc, n, result, HostPublish(), pc, own, NoHook(),
      (unsigned long long)control->trial);
  if (status && threadIdx.x == 0) control->pad[0] = int(status);  // a rank went missing
  if constexpr (STEP != 0) {
    __syncthreads();
    lmStepBody<S>(P, false, LmStart<S>(), own, own_index, true, state_words);
  }
}

// PEERS = false: no exchange between ranks, and no PeerCombine to read (a default-constructed one
// handed in by reference is a private array, i.e. scratch memory).
template <int T, typename Parked = NoHook, bool PEERS = true>
__device__ __forceinline__ unsigned long long finalizeMomentsBody(const double *partials, int grid,
                                                                  const AffineBasis &B,
                                                                  double *result,
                                                                  const HostPublish &pub,
                                                                  const PeerCombine &pc,
                                                                  double *result_lds = nullptr,
                                                                  Parked parked = Parked(),
                                                                  unsigned long long sequence_add = 0) {
  __shared__ double scratch[T];
  __shared__ double total[kMaxAccumulators];
  __shared__ double terms[36 * 16 + 6 * 4];
  columnTotals<T>(partials, grid, kAccMoments, scratch, total, parked);
  const int t = threadIdx.x;
  // 36 x 16 products for H, 6 x 4 for b: one per thread, in trips of T
  for (int item = t; item < 36 * 16 + 24; item += T) {
  if (item < 36 * 16) {
    const int o = item >> 4, ab = item & 15;
    const int i = o % 6, j = o / 6;
    int a = ab >> 2, b = ab & 3;
    double form = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double sj = 0.0;  // (S J_b)(r, j)
#pragma unroll
      for (int c = 0; c < 3; ++c) sj += B.cov[r * 3 + c] * B.J[b][c * 6 + j];
      form += B.J[a][r * 6 + i] * sj;
    }
    if (a > b) {
      const int tmp = a;
      a = b;
      b = tmp;
    }
    // (0,0)=0 (0,b)=b (1,1)=4 (1,2)=5 (1,3)=6 (2,2)=7 (2,3)=8 (3,3)=9
    const int wi = a == 0 ? b : (a == 1 ? 3 + b : (a == 2 ? 5 + b : 9));
    terms[item] = total[wi] * form;
  } else {
    const int u = item - 36 * 16;
    const int i = u >> 2, a = u & 3;
    double g = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double sv = 0.0;  // (S V(a, .))(r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        sv += B.cov[r * 3 + c] * (a == 0 ? total[10 + c] : total[13 + 3 * (a - 1) + c]);
      g += B.J[a][r * 6 + i] * sv;
    }
    terms[item] = g;
  }
  }
  __syncthreads();
  double v = 0.0;
  if (t < kResultDoubles) {
    if (t < 36) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v += terms[t * 16 + q];
    } else if (t < 42) {
      const int i = t - 36;
      v = ((terms[576 + i * 4] + terms[576 + i * 4 + 1]) + terms[576 + i * 4 + 2]) +
          terms[576 + i * 4 + 3];
    } else {
      v = total[22];
    }
  }
  unsigned long long status = 0;
  if constexpr (PEERS) v = peerCombine(pc, kResultDoubles, v, &status, sequence_add);
  if (t < kResultDoubles) {
    result[t] = v;
    if (result_lds) result_lds[t] = v;
  }
  publishToHost(pub, kResultDoubles, v, status);
  return status;
}

__global__ __launch_bounds__(kFinalThreads) void finalizeMomentsKernel(const double *partials,
                                                                        int grid,
                                                                        const AffineBasis B,
                                                                        double *result,
                                                                        const HostPublish pub,
                                                                        const PeerCombine pc) {
  finalizeMomentsBody<kFinalThreads>(partials, grid, B, result, pub, pc);
}

template <int STEP>
__global__ __launch_bounds__(kStepThreads) void finalizeMomentsResidentKernel(
    const double *partials, int grid, const AffineBasis *__restrict__ d_basis, double *result,
    LmControl *control, const PeerCombine pc, const LmProblem P, int own_index) {
#ifdef MOPT_LM_TIMING
  const unsigned long long tick_entry = wall_clock64();
#endif
  if (control->done) return;
#ifdef MOPT_LM_TIMING
  const unsigned long long tick_control = wall_clock64();
#endif
  using S = typename StepScalar<STEP>::type;
  LmStateWords state_words = {0u, 0u};
  if constexpr (STEP != 0) state_words = lmPrefetchState<S>(P);
  // the basis is requested now and parked in LDS once the partial rows have been requested too
  // (columnTotals' hook): one round trip for state, basis and rows together
  constexpr int kBasisDoubles = int(sizeof(AffineBasis) / sizeof(double));
  const double basis_value = reinterpret_cast<const double *>(
      d_basis)[int(threadIdx.x) < kBasisDoubles ? int(threadIdx.x) : 0];  // unconditional: no wait
  __shared__ AffineBasis B;
  __shared__ double own[kSlotData];
  const auto park_basis = [&]() {
    if (int(threadIdx.x) < kBasisDoubles) reinterpret_cast<double *>(&B)[threadIdx.x] = basis_value;
  };
#ifdef MOPT_LM_TIMING
  const unsigned long long tick_basis = wall_clock64();
#endif
  const unsigned long long status = finalizeMomentsBody<kStepThreads>(
      partials, grid, B, result, HostPublish(), pc, own, park_basis,
      (unsigned long long)control->trial);
  if (status && threadIdx.x == 0) control->pad[0] = int(status);
#ifdef MOPT_LM_TIMING
  if (threadIdx.x == 0)
    printf("finalize: control word %llu, basis + state in LDS %llu, sums contracted %llu (x10 ns)\n",
           tick_control - tick_entry, tick_basis - tick_control, wall_clock64() - tick_basis);
#endif
  if constexpr (STEP != 0) {
    __syncthreads();
    // (rows of moments are a point2point cost's: the problem has its six parameters)
    lmStepBody<S, kNumParams>(P, false, LmStart<S>(), own, own_index, true, state_words);
  }
}

// A cost whose forward-difference sweep is chosen per evaluated point (sweep.hpp kLmGateMoments): the
// sweep left rows of moments (the word zero) or dense rows (non-zero); the word is rewritten only by the
// step at the end of this kernel, so it still says which.
template <int STEP>
__global__ __launch_bounds__(kStepThreads) void finalizeEitherResidentKernel(
    const double *partials, int grid_moments, int grid_literal, int nacc,
    const AffineBasis *__restrict__ d_basis, double *result, LmControl *control, const PeerCombine pc,
    const LmProblem P, int own_index) {
  const int gate = control[kLmGateMoments].done;  // (uniform: one scalar load)
  if (gate == kLmGateStopped) return;
  const bool literal = gate == kLmGateLiteralForm;
  using S = typename StepScalar<STEP>::type;
  LmStateWords state_words = {0u, 0u};
  if constexpr (STEP != 0) state_words = lmPrefetchState<S>(P);
  // the basis is requested whichever form ran (the literal one does not use it): with the state it is in
  // flight before the branch, and the rows are requested right behind — one round trip, as in the
  // single-purpose finalize kernels
  constexpr int kBasisDoubles = int(sizeof(AffineBasis) / sizeof(double));
  const double basis_value = reinterpret_cast<const double *>(
      d_basis)[int(threadIdx.x) < kBasisDoubles ? int(threadIdx.x) : 0];
  const unsigned long long trial = (unsigned long long)control->trial;
  __shared__ AffineBasis B;
  __shared__ double own[kSlotData];
  unsigned long long status;
  if (literal) {
    status = finalizeDenseBody<kStepThreads>(partials, grid_literal, nacc, kNumParams, result, HostPublish(),
                                             pc, own, NoHook(), trial);
  } else {
    const auto park_basis = [&]() {
      if (int(threadIdx.x) < kBasisDoubles) reinterpret_cast<double *>(&B)[threadIdx.x] = basis_value;
    };
    status = finalizeMomentsBody<kStepThreads>(partials, grid_moments, B, result, HostPublish(), pc, own,
                                               park_basis, trial);
  }
  if (status && threadIdx.x == 0) control->pad[0] = int(status);
  if constexpr (STEP != 0) {
    __syncthreads();
    lmStepBody<S, kNumParams>(P, false, LmStart<S>(), own, own_index, true, state_words);
  }
}

// ---- a whole minimisation in one launch ---------------------------------------------------------
// Small point2point problems (the reference's own test sizes: tst/point2point.cpp registers 1 k
// correspondences): under the launch-per-point loop an evaluated point costs two launches, 11-12 us,
// of which the sweep of a tile or two is a fraction.  Here one workgroup runs the loop of
// levenberg_marquadt_dyn.cpp:34-119 without leaving the kernel.  The correspondences — at most
// kSolveSmallTiles tiles — are loaded once and stay in registers for every evaluated point; per
// point the workgroup adds up their moments (momentsOfPack, the arithmetic of the resident sweep),
// reduces them to one row in LDS, contracts it as finalizeMomentsResidentKernel does and takes the
// LM step, whose state stays in LDS; the per-x constants and the Jacobian basis the step leaves for
// the next sweep stay in LDS too (through HBM they would come back through the scalar cache, which
// does not see this kernel's own stores).  The sums are those of the launch-per-point loop added in
// another order (one row instead of one per tile): the same iterates to rounding
// (tests/test_gpu_device_lm.py).  At most `max_points` evaluated points: the loop's own bound, which
// every thread reaches.
constexpr int kSolveSmallTiles = 4;

// FD_COV >= 0 (a covariance form): the kernel also holds the literal forward-difference form for that
// covariance (fd_device.hpp, over the packs held in registers, rotations re-read from LDS: registers are
// what this kernel is short of) and the step says per point which form runs — a cost that differentiates
// numerically under MOPT_KERNEL_AUTO / _MOMENTS (LmProblem::fd_per_iterate).  FD_COV = -1: moments only.
template <typename S, int FD_COV>
__global__ __launch_bounds__(kBlockThreads) void p2pSolveSmallKernel(
    const S *tiles, int num_tiles, const P2PSweepArgs<S> *__restrict__ d_args,
    const AffineBasis *__restrict__ d_basis, double *result, const LmProblem problem,
    const LmStart<S> start, int max_points) {
  constexpr bool kChooses = FD_COV >= 0;
  __shared__ int fd_choice;  // written by the step: the next point takes the literal form
  if (threadIdx.x == 0) fd_choice = 0;
  __shared__ P2PSweepArgs<S> A;
  __shared__ AffineBasis B;
  // the problem's description, read from LDS inside the loop: as kernel arguments its 1.2 KB are
  // loop invariants the compiler loads up front into scalar registers, of which there are 104 —
  // some 400 spills to vector lanes, read back one by one in every step
  static_assert(sizeof(LmProblem) % 4 == 0, "copied by words");
  __shared__ alignas(16) unsigned int problem_words[sizeof(LmProblem) / 4];
  for (int i = threadIdx.x; i < int(sizeof(LmProblem) / 4); i += kBlockThreads)
    problem_words[i] = reinterpret_cast<const unsigned int *>(&problem)[i];
  const LmProblem &P = *reinterpret_cast<const LmProblem *>(problem_words);
  __shared__ double row[kAccMoments];
  __shared__ double own[kSlotData];
  constexpr int V = TileShape<S>::kVec, TP = TileShape<S>::kPoints;
  Pack<S> held[kSolveSmallTiles][6];
#pragma unroll
  for (int t = 0; t < kSolveSmallTiles; ++t) {
    // (past the last tile: that tile again — an unconditional load; its values are never used)
    const S *base = tiles + size_t(t < num_tiles ? t : num_tiles - 1) * TileShape<S>::kP2PScalars +
                    threadIdx.x * V;
#pragma unroll
    for (int pl = 0; pl < 6; ++pl) held[t][pl] = loadPack<S>(base + pl * TP);
  }
  static_assert(sizeof(P2PSweepArgs<S>) % 4 == 0 && sizeof(AffineBasis) % 4 == 0, "copied by words");
  constexpr int kArgWords = int(sizeof(P2PSweepArgs<S>) / 4), kBasisWords = int(sizeof(AffineBasis) / 4);
  for (int i = threadIdx.x; i < kArgWords; i += kBlockThreads)
    reinterpret_cast<unsigned int *>(&A)[i] = reinterpret_cast<const unsigned int *>(d_args)[i];
  for (int i = threadIdx.x; i < kBasisWords; i += kBlockThreads)
    reinterpret_cast<unsigned int *>(&B)[i] = reinterpret_cast<const unsigned int *>(d_basis)[i];
  __syncthreads();
  // (one inlined copy of the step: its first run starts the minimisation, like lmStepKernel's `init`)
  for (int point = 0;; ++point) {
#ifdef MOPT_LM_TIMING
    const unsigned long long tick_step = wall_clock64();
#endif
    const bool finished = lmStepBodyFor<S, kMaxParams, true, kNumParams, kChooses>(
        P, point == 0, start, own, 0, false, LmStateWords(), &A, &B, &fd_choice);
    if (finished || point >= max_points) break;
#ifdef MOPT_LM_TIMING
    const unsigned long long tick_sweep = wall_clock64();
    unsigned long long tick_finalize = tick_sweep;
#endif
    bool literal = false;
    if constexpr (kChooses) literal = fd_choice != 0;  // (uniform; the step's barriers precede this read)
    if (literal) {
      if constexpr (kChooses) {
        constexpr int NACC = (FD_COV == kCovGeneral) ? kAccFull : kAccSym;
        __shared__ double dense_row[NACC];
        p2pForwardDiffRow<S, FD_COV, kFdRotationLds>(
            A, A,
            [&](auto &&body) {
#pragma unroll
              for (int t = 0; t < kSolveSmallTiles; ++t)
                if (t < num_tiles) body(held[t], (long long)t * TP + threadIdx.x * V);
            },
            dense_row);
        __syncthreads();
        // one row: nothing to add up — H (column-major) | b | sum_sq out of the row's layout, as
        // finalizeDenseBody lays them out
        const int k = threadIdx.x;
        constexpr int n = kNumParams, count = n * n + n + 1;
        if (k < count) {
          double v;
          if (k < n * n) {
            const int i = k % n, j = k / n;
            if (FD_COV == kCovGeneral) {
              v = dense_row[j * n + i];
            } else {
              const int lo = i < j ? i : j, hi = i < j ? j : i;
              v = dense_row[hi * (hi + 1) / 2 + lo];
            }
          } else {
            v = dense_row[(FD_COV == kCovGeneral ? n * n : n * (n + 1) / 2) + (k - n * n)];
          }
          result[k] = v;
          own[k] = v;
        }
        __syncthreads();
      }
    } else {
      double acc[kAccMoments];
#pragma unroll
      for (int k = 0; k < kAccMoments; ++k) acc[k] = 0.0;
#pragma unroll
      for (int t = 0; t < kSolveSmallTiles; ++t)
        if (t < num_tiles) momentsOfPack<S>(acc, held[t], (long long)t * TP + threadIdx.x * V, A);
      blockReduceStore<kAccMoments>(acc, row);
      __syncthreads();
#ifdef MOPT_LM_TIMING
      tick_finalize = wall_clock64();
#endif
      finalizeMomentsBody<kBlockThreads, NoHook, false>(row, 1, B, result, HostPublish(), PeerCombine(),
                                                        own);
      __syncthreads();
    }
#ifdef MOPT_LM_TIMING
    if (threadIdx.x == 0)
      printf("one launch, point %d: step %llu sweep of %d tiles %llu contraction %llu (x10 ns)\n", point,
             tick_sweep - tick_step, num_tiles, tick_finalize - tick_sweep, wall_clock64() - tick_finalize);
#endif
  }
}

__device__ __forceinline__ void finalizeCostBody(const double *partials, int grid, double *result,
                                                 const HostPublish &pub, const PeerCombine &pc) {
  __shared__ double lds[kFinalThreads / 64];
  double s = 0.0;
  for (int row = threadIdx.x; row < grid; row += kFinalThreads) s += partials[row];
  s = waveSum(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < kFinalThreads / 64; ++k) v += lds[k];
    lds[0] = v;
  }
  __syncthreads();
  unsigned long long status;
  const double v = peerCombine(pc, 1, lds[0], &status);
  if (threadIdx.x == 0) result[0] = v;
  publishToHost(pub, 1, v, status);
}

__global__ __launch_bounds__(kFinalThreads) void finalizeCostKernel(const double *partials,
                                                                     int grid, double *result,
                                                                     const HostPublish pub,
                                                                     const PeerCombine pc) {
  finalizeCostBody(partials, grid, result, pub, pc);
}

__global__ void publishKernel(const double *values, int count, const HostPublish pub) {
  const double v = int(threadIdx.x) < count ? values[threadIdx.x] : 0.0;
  publishToHost(pub, count, v);
}

}  // namespace

// ---- host-side launch wrappers -----------------------------------------------------------------
template <typename S>
hipError_t launchRelayoutP2P(const S *src_xyz, const S *tgt_xyz, long long count, S *tiles,
                             int num_tiles, hipStream_t stream) {
  const long long padded = (long long)num_tiles * TileShape<S>::kPoints;
  if (padded == 0) return hipSuccess;
  const unsigned blocks = unsigned((padded + 255) / 256);
  hipLaunchKernelGGL((relayoutP2PKernel<S>), dim3(blocks), dim3(256), 0, stream, src_xyz, tgt_xyz,
                     count, tiles, padded);
  return hipGetLastError();
}
template hipError_t launchRelayoutP2P<float>(const float *, const float *, long long, float *, int,
                                             hipStream_t);
template hipError_t launchRelayoutP2P<double>(const double *, const double *, long long, double *,
                                              int, hipStream_t);

hipError_t launchRelayoutReproj(const double *points_xyzw, const int32_t *pixels_uv,
                                long long count, unsigned char *tiles, int num_tiles,
                                hipStream_t stream) {
  const long long padded = (long long)num_tiles * kReprojTilePoints;
  if (padded == 0) return hipSuccess;
  const unsigned blocks = unsigned((padded + 255) / 256);
  hipLaunchKernelGGL(relayoutReprojKernel, dim3(blocks), dim3(256), 0, stream, points_xyzw,
                     pixels_uv, count, tiles, padded);
  return hipGetLastError();
}

namespace {
template <typename Kernel, typename Args>
hipError_t launchSweep(Kernel kernel, int grid, const LaunchSite &site, const Args &args) {
  if (site.aql.queue && site.aql_used && !site.time_start &&
      mopt_detail::aqlLaunch(site.aql, kernel, uint32_t(grid), uint32_t(kBlockThreads), args)) {
    *site.aql_used = true;
    return hipSuccess;
  }
  if (site.time_start && site.time_stop)
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kBlockThreads), 0, site.stream, site.time_start,
                          site.time_stop, 0, args);
  else
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlockThreads), 0, site.stream, args);
  return hipGetLastError();
}

template <typename S, int JAC>
hipError_t launchLiteralCov(const P2PSweepArgs<S> &args, int cov_mode, int grid,
                            const LaunchSite &site) {
#define MOPT_LAUNCH_LITERAL(COV)                                                                   \
  (site.streaming ? launchTiled(p2pLinearizeLiteralKernel<S, JAC, COV, true>, grid, site, args)   \
                  : launchTiled(p2pLinearizeLiteralKernel<S, JAC, COV, false>, grid, site, args))
  switch (cov_mode) {
    case kCovIdentity:
      return MOPT_LAUNCH_LITERAL(kCovIdentity);
    case kCovSymmetric:
      return MOPT_LAUNCH_LITERAL(kCovSymmetric);
    default:
      return MOPT_LAUNCH_LITERAL(kCovGeneral);
  }
#undef MOPT_LAUNCH_LITERAL
}

}  // namespace

template <typename S>
hipError_t launchP2PLinearizeLiteral(const P2PSweepArgs<S> &args, int jac_mode, int cov_mode,
                                     int grid, const LaunchSite &site) {
  switch (jac_mode) {
    case kJacAnalytic:
      return launchLiteralCov<S, kJacAnalytic>(args, cov_mode, grid, site);
    case kJacAnalyticTst:
      return launchLiteralCov<S, kJacAnalyticTst>(args, cov_mode, grid, site);
    case kJacNumeric:
      return launchForwardDiff<S>(args, cov_mode, grid, site);
    case kJacAnalyticLeft:
      return launchLiteralCov<S, kJacAnalyticLeft>(args, cov_mode, grid, site);
    case kJacAnalyticRight:
      return launchLiteralCov<S, kJacAnalyticRight>(args, cov_mode, grid, site);
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launchP2PLinearizeLiteral<float>(const P2PSweepArgs<float> &, int, int, int,
                                                     const LaunchSite &);
template hipError_t launchP2PLinearizeLiteral<double>(const P2PSweepArgs<double> &, int, int, int,
                                                      const LaunchSite &);

template <typename S>
hipError_t launchP2PMoments(const P2PSweepArgs<S> &args, int grid, const LaunchSite &site) {
  return site.streaming ? launchTiled(p2pMomentsKernel<S, true>, grid, site, args)
                        : launchTiled(p2pMomentsKernel<S, false>, grid, site, args);
}
template hipError_t launchP2PMoments<float>(const P2PSweepArgs<float> &, int, const LaunchSite &);
template hipError_t launchP2PMoments<double>(const P2PSweepArgs<double> &, int,
                                             const LaunchSite &);

template <typename S>
hipError_t launchP2PCost(const P2PSweepArgs<S> &args, int grid, const LaunchSite &site) {
  return site.streaming ? launchTiled(p2pCostKernel<S, true>, grid, site, args)
                        : launchTiled(p2pCostKernel<S, false>, grid, site, args);
}
template hipError_t launchP2PCost<float>(const P2PSweepArgs<float> &, int, const LaunchSite &);
template hipError_t launchP2PCost<double>(const P2PSweepArgs<double> &, int, const LaunchSite &);

hipError_t launchReprojLinearize(const ReprojSweepArgs &args, int cov_mode, int grid,
                                 const LaunchSite &site) {
  switch (cov_mode) {
    case kCovIdentity:
      return launchSweep(reprojKernel<kCovIdentity, false>, grid, site, args);
    case kCovSymmetric:
      return launchSweep(reprojKernel<kCovSymmetric, false>, grid, site, args);
    default:
      return launchSweep(reprojKernel<kCovGeneral, false>, grid, site, args);
  }
}

hipError_t launchReprojCost(const ReprojSweepArgs &args, int grid, const LaunchSite &site) {
  return launchSweep(reprojKernel<kCovIdentity, true>, grid, site, args);
}

bool aqlFinalizersLoaded(const mopt_detail::AqlSite &site) {
  return site.queue &&
         mopt_detail::aqlLookup(site.device, reinterpret_cast<const void *>(finalizeDenseKernel)) &&
         mopt_detail::aqlLookup(site.device, reinterpret_cast<const void *>(finalizeMomentsKernel)) &&
         mopt_detail::aqlLookup(site.device, reinterpret_cast<const void *>(finalizeCostKernel));
}

hipError_t launchFinalizeDense(const double *partials, int grid, int nacc, int n, double *result,
                               const HostPublish &pub, hipStream_t stream,
                               const PeerCombine *peers, const mopt_detail::AqlSite *aql) {
  if (n < 1 || n > kMaxWideParams || (nacc != n * (n + 1) / 2 + n + 1 && nacc != n * n + n + 1))
    return hipErrorInvalidValue;
  if (peers && peers->num_ranks > 0 && n * n + n + 1 > kSlotData) return hipErrorInvalidValue;
  if (aql)  // the sweep went to this queue: so must its finalize (looked up beforehand)
    return mopt_detail::aqlLaunch(*aql, finalizeDenseKernel, 1u, uint32_t(kFinalThreads), partials, grid,
                                  nacc, n, result, pub, peers ? *peers : PeerCombine())
               ? hipSuccess
               : hipErrorLaunchFailure;
  hipLaunchKernelGGL(finalizeDenseKernel, dim3(1), dim3(kFinalThreads), 0, stream, partials, grid,
                     nacc, n, result, pub, peers ? *peers : PeerCombine());
  return hipGetLastError();
}

hipError_t launchFinalizeMoments(const double *partials, int grid, const AffineBasis &basis,
                                 double *result, const HostPublish &pub, hipStream_t stream,
                                 const PeerCombine *peers, const mopt_detail::AqlSite *aql) {
  if (aql)
    return mopt_detail::aqlLaunch(*aql, finalizeMomentsKernel, 1u, uint32_t(kFinalThreads), partials, grid,
                                  basis, result, pub, peers ? *peers : PeerCombine())
               ? hipSuccess
               : hipErrorLaunchFailure;
  hipLaunchKernelGGL(finalizeMomentsKernel, dim3(1), dim3(kFinalThreads), 0, stream, partials,
                     grid, basis, result, pub, peers ? *peers : PeerCombine());
  return hipGetLastError();
}

hipError_t launchFinalizeCost(const double *partials, int grid, double *result,
                              const HostPublish &pub, hipStream_t stream,
                              const PeerCombine *peers, const mopt_detail::AqlSite *aql) {
  if (aql)
    return mopt_detail::aqlLaunch(*aql, finalizeCostKernel, 1u, uint32_t(kFinalThreads), partials, grid,
                                  result, pub, peers ? *peers : PeerCombine())
               ? hipSuccess
               : hipErrorLaunchFailure;
  hipLaunchKernelGGL(finalizeCostKernel, dim3(1), dim3(kFinalThreads), 0, stream, partials, grid,
                     result, pub, peers ? *peers : PeerCombine());
  return hipGetLastError();
}

namespace {
template <typename S, template <typename> class ModelT>
hipError_t launchScalarFor(const ScalarSweepArgs<S> &args, bool cost_only, int jac_mode,
                           int cov_mode, int grid, const LaunchSite &site) {
  if (cost_only)
    return launchSweep(scalarModelKernel<S, ModelT, kJacNumeric, kCovIdentity, true>, grid, site, args);
  const bool numeric = (jac_mode == kJacNumeric);
#define MOPT_LAUNCH_SCALAR(JAC, COV) \
  return launchSweep(scalarModelKernel<S, ModelT, JAC, COV, false>, grid, site, args)
  switch (cov_mode) {
    case kCovIdentity:
      if (numeric) MOPT_LAUNCH_SCALAR(kJacNumeric, kCovIdentity);
      else MOPT_LAUNCH_SCALAR(kJacAnalytic, kCovIdentity);
    case kCovSymmetric:
      if (numeric) MOPT_LAUNCH_SCALAR(kJacNumeric, kCovSymmetric);
      else MOPT_LAUNCH_SCALAR(kJacAnalytic, kCovSymmetric);
    default:
      if (numeric) MOPT_LAUNCH_SCALAR(kJacNumeric, kCovGeneral);
      else MOPT_LAUNCH_SCALAR(kJacAnalytic, kCovGeneral);
  }
#undef MOPT_LAUNCH_SCALAR
}
}  // namespace

template <typename S>
hipError_t launchScalarModel(const ScalarSweepArgs<S> &args, int model, bool cost_only,
                             int jac_mode, int cov_mode, int grid, const LaunchSite &site) {
  switch (model) {
    case kScalarExpCurve:
      return launchScalarFor<S, ExpCurve>(args, cost_only, jac_mode, cov_mode, grid, site);
    case kScalarRational:
      return launchScalarFor<S, Rational>(args, cost_only, jac_mode, cov_mode, grid, site);
    case kScalarPowell:
      return launchScalarFor<S, Powell>(args, cost_only, jac_mode, cov_mode, grid, site);
    case kScalarExpCurveMarked:
      return launchScalarFor<S, ExpCurveMarked>(args, cost_only, jac_mode, cov_mode, grid, site);
    case kScalarRationalMarked:
      return launchScalarFor<S, RationalMarked>(args, cost_only, jac_mode, cov_mode, grid, site);
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launchScalarModel<float>(const ScalarSweepArgs<float> &, int, bool, int, int,
                                             int, const LaunchSite &);
template hipError_t launchScalarModel<double>(const ScalarSweepArgs<double> &, int, bool, int, int,
                                              int, const LaunchSite &);

// MOPT_ICP_FIRST_ROUND=0: the row-by-row search alone (measurements: scripts/icp_offsets_timing.py)
static bool icpFirstRound() {
  static const bool on = [] {
    const char *v = getenv("MOPT_ICP_FIRST_ROUND");
    return !(v && v[0] == '0');
  }();
  return on;
}

template <typename S>
hipError_t launchIcpMatch(const IcpMatchArgs<S> &args, hipStream_t stream) {
  const long long padded = (long long)args.num_tiles * TileShape<S>::kPoints;
  if (padded == 0) return hipSuccess;
  const unsigned blocks = unsigned((padded + kBlockThreads - 1) / kBlockThreads);
  if (icpFirstRound())
    hipLaunchKernelGGL((icpMatchKernel<S, kIcpTrip>), dim3(blocks), dim3(kBlockThreads), 0, stream, args);
  else
    hipLaunchKernelGGL((icpMatchKernel<S, 0>), dim3(blocks), dim3(kBlockThreads), 0, stream, args);
  return hipGetLastError();
}
template hipError_t launchIcpMatch<float>(const IcpMatchArgs<float> &, hipStream_t);
template hipError_t launchIcpMatch<double>(const IcpMatchArgs<double> &, hipStream_t);

template <typename S>
hipError_t launchIcpMatchResident(const IcpMatchArgs<S> &args, const P2PSweepArgs<S> *d_args,
                                  const LmControl *control, hipStream_t stream) {
  const long long padded = (long long)args.num_tiles * TileShape<S>::kPoints;
  if (padded == 0) return hipSuccess;
  const unsigned blocks = unsigned((padded + kBlockThreads - 1) / kBlockThreads);
  hipLaunchKernelGGL((icpMatchResidentKernel<S, kIcpTrip>), dim3(blocks), dim3(kBlockThreads), 0, stream, args,
                     d_args, control);
  return hipGetLastError();
}
template hipError_t launchIcpMatchResident<float>(const IcpMatchArgs<float> &,
                                                  const P2PSweepArgs<float> *, const LmControl *,
                                                  hipStream_t);
template hipError_t launchIcpMatchResident<double>(const IcpMatchArgs<double> &,
                                                   const P2PSweepArgs<double> *, const LmControl *,
                                                   hipStream_t);

template <typename S>
hipError_t launchGatherTargets(const S *tiles, long long count, const int *order, S *out_xyz,
                               hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const unsigned blocks = unsigned((count + 255) / 256);
  hipLaunchKernelGGL((gatherTargetsKernel<S>), dim3(blocks), dim3(256), 0, stream, tiles, count,
                     order, out_xyz);
  return hipGetLastError();
}
template hipError_t launchGatherTargets<float>(const float *, long long, const int *, float *,
                                               hipStream_t);
template hipError_t launchGatherTargets<double>(const double *, long long, const int *, double *,
                                                hipStream_t);

// matched sources of the correspondence search: the per-wave counts (icpMatchBody) added up and
// handed to mapped host memory as one double — no memset launch, no copy, no stream synchronisation
__global__ __launch_bounds__(1024) void publishCounterKernel(const unsigned int *wave_counts,
                                                             long long num_waves,
                                                             const HostPublish pub) {
  __shared__ unsigned long long per_wave[16];
  // (16-byte loads, all of a thread's requested before the first is added: 15.6 k counts of a
  // 1 M search are four loads per thread, one round trip)
  unsigned long long sum = 0;
  const long long quads = num_waves / 4;
  const uint4 *counts4 = reinterpret_cast<const uint4 *>(wave_counts);
  for (long long k = threadIdx.x; k < quads; k += 4 * blockDim.x) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long at = k + u * (long long)blockDim.x;
      v[u] = counts4[at < quads ? at : k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k + u * (long long)blockDim.x < quads) sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  if (threadIdx.x < num_waves - quads * 4) sum += wave_counts[quads * 4 + threadIdx.x];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
  if ((threadIdx.x & 63) == 0) per_wave[threadIdx.x / 64] = sum;
  __syncthreads();
  double v = 0.0;
  if (threadIdx.x == 0) {
    unsigned long long total = 0;
    for (int k = 0; k < int(blockDim.x) / 64; ++k) total += per_wave[k];
    v = double(total);
  }
  publishToHost(pub, 1, v);
}

hipError_t launchPublishCounter(const unsigned int *d_wave_counts, long long num_waves,
                                const HostPublish &pub, hipStream_t stream) {
  hipLaunchKernelGGL(publishCounterKernel, dim3(1), dim3(1024), 0, stream, d_wave_counts, num_waves,
                     pub);
  return hipGetLastError();
}

hipError_t launchPublish(const double *d_values, int count, const HostPublish &pub,
                         hipStream_t stream) {
  if (count < 0 || count > kMaxAccumulators) return hipErrorInvalidValue;
  hipLaunchKernelGGL(publishKernel, dim3(1), dim3((kMaxAccumulators + 63) / 64 * 64), 0, stream, d_values, count, pub);
  return hipGetLastError();
}

// ---- resident forms (device-resident LM) ---------------------------------------------------------
int solveSmallMaxTiles() { return kSolveSmallTiles; }

template <typename S>
hipError_t launchP2PSolveSmall(const S *tiles, int num_tiles, const P2PSweepArgs<S> *d_args,
                               const AffineBasis *d_basis, double *result, const LmProblem &problem,
                               const S *x0, int max_points, int fd_cov, hipStream_t stream) {
  if (num_tiles < 1 || num_tiles > kSolveSmallTiles || problem.num_costs != 1 ||
      problem.n != kNumParams || max_points < 1)
    return hipErrorInvalidValue;
  LmStart<S> start;
  for (int i = 0; i < kMaxWideParams; ++i) start.x[i] = i < problem.n ? x0[i] : S(0);
  const dim3 g(1), b(kBlockThreads);
  switch (problem.fd_per_iterate ? fd_cov : -1) {
    case kCovIdentity:
      hipLaunchKernelGGL((p2pSolveSmallKernel<S, kCovIdentity>), g, b, 0, stream, tiles, num_tiles, d_args,
                         d_basis, result, problem, start, max_points);
      break;
    case kCovSymmetric:
      hipLaunchKernelGGL((p2pSolveSmallKernel<S, kCovSymmetric>), g, b, 0, stream, tiles, num_tiles, d_args,
                         d_basis, result, problem, start, max_points);
      break;
    case kCovGeneral:
      hipLaunchKernelGGL((p2pSolveSmallKernel<S, kCovGeneral>), g, b, 0, stream, tiles, num_tiles, d_args,
                         d_basis, result, problem, start, max_points);
      break;
    default:
      if (problem.fd_per_iterate) return hipErrorInvalidValue;  // (needs the literal form: name its covariance)
      hipLaunchKernelGGL((p2pSolveSmallKernel<S, -1>), g, b, 0, stream, tiles, num_tiles, d_args, d_basis,
                         result, problem, start, max_points);
      break;
  }
  return hipGetLastError();
}
template hipError_t launchP2PSolveSmall<float>(const float *, int, const P2PSweepArgs<float> *,
                                               const AffineBasis *, double *, const LmProblem &,
                                               const float *, int, int, hipStream_t);
template hipError_t launchP2PSolveSmall<double>(const double *, int, const P2PSweepArgs<double> *,
                                                const AffineBasis *, double *, const LmProblem &,
                                                const double *, int, int, hipStream_t);

template <typename S>
hipError_t launchP2PMomentsResident(const S *tiles, int num_tiles, const P2PSweepArgs<S> *d_args,
                                    const LmControl *control, int grid, const LaunchSite &site) {
  if (site.streaming)
    hipLaunchKernelGGL((p2pMomentsResidentKernel<S, true>), dim3(grid), dim3(kBlockThreads), 0,
                       site.stream, tiles, num_tiles, d_args, control);
  else
    hipLaunchKernelGGL((p2pMomentsResidentKernel<S, false>), dim3(grid), dim3(kBlockThreads), 0,
                       site.stream, tiles, num_tiles, d_args, control);
  return hipGetLastError();
}
template hipError_t launchP2PMomentsResident<float>(const float *, int, const P2PSweepArgs<float> *,
                                                    const LmControl *, int, const LaunchSite &);
template hipError_t launchP2PMomentsResident<double>(const double *, int,
                                                     const P2PSweepArgs<double> *,
                                                     const LmControl *, int, const LaunchSite &);

namespace {
template <typename S, int JAC>
hipError_t launchLiteralResidentCov(const P2PSweepArgs<S> *d_args, const LmControl *control,
                                    int cov_mode, int grid, const LaunchSite &site) {
  const dim3 g(grid), b(kBlockThreads);
  switch (cov_mode) {
    case kCovIdentity:
      hipLaunchKernelGGL((p2pLinearizeLiteralResidentKernel<S, JAC, kCovIdentity>), g, b, 0,
                         site.stream, d_args, control);
      break;
    case kCovSymmetric:
      hipLaunchKernelGGL((p2pLinearizeLiteralResidentKernel<S, JAC, kCovSymmetric>), g, b, 0,
                         site.stream, d_args, control);
      break;
    default:
      hipLaunchKernelGGL((p2pLinearizeLiteralResidentKernel<S, JAC, kCovGeneral>), g, b, 0,
                         site.stream, d_args, control);
      break;
  }
  return hipGetLastError();
}
}  // namespace

template <typename S>
hipError_t launchP2PLiteralResident(const P2PSweepArgs<S> *d_args, const LmControl *control,
                                    int jac_mode, int cov_mode, int grid, const LaunchSite &site) {
  switch (jac_mode) {
    case kJacAnalytic:
      return launchLiteralResidentCov<S, kJacAnalytic>(d_args, control, cov_mode, grid, site);
    case kJacAnalyticTst:
      return launchLiteralResidentCov<S, kJacAnalyticTst>(d_args, control, cov_mode, grid, site);
    case kJacNumeric:
      return launchForwardDiffResident<S>(d_args, control, cov_mode, grid, site);
    case kJacAnalyticLeft:
      return launchLiteralResidentCov<S, kJacAnalyticLeft>(d_args, control, cov_mode, grid, site);
    case kJacAnalyticRight:
      return launchLiteralResidentCov<S, kJacAnalyticRight>(d_args, control, cov_mode, grid, site);
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launchP2PLiteralResident<float>(const P2PSweepArgs<float> *, const LmControl *,
                                                    int, int, int, const LaunchSite &);
template hipError_t launchP2PLiteralResident<double>(const P2PSweepArgs<double> *,
                                                     const LmControl *, int, int, int,
                                                     const LaunchSite &);

namespace {
template <typename S, int JAC>
hipError_t launchLiteralResidentSetCov(const ResidentSweepSet &set, const LmControl *control,
                                       int cov_mode, const LaunchSite &site) {
  const dim3 g(set.first_block[set.num_costs]), b(kBlockThreads);
  switch (cov_mode) {
    case kCovIdentity:
      hipLaunchKernelGGL((p2pLinearizeLiteralResidentSetKernel<S, JAC, kCovIdentity>), g, b, 0,
                         site.stream, set, control);
      break;
    case kCovSymmetric:
      hipLaunchKernelGGL((p2pLinearizeLiteralResidentSetKernel<S, JAC, kCovSymmetric>), g, b, 0,
                         site.stream, set, control);
      break;
    default:
      hipLaunchKernelGGL((p2pLinearizeLiteralResidentSetKernel<S, JAC, kCovGeneral>), g, b, 0,
                         site.stream, set, control);
      break;
  }
  return hipGetLastError();
}
}  // namespace

template <typename S>
hipError_t launchP2PLiteralResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                       int jac_mode, int cov_mode, const LaunchSite &site) {
  switch (jac_mode) {
    case kJacAnalytic:
      return launchLiteralResidentSetCov<S, kJacAnalytic>(set, control, cov_mode, site);
    case kJacAnalyticTst:
      return launchLiteralResidentSetCov<S, kJacAnalyticTst>(set, control, cov_mode, site);
    case kJacNumeric:
      return launchForwardDiffResidentSet<S>(set, control, cov_mode, site);
    case kJacAnalyticLeft:
      return launchLiteralResidentSetCov<S, kJacAnalyticLeft>(set, control, cov_mode, site);
    case kJacAnalyticRight:
      return launchLiteralResidentSetCov<S, kJacAnalyticRight>(set, control, cov_mode, site);
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launchP2PLiteralResidentSet<float>(const ResidentSweepSet &, const LmControl *,
                                                       int, int, const LaunchSite &);
template hipError_t launchP2PLiteralResidentSet<double>(const ResidentSweepSet &, const LmControl *,
                                                        int, int, const LaunchSite &);

hipError_t launchReprojResident(const ReprojSweepArgs *d_args, const LmControl *control,
                                int cov_mode, int grid, const LaunchSite &site) {
  const dim3 g(grid), b(kBlockThreads);
  switch (cov_mode) {
    case kCovIdentity:
      hipLaunchKernelGGL((reprojResidentKernel<kCovIdentity>), g, b, 0, site.stream, d_args, control);
      break;
    case kCovSymmetric:
      hipLaunchKernelGGL((reprojResidentKernel<kCovSymmetric>), g, b, 0, site.stream, d_args, control);
      break;
    default:
      hipLaunchKernelGGL((reprojResidentKernel<kCovGeneral>), g, b, 0, site.stream, d_args, control);
      break;
  }
  return hipGetLastError();
}

hipError_t launchReprojResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                   int cov_mode, const LaunchSite &site) {
  const dim3 g(set.first_block[set.num_costs]), b(kBlockThreads);
  switch (cov_mode) {
    case kCovIdentity:
      hipLaunchKernelGGL((reprojResidentSetKernel<kCovIdentity>), g, b, 0, site.stream, set, control);
      break;
    case kCovSymmetric:
      hipLaunchKernelGGL((reprojResidentSetKernel<kCovSymmetric>), g, b, 0, site.stream, set, control);
      break;
    default:
      hipLaunchKernelGGL((reprojResidentSetKernel<kCovGeneral>), g, b, 0, site.stream, set, control);
      break;
  }
  return hipGetLastError();
}

namespace {
template <typename S, template <typename> class ModelT>
hipError_t launchScalarResidentFor(const ScalarSweepArgs<S> *d_args, const LmControl *control,
                                   int jac_mode, int cov_mode, int grid, hipStream_t stream) {
  const dim3 g(grid), b(kBlockThreads);
  const bool numeric = (jac_mode == kJacNumeric);
#define MOPT_LAUNCH_SCALAR_RESIDENT(JAC, COV)                                                    \
  hipLaunchKernelGGL((scalarModelResidentKernel<S, ModelT, JAC, COV>), g, b, 0, stream, d_args, \
                     control)
  switch (cov_mode) {
    case kCovIdentity:
      if (numeric) MOPT_LAUNCH_SCALAR_RESIDENT(kJacNumeric, kCovIdentity);
      else MOPT_LAUNCH_SCALAR_RESIDENT(kJacAnalytic, kCovIdentity);
      break;
    case kCovSymmetric:
      if (numeric) MOPT_LAUNCH_SCALAR_RESIDENT(kJacNumeric, kCovSymmetric);
      else MOPT_LAUNCH_SCALAR_RESIDENT(kJacAnalytic, kCovSymmetric);
      break;
    default:
      if (numeric) MOPT_LAUNCH_SCALAR_RESIDENT(kJacNumeric, kCovGeneral);
      else MOPT_LAUNCH_SCALAR_RESIDENT(kJacAnalytic, kCovGeneral);
      break;
  }
#undef MOPT_LAUNCH_SCALAR_RESIDENT
  return hipGetLastError();
}
}  // namespace

namespace {
template <typename S, template <typename> class ModelT>
hipError_t launchScalarResidentSetFor(const ResidentSweepSet &set, const LmControl *control,
                                      int jac_mode, int cov_mode, hipStream_t stream) {
  const dim3 g(set.first_block[set.num_costs]), b(kBlockThreads);
  const bool numeric = (jac_mode == kJacNumeric);
#define MOPT_LAUNCH_SCALAR_SET(JAC, COV)                                                          \
  hipLaunchKernelGGL((scalarModelResidentSetKernel<S, ModelT, JAC, COV>), g, b, 0, stream, set, \
                     control)
  switch (cov_mode) {
    case kCovIdentity:
      if (numeric) MOPT_LAUNCH_SCALAR_SET(kJacNumeric, kCovIdentity);
      else MOPT_LAUNCH_SCALAR_SET(kJacAnalytic, kCovIdentity);
      break;
    case kCovSymmetric:
      if (numeric) MOPT_LAUNCH_SCALAR_SET(kJacNumeric, kCovSymmetric);
      else MOPT_LAUNCH_SCALAR_SET(kJacAnalytic, kCovSymmetric);
      break;
    default:
      if (numeric) MOPT_LAUNCH_SCALAR_SET(kJacNumeric, kCovGeneral);
      else MOPT_LAUNCH_SCALAR_SET(kJacAnalytic, kCovGeneral);
      break;
  }
#undef MOPT_LAUNCH_SCALAR_SET
  return hipGetLastError();
}
}  // namespace

template <typename S>
hipError_t launchScalarModelResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                        int model, int jac_mode, int cov_mode, hipStream_t stream) {
  switch (model) {
    case kScalarExpCurve:
      return launchScalarResidentSetFor<S, ExpCurve>(set, control, jac_mode, cov_mode, stream);
    case kScalarRational:
      return launchScalarResidentSetFor<S, Rational>(set, control, jac_mode, cov_mode, stream);
    case kScalarPowell:
      return launchScalarResidentSetFor<S, Powell>(set, control, jac_mode, cov_mode, stream);
    case kScalarExpCurveMarked:
      return launchScalarResidentSetFor<S, ExpCurveMarked>(set, control, jac_mode, cov_mode, stream);
    case kScalarRationalMarked:
      return launchScalarResidentSetFor<S, RationalMarked>(set, control, jac_mode, cov_mode, stream);
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launchScalarModelResidentSet<float>(const ResidentSweepSet &, const LmControl *,
                                                        int, int, int, hipStream_t);
template hipError_t launchScalarModelResidentSet<double>(const ResidentSweepSet &, const LmControl *,
                                                         int, int, int, hipStream_t);

template <typename S>
hipError_t launchScalarModelResident(const ScalarSweepArgs<S> *d_args, const LmControl *control,
                                     int model, int jac_mode, int cov_mode, int grid,
                                     hipStream_t stream) {
  switch (model) {
    case kScalarExpCurve:
      return launchScalarResidentFor<S, ExpCurve>(d_args, control, jac_mode, cov_mode, grid, stream);
    case kScalarRational:
      return launchScalarResidentFor<S, Rational>(d_args, control, jac_mode, cov_mode, grid, stream);
    case kScalarPowell:
      return launchScalarResidentFor<S, Powell>(d_args, control, jac_mode, cov_mode, grid, stream);
    case kScalarExpCurveMarked:
      return launchScalarResidentFor<S, ExpCurveMarked>(d_args, control, jac_mode, cov_mode, grid, stream);
    case kScalarRationalMarked:
      return launchScalarResidentFor<S, RationalMarked>(d_args, control, jac_mode, cov_mode, grid, stream);
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launchScalarModelResident<float>(const ScalarSweepArgs<float> *,
                                                     const LmControl *, int, int, int, int,
                                                     hipStream_t);
template hipError_t launchScalarModelResident<double>(const ScalarSweepArgs<double> *,
                                                      const LmControl *, int, int, int, int,
                                                      hipStream_t);

hipError_t launchFinalizeDenseResident(const double *partials, int grid, int nacc, int n,
                                       double *result, LmControl *control, hipStream_t stream,
                                       const PeerCombine *peers, const LmProblem *step,
                                       int own_index, int scalar_bytes) {
  if (n < 1 || n > kMaxWideParams || (nacc != n * (n + 1) / 2 + n + 1 && nacc != n * n + n + 1))
    return hipErrorInvalidValue;
  const PeerCombine pc = peers ? *peers : PeerCombine();
  const dim3 g(1), b(kStepThreads);
  if (!step)
    hipLaunchKernelGGL(finalizeDenseResidentKernel<0>, g, b, 0, stream, partials, grid, nacc, n,
                       result, control, pc, LmProblem(), 0);
  else if (scalar_bytes == 8)
    hipLaunchKernelGGL(finalizeDenseResidentKernel<8>, g, b, 0, stream, partials, grid, nacc, n,
                       result, control, pc, *step, own_index);
  else
    hipLaunchKernelGGL(finalizeDenseResidentKernel<4>, g, b, 0, stream, partials, grid, nacc, n,
                       result, control, pc, *step, own_index);
  return hipGetLastError();
}

hipError_t launchFinalizeEitherResident(const double *partials, int grid_moments, int grid_literal,
                                        int nacc, const AffineBasis *d_basis, double *result,
                                        LmControl *control, hipStream_t stream,
                                        const PeerCombine *peers, const LmProblem *step,
                                        int own_index, int scalar_bytes) {
  if (nacc != kAccSym && nacc != kAccFull) return hipErrorInvalidValue;
  const PeerCombine pc = peers ? *peers : PeerCombine();
  const dim3 g(1), b(kStepThreads);
  if (!step)
    hipLaunchKernelGGL(finalizeEitherResidentKernel<0>, g, b, 0, stream, partials, grid_moments,
                       grid_literal, nacc, d_basis, result, control, pc, LmProblem(), 0);
  else if (scalar_bytes == 8)
    hipLaunchKernelGGL(finalizeEitherResidentKernel<8>, g, b, 0, stream, partials, grid_moments,
                       grid_literal, nacc, d_basis, result, control, pc, *step, own_index);
  else
    hipLaunchKernelGGL(finalizeEitherResidentKernel<4>, g, b, 0, stream, partials, grid_moments,
                       grid_literal, nacc, d_basis, result, control, pc, *step, own_index);
  return hipGetLastError();
}

hipError_t launchFinalizeMomentsResident(const double *partials, int grid,
                                         const AffineBasis *d_basis, double *result,
                                         LmControl *control, hipStream_t stream,
                                         const PeerCombine *peers, const LmProblem *step,
                                         int own_index, int scalar_bytes) {
  const PeerCombine pc = peers ? *peers : PeerCombine();
  const dim3 g(1), b(kStepThreads);
  if (!step)
    hipLaunchKernelGGL(finalizeMomentsResidentKernel<0>, g, b, 0, stream, partials, grid, d_basis,
                       result, control, pc, LmProblem(), 0);
  else if (scalar_bytes == 8)
    hipLaunchKernelGGL(finalizeMomentsResidentKernel<8>, g, b, 0, stream, partials, grid, d_basis,
                       result, control, pc, *step, own_index);
  else
    hipLaunchKernelGGL(finalizeMomentsResidentKernel<4>, g, b, 0, stream, partials, grid, d_basis,
                       result, control, pc, *step, own_index);
  return hipGetLastError();
}

}  // namespace mopt
