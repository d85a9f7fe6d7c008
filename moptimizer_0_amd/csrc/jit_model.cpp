// User-defined device models: the GPU counterpart of subclassing moptimizer::IBaseModel.
//
// In the reference a new model is a C++ class whose setup() runs once per parameter vector and
// whose f() (and optionally f_df()) the sweep calls per index through a virtual
// (include/moptimizer/model.h:11-47).  Device code cannot call host virtuals, so here a model is
// the *text* of those function bodies; hipRTC compiles it for gfx950 into the same per-element
// sweep the built-in models use (residual, forward-difference or supplied Jacobian, loss weight,
// w J^T S J / w J^T S r / r^T r accumulation, one partial row per workgroup).  The workgroup rows
// are finished by the library's finalizeDenseKernel.
//
// One kernel is compiled per (sweep kind, covariance form) the cost is actually used with -
// cost only / supplied Jacobian / forward differences, symmetric or general covariance - so that
// each carries only its own registers: a single kernel with run-time switches held both Jacobian
// paths and all n*n accumulators live and ran the point-to-point model at occupancy 1 with
// scratch spills (825 us for 10 M elements against 76 us for the hand-written sweep).
#include "jit_model.hpp"

#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace mopt {
namespace {

// Compiled at run time as  kJitPrologue + the three user functions + kJitSweep;  MOPT_S / MOPT_N /
// MOPT_M / MOPT_D / MOPT_A / MOPT_MODE / MOPT_COV / MOPT_S_BYTES arrive as -D options.
const char *const kJitPrologue = R"JIT(
#define kBlock 256
typedef MOPT_S S;
#define N MOPT_N
#define M MOPT_M
#define D MOPT_D
#define AUX MOPT_A
#define MODE MOPT_MODE       /* 0 cost only, 1 supplied Jacobian, 2 forward differences */
#define COV MOPT_COV         /* 0 identity, 1 symmetric, 2 general covariance */
#define COVSYM (COV != 2)    /* S symmetric -> H symmetric, upper triangle only */
#define NH (COVSYM ? N * (N + 1) / 2 : N * N)
#define NACC (MODE == 0 ? 1 : NH + N + 1)
/* forward differences, fp64 (two elements per pack), per-x values in LDS, a small Jacobian: the two
   elements of a pack evaluated side by side (sweep_elements) */
#define PAIRED (MODE == 2 && MOPT_S_BYTES == 8 && AUX > 0 && M * N <= 18)

struct JitArgs {
  const S *data;       // planes: data[p * stride + i]
  long long count;
  long long stride;
  int loss_kind;       // 0 none, 1 Geman-McClure
  int pad_[3];
  S loss_param;
  S x[8];
  S h[8];
  S cov[16];           // row-major M x M
  double *partials;    // [grid][NACC]
};
)JIT";

const char *const kJitSweep = R"JIT(
// 16-byte loads: a lane takes VEC consecutive elements of every data plane per step (planes start
// 16-byte aligned: the library pads their stride), so one wavefront instruction reads 1 KiB of
// contiguous memory as the hand-written sweeps do.
#define VEC (16 / (int)sizeof(S))
struct __attribute__((aligned(16))) Pack {
  S v[VEC];
};

// Per-thread accumulators -> one row of NACC doubles per workgroup through an LDS transpose (the
// epilogue of the hand-written sweeps, sweep_kernels.hip blockReduceStore): every lane stores its
// values once, thread (k, part) adds the 4 waves x 8 lanes of value k whose lane index is = part
// (mod 8), three xor-shuffles combine the parts.  Six shuffle steps per value through the CU's
// single LDS crossbar cost several times as much.
#define RCHUNK (NACC < 23 ? NACC : 23)
#define RROW 72
__device__ inline void block_reduce_store(double (&acc)[NACC], double *out_row) {
  __shared__ double lds[kBlock / 64][RCHUNK][RROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k_read = threadIdx.x >> 3, part = threadIdx.x & 7;
#pragma unroll
  for (int pass = 0; pass < (NACC + RCHUNK - 1) / RCHUNK; ++pass) {
    if (pass > 0) __syncthreads();
#pragma unroll
    for (int kk = 0; kk < RCHUNK; ++kk) {
      const int k = pass * RCHUNK + kk;
      if (k < NACC) lds[wave][kk][lane] = acc[k];
    }
    __syncthreads();
    const int k_out = pass * RCHUNK + k_read;
    if (k_read < RCHUNK && k_out < NACC) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w)
#pragma unroll
        for (int j = 0; j < 8; ++j) v += lds[w][k_read][j * 8 + part];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      if (part == 0) out_row[k_out] = v;
    }
  }
}

// acc += a . b over M terms: in fp64 a chain of fused multiply-adds straight into the accumulator
// (one instruction per term); in fp32 the sum is formed in fp32 as the reference's float
// instantiation does and added to the fp64 running sum
__device__ inline void acc_dot(double &acc, const S *a, int sa, const S *b, int sb) {
  if (sizeof(S) == 8) {
    double v = acc;
#pragma unroll
    for (int q = 0; q < M; ++q) v = __builtin_fma((double)a[q * sa], (double)b[q * sb], v);
    acc = v;
  } else {
    S v = 0;
#pragma unroll
    for (int q = 0; q < M; ++q) v += a[q * sa] * b[q * sb];
    acc += (double)v;
  }
}

struct RobustLoss { static constexpr bool value = true; };
struct NoLoss { static constexpr bool value = false; };

// `Loss`: the loss kind is a property of the whole sweep, not of the element — a branch inside the
// element would split its basic block and let the compiler sink one element's accumulation below
// the next element's residual evaluations (then both elements' Jacobians are live at once).
template <typename Loss>
__device__ inline void sweep_elements(const JitArgs &A, const S *aux, double (&acc)[NACC], int block,
                                      int num_blocks) {
#if MODE == 2
  // the quotient (r+ - r) / h_j (linearization.h:105) as a product with 1 / h_j, formed once: M * N
  // fp64 divisions per element were a quarter of the forward-difference sweep's instructions (the
  // hand-written sweeps receive 1 / h_j from the host for the same reason; <= 1 ulp per entry)
  S inv_h[N];
#pragma unroll
  for (int j = 0; j < N; ++j) inv_h[j] = S(1) / A.h[j];
#endif
#if MODE != 0
  // w J^T S J, w J^T S r, r^T r of one element into the sums (linearization.h:113-115, :150-152)
  auto accumulate = [&](const S (&r)[M], S rr, const S (&J)[M * N], bool valid) {
    S w = 1;
    if (Loss::value) {
      const S den = rr + A.loss_param;
      w = (A.loss_param * A.loss_param) / (den * den);
    }
    w = valid ? w : S(0);
    // under the identity covariance S J is J
    S wJ[M * N];
#pragma unroll
    for (int q = 0; q < M * N; ++q) wJ[q] = w * J[q];
#if COV == 0
    const S *SJ = J;
    const S *Sr = r;
#else
    S SJ[M * N], Sr[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        S v = 0;
#pragma unroll
        for (int c = 0; c < M; ++c) v += A.cov[a * M + c] * J[c * N + j];
        SJ[a * N + j] = v;
      }
      S v = 0;
#pragma unroll
      for (int c = 0; c < M; ++c) v += A.cov[a * M + c] * r[c];
      Sr[a] = v;
    }
#endif
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int i2 = 0; i2 < (COVSYM ? j + 1 : N); ++i2)
        acc_dot(acc[COVSYM ? j * (j + 1) / 2 + i2 : j * N + i2], wJ + i2, N, SJ + j, N);   // H(i2, j)
#pragma unroll
    for (int i2 = 0; i2 < N; ++i2) acc_dot(acc[NH + i2], wJ + i2, N, Sr, 1);
    acc[NH + N] += (double)rr;
  };
#endif
  // one element: residual, Jacobian (supplied or by forward differences), loss weight, the sums
  auto element = [&](const S (&d)[D > 0 ? D : 1], bool valid) {
    // the per-x values stay in LDS: without this the compiler keeps all (N + 1) * AUX of them in
    // registers across the loop
    asm volatile("" ::: "memory");
    S r[M];
    // an index the model rejects (f / f_df returning false: model.h:32, linearization.h:102,144) is
    // skipped like a slot past the end: weight zero, and its values — which may be anything, NaN
    // included — replaced by zeros so that 0 * value stays 0
    // (`ok` is a constant for a body that never touches `valid`: the selects on it then fold away;
    // a slot past the end needs none — it is evaluated on the pack's first element, which is finite)
    bool ok = user_residual(A.x, aux, d, r);
    S rr = 0;
#pragma unroll
    for (int a = 0; a < M; ++a) {
      r[a] = ok ? r[a] : S(0);
      rr += r[a] * r[a];
    }
    rr = valid ? rr : S(0);
#if MODE == 0
    acc[0] += (double)rr;
#else
    S J[M * N];  // row-major, as IBaseModel::f_df fills it
#if MODE == 2
#pragma unroll
    for (int j = 0; j < N; ++j) {
      S xp[N];
#pragma unroll
      for (int k = 0; k < N; ++k) xp[k] = A.x[k];
      xp[j] += A.h[j];                                   // linearization.h:89
      asm volatile("" ::: "memory");                     // fetch this vector's per-x values now
      S rp[M];
      (void)user_residual(xp, aux + (1 + j) * AUX, d, rp);   // its result is ignored, :104
#pragma unroll
      for (int a = 0; a < M; ++a) J[a * N + j] = ok ? (rp[a] - r[a]) * inv_h[j] : S(0);   // :105
    }
#else
    ok = user_jacobian(A.x, aux, d, J) && ok;
#pragma unroll
    for (int q = 0; q < M * N; ++q) J[q] = ok ? J[q] : S(0);
    rr = ok ? rr : S(0);   // (selects, not a branch: a branch splits the element's basic block)
#pragma unroll
    for (int a = 0; a < M; ++a) r[a] = ok ? r[a] : S(0);
#endif
    accumulate(r, rr, J, valid && ok);
#endif
  };
#if PAIRED
  // Forward differences over a model with per-x values (setup output): every evaluation of the
  // residual reads its parameter vector's values from LDS, all lanes the same address — and such a
  // read still returns 1 KiB to the wave, so (N + 1) * AUX of them per element set the pace (the
  // point2point model written as a user model: 84 values, 143 us for 10 M elements).  The two
  // elements of a lane's pack are therefore evaluated side by side, parameter vector by parameter
  // vector, on one read of the values; their Jacobians are live together (2 M N registers more),
  // which is why this form is kept to small M N.
  auto element_pair = [&](const S (&d0)[D > 0 ? D : 1], const S (&d1)[D > 0 ? D : 1], bool valid0,
                          bool valid1) {
    asm volatile("" ::: "memory");
    S r0[M], r1[M];
    const bool ok0 = user_residual(A.x, aux, d0, r0);   // (rejected indices: as in element())
    const bool ok1 = user_residual(A.x, aux, d1, r1);
    S rr0 = 0, rr1 = 0;
#pragma unroll
    for (int a = 0; a < M; ++a) {
      r0[a] = ok0 ? r0[a] : S(0);
      r1[a] = ok1 ? r1[a] : S(0);
      rr0 += r0[a] * r0[a];
      rr1 += r1[a] * r1[a];
    }
    rr0 = valid0 ? rr0 : S(0);
    rr1 = valid1 ? rr1 : S(0);
    valid0 = valid0 && ok0;
    valid1 = valid1 && ok1;
    S J0[M * N], J1[M * N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      S xp[N];
#pragma unroll
      for (int k = 0; k < N; ++k) xp[k] = A.x[k];
      xp[j] += A.h[j];                                   // linearization.h:89
      asm volatile("" ::: "memory");                     // fetch this vector's per-x values now
      S rp0[M], rp1[M];
      (void)user_residual(xp, aux + (1 + j) * AUX, d0, rp0);   // results ignored, :104
      (void)user_residual(xp, aux + (1 + j) * AUX, d1, rp1);
#pragma unroll
      for (int a = 0; a < M; ++a) {
        J0[a * N + j] = ok0 ? (rp0[a] - r0[a]) * inv_h[j] : S(0);   // :105
        J1[a * N + j] = ok1 ? (rp1[a] - r1[a]) * inv_h[j] : S(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    accumulate(r0, rr0, J0, valid0);
    __builtin_amdgcn_sched_barrier(0);
    accumulate(r1, rr1, J1, valid1);
  };
#endif

  const long long step = (long long)num_blocks * kBlock * VEC;
  long long i = ((long long)block * kBlock + threadIdx.x) * VEC;
  Pack cur[D > 0 ? D : 1], nxt[D > 0 ? D : 1];
  if (i < A.count) {
#pragma unroll
    for (int p = 0; p < D; ++p) nxt[p] = *(const Pack *)(A.data + p * A.stride + i);
  }
  for (; i < A.count; i += step) {
#pragma unroll
    for (int p = 0; p < D; ++p) cur[p] = nxt[p];
    // the next step's data are requested before this one's arithmetic starts
    const long long ahead = i + step < A.count ? i + step : i;
#pragma unroll
    for (int p = 0; p < D; ++p) nxt[p] = *(const Pack *)(A.data + p * A.stride + ahead);
    // an element past the end (the padded tail of the last pack) is evaluated on the pack's first
    // element, which is in range, and enters every sum with weight zero
#if MODE == 2 && MOPT_S_BYTES == 4
    // forward differences evaluate the residual N + 1 times per element; four fp32 elements per
    // pack are taken in a real loop (unrolled they need more than 256 registers), the element
    // picked by selects
#pragma unroll 1
    for (int e = 0; e < VEC; ++e) {
      const bool valid = i + e < A.count;
      S d[D > 0 ? D : 1];
#pragma unroll
      for (int p = 0; p < D; ++p) {
        d[p] = cur[p].v[0];
#pragma unroll
        for (int k = 1; k < VEC; ++k) d[p] = (valid && e == k) ? cur[p].v[k] : d[p];
      }
      element(d, valid);
    }
#elif PAIRED
    {
      const bool valid1 = i + 1 < A.count;
      S d0[D > 0 ? D : 1], d1[D > 0 ? D : 1];
#pragma unroll
      for (int p = 0; p < D; ++p) {
        d0[p] = cur[p].v[0];
        d1[p] = valid1 ? cur[p].v[1] : cur[p].v[0];
      }
      __builtin_amdgcn_sched_barrier(0);
      element_pair(d0, d1, true, valid1);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#if MODE == 2
      __builtin_amdgcn_sched_barrier(0);  // one element after the other: the scheduler would
#endif                                    // interleave them and double the live Jacobian entries
      const bool valid = i + e < A.count;
      S d[D > 0 ? D : 1];
#pragma unroll
      for (int p = 0; p < D; ++p) d[p] = valid ? cur[p].v[e] : cur[p].v[0];
      element(d, valid);
    }
#if MODE == 2
    __builtin_amdgcn_sched_barrier(0);
#endif
#endif
  }
}

// block / num_blocks: this workgroup's place among those sweeping this cost (the whole grid, or its
// share of a launch that carries several costs)
__device__ inline void sweep_body(const JitArgs &A, int block, int num_blocks) {
  // IBaseModel::setup (model.h:19-22): once per parameter vector, here once per workgroup and
  // parameter vector - x itself and, for forward differences, x + h_j e_j (the reference sets
  // up one clone of the model per perturbed vector, linearization.h:91-95).
  __shared__ __attribute__((aligned(16))) S aux[(N + 1) * (AUX > 0 ? AUX : 1)];
  if (AUX > 0) {
    const int variants = MODE == 2 ? N + 1 : 1;
    if ((int)threadIdx.x < variants) {
      S xs[N];
#pragma unroll
      for (int k = 0; k < N; ++k) xs[k] = A.x[k] + ((int)threadIdx.x == k + 1 ? A.h[k] : S(0));
      user_setup(xs, aux + threadIdx.x * AUX);
    }
    __syncthreads();
  }
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  if (MODE != 0 && A.loss_kind == 1)
    sweep_elements<RobustLoss>(A, aux, acc, block, num_blocks);
  else
    sweep_elements<NoLoss>(A, aux, acc, block, num_blocks);
  block_reduce_store(acc, A.partials + (size_t)block * NACC);
}

extern "C" __global__ __launch_bounds__(kBlock) void mopt_jit_sweep(const JitArgs A) {
  sweep_body(A, blockIdx.x, gridDim.x);
}

// Resident form for the device-resident LM (mopt_lm_minimize): the argument block — x and the
// forward-difference steps in it rewritten by the LM step kernel for every trial point — is read
// from HBM, and a launch queued past the end of the minimisation returns at once.
struct LmControl {
  int done, trial, pad[2];
};
extern "C" __global__ __launch_bounds__(kBlock) void mopt_jit_sweep_resident(
    const JitArgs *__restrict__ d_args, const LmControl *__restrict__ control) {
  if (control->done) return;
  const JitArgs A = *d_args;
  sweep_body(A, blockIdx.x, gridDim.x);
}

// Several costs over this same model (same source, same shape) in one launch: workgroups
// [first_block[k], first_block[k + 1]) sweep cost k (sweep.hpp ResidentSweepSet).
struct ResidentSweepSet {
  const void *args[4];
  int first_block[5];
  int num_costs;
};
extern "C" __global__ __launch_bounds__(kBlock) void mopt_jit_sweep_resident_set(
    const ResidentSweepSet set, const LmControl *__restrict__ control) {
  if (control->done) return;
  int k = 0;
  while (k + 1 < set.num_costs && (int)blockIdx.x >= set.first_block[k + 1]) ++k;
  const JitArgs A = *(const JitArgs *)set.args[k];
  sweep_body(A, (int)blockIdx.x - set.first_block[k], set.first_block[k + 1] - set.first_block[k]);
}
)JIT";


// ---- wide models: 8 < n <= 16 or 4 < m <= 16 ---------------------------------------------------
// The per-thread sweep above keeps the m x n Jacobian and n(n+1)/2 + n + 1 sums of an element in
// one lane's registers; at n = m = 15 (tst/state_model.cpp:83-112) that is 225 + 136 values.  Here
// an element is worked on by 16 lanes, lane j owning column j: it evaluates the residual at x and
// at x + h_j e_j (or reads its column of the supplied Jacobian, which one lane of the element wrote
// to LDS), forms S J(:, j), and accumulates H(:, j) = sum_a (w J(a, :))^T (S J)(a, j) against the
// other lanes' columns in LDS, plus b(j) — the same products, in the same order, as above
// (linearization.h:105-115, :144-152).  256 threads = 16 elements x 16 lanes; rows are always the
// full form (n*n + n + 1).  These models come with a handful of residuals (a state prior, an IMU
// term): the layout is for register pressure, not for bandwidth.
const char *const kJitWidePrologue = R"JIT(
#define kBlock 256
typedef MOPT_S S;
#define N MOPT_N
#define M MOPT_M
#define D MOPT_D
#define AUX MOPT_A
#define MODE MOPT_MODE       /* 0 cost only, 1 supplied Jacobian, 2 forward differences */
#define COLS 16
#define EPB (kBlock / COLS)
#define NACC (MODE == 0 ? 1 : N * N + N + 1)

struct JitArgs {
  const S *data;       // planes: data[p * stride + i]
  long long count;
  long long stride;
  int loss_kind;       // 0 none, 1 Geman-McClure
  int pad_[3];
  S loss_param;
  S x[16];
  S h[16];
  S cov[256];          // row-major M x M
  double *partials;    // [grid][NACC]
};
)JIT";

const char *const kJitWideSweep = R"JIT(
__device__ inline void wide_body(const JitArgs &A) {
  __shared__ S aux[(N + 1) * (AUX > 0 ? AUX : 1)];
  // Jacobians of the 16 elements in flight (row-major M x N each); afterwards the lanes' sums
  __shared__ double pool[kBlock * (N + 2)];
  S *ldsJ = (S *)pool;
  if (AUX > 0) {
    const int variants = MODE == 2 ? N + 1 : 1;
    if ((int)threadIdx.x < variants) {
      S xs[N];
#pragma unroll
      for (int k = 0; k < N; ++k) xs[k] = A.x[k] + ((int)threadIdx.x == k + 1 ? A.h[k] : S(0));
      user_setup(xs, aux + threadIdx.x * AUX);
    }
    __syncthreads();
  }
  const int e = threadIdx.x / COLS, j = threadIdx.x % COLS;
  const int jc = j < N ? j : 0;   // lanes beyond the last column repeat column 0 and add nothing
  const bool owner = j < N;
  double acc[N], acc_b = 0.0, acc_s = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = 0.0;
#if MODE == 2
  S xj[N], inv_h = S(0);   // this lane's perturbed vector x + h_j e_j (linearization.h:89) and 1 / h_j
#pragma unroll
  for (int k = 0; k < N; ++k) {
    xj[k] = A.x[k] + (k == jc ? A.h[k] : S(0));
    inv_h = k == jc ? S(1) / A.h[k] : inv_h;
  }
#endif
  S *Je = ldsJ + e * (M * N);
  for (long long base = (long long)blockIdx.x * EPB; base < A.count;
       base += (long long)gridDim.x * EPB) {
    const long long i = base + e;
    const bool valid = i < A.count;   // the padding slots repeat the last element with weight zero
    const long long at = valid ? i : A.count - 1;
    S d[D > 0 ? D : 1];
#pragma unroll
    for (int p = 0; p < D; ++p) d[p] = A.data[p * A.stride + at];
    S r[M];
    // an index the model rejects (f / f_df returning false: model.h:32, linearization.h:102,144) is
    // skipped like a padding slot: weight zero, its values replaced by zeros (0 * NaN is NaN)
    // (`ok` is the model's verdict alone — a constant for a body that never touches `valid`, and the
    // selects on it fold away; padding slots repeat the last element, which is finite)
    bool ok = user_residual(A.x, aux, d, r);
#if MODE == 1
    // f_df's own verdict: lane 0 of the element evaluates it, the element's lanes share it
    {
      const bool jac_ok = j == 0 ? user_jacobian(A.x, aux, d, Je) : true;   // row-major M x N
      ok = ok && __shfl(jac_ok ? 1 : 0, (int)(threadIdx.x & 63) - j, 64) != 0;
    }
#endif
    S rr = 0;
#pragma unroll
    for (int a = 0; a < M; ++a) {
      r[a] = ok ? r[a] : S(0);
      rr += r[a] * r[a];
    }
    rr = valid ? rr : S(0);
#if MODE == 0
    if (j == 0) acc_s += (double)rr;
#else
    S Jc[M];
#if MODE == 2
    {
      S rp[M];
      (void)user_residual(xj, aux + (1 + jc) * AUX, d, rp);   // its result is ignored, :104
#pragma unroll
      for (int a = 0; a < M; ++a) Jc[a] = ok ? (rp[a] - r[a]) * inv_h : S(0);   // :105
      if (owner) {
#pragma unroll
        for (int a = 0; a < M; ++a) Je[a * N + j] = Jc[a];
      }
    }
    __syncthreads();
#else
    __syncthreads();
#pragma unroll
    for (int a = 0; a < M; ++a) Jc[a] = ok ? Je[a * N + jc] : S(0);
#endif
    S w = 1;
    if (A.loss_kind == 1) {
      const S den = rr + A.loss_param;
      w = (A.loss_param * A.loss_param) / (den * den);
    }
    w = (ok && valid) ? w : S(0);
    S SJc[M], Sr[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
      S v = 0, u = 0;
#pragma unroll
      for (int c = 0; c < M; ++c) {
        v += A.cov[a * M + c] * Jc[c];
        u += A.cov[a * M + c] * r[c];
      }
      SJc[a] = v;
      Sr[a] = u;
    }
    if (owner) {
#pragma unroll
      for (int i2 = 0; i2 < N; ++i2) {
        S v = 0;
#pragma unroll
        for (int a = 0; a < M; ++a) v += (w * (ok ? Je[a * N + i2] : S(0))) * SJc[a];
        acc[i2] += (double)v;   // H(i2, j)
      }
      S v = 0;
#pragma unroll
      for (int a = 0; a < M; ++a) v += (w * Jc[a]) * Sr[a];
      acc_b += (double)v;       // b(j)
    }
    if (j == 0) acc_s += (double)rr;
    __syncthreads();            // the Jacobians are rewritten by the next step
#endif
  }
  __syncthreads();
  double *mine = pool + (size_t)threadIdx.x * (N + 2);
#pragma unroll
  for (int k = 0; k < N; ++k) mine[k] = acc[k];
  mine[N] = acc_b;
  mine[N + 1] = acc_s;
  __syncthreads();
  double *out_row = A.partials + (size_t)blockIdx.x * NACC;
  for (int t = threadIdx.x; t < NACC; t += kBlock) {
    int col = 0, k = N + 1;             // the sum of squares: lane 0 of every element
#if MODE != 0
    if (t < N * N) {
      col = t / N; k = t % N;            // H(k, col), column-major
    } else if (t < N * N + N) {
      col = t - N * N; k = N;            // b(col)
    }
#endif
    double v = 0.0;
    for (int s = 0; s < EPB; ++s) v += pool[(size_t)(s * COLS + col) * (N + 2) + k];
    out_row[t] = v;
  }
}

extern "C" __global__ __launch_bounds__(kBlock) void mopt_jit_sweep(const JitArgs A) { wide_body(A); }

// Resident form for the device-resident LM, as the narrow sweep's: x and the forward-difference
// steps in the argument block are rewritten by the LM step kernel for every trial point.
struct LmControl {
  int done, trial, pad[2];
};
extern "C" __global__ __launch_bounds__(kBlock) void mopt_jit_sweep_resident(
    const JitArgs *__restrict__ d_args, const LmControl *__restrict__ control) {
  if (control->done) return;
  wide_body(*d_args);
}
)JIT";

std::string &jitError() {
  static thread_local std::string e;
  return e;
}

bool compileVariant(JitKernel &k, int mode, int cov_mode, JitVariant &out) {
  const std::string defs[] = {
      std::string("-DMOPT_S=") + (k.scalar_bytes == 8 ? "double" : "float"),
      "-DMOPT_N=" + std::to_string(k.n_params),
      "-DMOPT_M=" + std::to_string(k.n_outputs),
      "-DMOPT_D=" + std::to_string(k.n_planes),
      "-DMOPT_A=" + std::to_string(k.n_aux),
      "-DMOPT_MODE=" + std::to_string(mode),
      "-DMOPT_COV=" + std::to_string(cov_mode),
      "-DMOPT_S_BYTES=" + std::to_string(k.scalar_bytes),
      "--offload-arch=gfx950",
      "-O3",
      "-std=c++17",
  };
  std::vector<const char *> opts;
  for (const auto &d : defs) opts.push_back(d.c_str());

  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, k.source.c_str(), "mopt_jit_model.hip", 0, nullptr, nullptr) !=
      HIPRTC_SUCCESS) {
    jitError() = "hiprtcCreateProgram failed";
    return false;
  }
  const hiprtcResult rc = hiprtcCompileProgram(prog, int(opts.size()), opts.data());
  if (rc != HIPRTC_SUCCESS) {
    size_t log_size = 0;
    hiprtcGetProgramLogSize(prog, &log_size);
    std::string log(log_size, '\0');
    if (log_size) hiprtcGetProgramLog(prog, &log[0]);
    jitError() = std::string("model source did not compile: ") + hiprtcGetErrorString(rc) + "\n" + log;
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t code_size = 0;
  hiprtcGetCodeSize(prog, &code_size);
  std::vector<char> code(code_size);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  // MOPT_JIT_DUMP_DIR=<dir>: keep every compiled code object (n, m, mode, covariance form in the
  // name) for llvm-objdump / llvm-readelf --notes — register counts, spills, scratch of a model
  if (const char *dir = getenv("MOPT_JIT_DUMP_DIR")) {
    const std::string path = std::string(dir) + "/mopt_jit_n" + std::to_string(k.n_params) + "_m" +
                             std::to_string(k.n_outputs) + "_s" + std::to_string(k.scalar_bytes) +
                             "_mode" + std::to_string(mode) + "_cov" + std::to_string(cov_mode) +
                             (k.wide ? "_wide" : "") + ".co";
    if (FILE *f = fopen(path.c_str(), "wb")) {
      fwrite(code.data(), 1, code.size(), f);
      fclose(f);
    }
  }
  if (hipModuleLoadData(&out.module, code.data()) != hipSuccess) {
    jitError() = "hipModuleLoadData failed for the compiled model";
    return false;
  }
  if (hipModuleGetFunction(&out.sweep, out.module, "mopt_jit_sweep") != hipSuccess ||
      hipModuleGetFunction(&out.sweep_resident, out.module, "mopt_jit_sweep_resident") != hipSuccess) {
    jitError() = "compiled model has no mopt_jit_sweep";
    (void)hipModuleUnload(out.module);
    out.module = nullptr;
    out.sweep = out.sweep_resident = nullptr;
    return false;
  }
  // (the narrow sweep only; absent from the wide module)
  if (hipModuleGetFunction(&out.sweep_resident_set, out.module, "mopt_jit_sweep_resident_set") !=
      hipSuccess) {
    (void)hipGetLastError();
    out.sweep_resident_set = nullptr;
  }
  return true;
}

}  // namespace

const char *jitLastError() { return jitError().c_str(); }

bool jitCreate(int scalar_bytes, int n_params, int n_outputs, int n_planes, int n_aux,
               const char *setup_body, const char *residual_body, const char *jacobian_body,
               JitKernel &out) {
  if (!residual_body || n_params < 1 || n_params > kMaxWideParams || n_outputs < 1 ||
      n_outputs > kMaxWideOutputs || n_planes < 0 || n_planes > 16 || n_aux < 0 || n_aux > 64 ||
      (n_aux > 0 && !setup_body)) {
    jitError() = "bad model shape (1 <= n <= 16, 1 <= m <= 16, planes <= 16, aux <= 64 with a setup "
                 "source) or no residual source";
    return false;
  }
  out.wide = n_params > kMaxParams || n_outputs > 4;
  out.scalar_bytes = scalar_bytes;
  out.n_params = n_params;
  out.n_outputs = n_outputs;
  out.n_planes = n_planes;
  out.n_aux = n_aux;
  out.has_jacobian = jacobian_body && *jacobian_body;
  out.source = out.wide ? kJitWidePrologue : kJitPrologue;
  out.source += "__device__ inline void user_setup(const S *x, S *a) {\n";
  out.source += (n_aux > 0 && setup_body) ? setup_body : "";
  // IBaseModel::f / f_df return bool (model.h:32,43): "this index is not a residual" — the loops skip
  // it (linearization.h:102,144).  A body says so by clearing the reserved local `valid`.
  out.source += "\n}\n__device__ inline bool user_residual(const S *x, const S *a, const S *d, S *r) {\n"
                "  bool valid = true;\n";
  out.source += residual_body;
  out.source += "\n  return valid;\n}\n"
                "__device__ inline bool user_jacobian(const S *x, const S *a, const S *d, S *J) {\n"
                "  bool valid = true;\n";
  out.source += jacobian_body ? jacobian_body : "";
  out.source += "\n  return valid;\n}\n";
  out.source += out.wide ? kJitWideSweep : kJitSweep;
  // Errors in the user's text must surface at construction: build the sweeps that touch each
  // body now (cost only: setup + residual; supplied Jacobian); the others on first use.
  if (!jitVariant(out, 0, kCovIdentity)) return false;
  if (out.has_jacobian && !jitVariant(out, 1, kCovIdentity)) return false;
  return true;
}

const JitVariant *jitVariant(JitKernel &k, int mode, int cov_mode) {
  // cost only: one form; wide models: one row form (n*n + n + 1), covariance always applied
  if (mode == 0) cov_mode = kCovIdentity;
  if (k.wide && mode != 0) cov_mode = kCovGeneral;
  const int key = mode == 0 ? 0 : 1 + (mode - 1) * 3 + cov_mode;
  JitVariant &v = k.variants[key];
  if (!v.sweep && !compileVariant(k, mode, cov_mode, v)) return nullptr;
  return &v;
}

void jitRelease(JitKernel &k) {
  for (auto &v : k.variants) {
    if (v.module) (void)hipModuleUnload(v.module);
    v.module = nullptr;
    v.sweep = v.sweep_resident = v.sweep_resident_set = nullptr;
  }
}

hipError_t jitLaunch(const JitVariant &v, const void *args, size_t args_bytes, int grid,
                     hipStream_t stream) {
  size_t size = args_bytes;
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void *>(args),
                    HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(v.sweep, unsigned(grid), 1, 1, kBlockThreads, 1, 1, 0, stream, nullptr,
                               config);
}

hipError_t jitLaunchResident(const JitVariant &v, const void *d_args, const LmControl *control,
                             int grid, hipStream_t stream) {
  struct {
    const void *args;
    const LmControl *control;
  } params = {d_args, control};
  size_t size = sizeof params;
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &params, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size,
                    HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(v.sweep_resident, unsigned(grid), 1, 1, kBlockThreads, 1, 1, 0, stream,
                               nullptr, config);
}

hipError_t jitLaunchResidentSet(const JitVariant &v, const ResidentSweepSet &set,
                                const LmControl *control, hipStream_t stream) {
  if (!v.sweep_resident_set) return hipErrorInvalidValue;
  static_assert(kLmMaxCosts == 4, "the run-time compiled source declares ResidentSweepSet for 4 costs");
  struct {
    ResidentSweepSet set;
    const LmControl *control;
  } params = {set, control};
  size_t size = sizeof params;
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &params, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size,
                    HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(v.sweep_resident_set, unsigned(set.first_block[set.num_costs]), 1, 1,
                               kBlockThreads, 1, 1, 0, stream, nullptr, config);
}

}  // namespace mopt
