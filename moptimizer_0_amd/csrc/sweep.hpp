// Internal interface between the C ABI (c_abi.cpp) and the gfx950 kernels (sweep_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include "aql.hpp"

#include <cstdint>

namespace mopt {

constexpr int kBlockThreads = 256;  // 4 wavefronts of 64
constexpr int kNumParams = 6;       // (t, w) of SE(3)
constexpr int kMaxParams = 8;       // small parametric models (scalarModelKernel)
constexpr int kMaxWideParams = 16;  // run-time compiled models with n > 8 or m > 4 (jit_model.cpp,
constexpr int kMaxWideOutputs = 16; // "wide" sweep: tst/state_model.cpp is n = m = 15)

// ---- HBM layout ----------------------------------------------------------------------------
// Correspondences live in HBM as *tiled structure-of-arrays*: a tile holds kBlockThreads * V
// consecutive correspondences (V = 16 B / sizeof(Scalar): 2 in fp64, 4 in fp32) as six planes
// sx | sy | sz | tx | ty | tz of tile-length each.  One wavefront load instruction then reads
// 64 lanes x 16 B = 1 KiB of contiguous memory, a workgroup's six loads cover one contiguous
// 24 KiB tile, and consecutive workgroups walk consecutive tiles — the whole sweep is a single
// linear stream over count * 48 B (fp64).  The last tile is zero-padded and masked by index.
template <typename S>
struct TileShape {
  static constexpr int kVec = 16 / int(sizeof(S));
  static constexpr int kPoints = kBlockThreads * kVec;
  static constexpr int kP2PPlanes = 6;
  static constexpr int kP2PScalars = kPoints * kP2PPlanes;
};
// Reprojection elements: planes px | py | pz | pw (fp64) followed by interleaved int32 (u, v)
// pairs; 256 elements per tile — one per lane of a workgroup —, 10 KiB, 40 B per element.
constexpr int kReprojTilePoints = kBlockThreads;
constexpr int kReprojTileBytes = kReprojTilePoints * (4 * 8 + 2 * 4);

enum JacMode : int { kJacAnalytic = 0, kJacAnalyticTst = 1, kJacNumeric = 2, kJacAnalyticLeft = 3,
                     kJacAnalyticRight = 4 };
enum CovMode : int { kCovIdentity = 0, kCovSymmetric = 1, kCovGeneral = 2 };
enum LossKind : int { kLossNone = 0, kLossGemanMcClure = 1 };

// Accumulator counts per workgroup partial.
constexpr int kAccSym = 21 + 6 + 1;   // upper triangle of H | b | sum_sq
constexpr int kAccFull = 36 + 6 + 1;  // full H (non-symmetric covariance) | b | sum_sq
constexpr int kAccMoments = 23;       // w | w p (3) | w p p^T (6) | w r (3) | w p r^T (9) | r^T r
constexpr int kAccCost = 1;
constexpr int kResultDoubles = kNumParams * kNumParams + kNumParams + 1;  // 43

// Per-sweep constants, passed by value as the kernel argument.
template <typename S>
struct P2PSweepArgs {
  const S *tiles;
  long long count;
  int num_tiles;
  int loss_kind;
  S loss_param;
  S T[1 + kNumParams][12];  // row-major [R | t] at x and at x + h_j e_j
  S inv_h[kNumParams];      // 1 / h_j (forward-difference steps)
  S cov[9];                 // row-major m x m (cov[a * 3 + c] = S(a, c))
  double *partials;         // [grid][num accumulators]
};

struct ReprojSweepArgs {
  const unsigned char *tiles;
  long long count;
  int num_tiles;
  int loss_kind;
  double loss_param;
  double M[1 + kNumParams][12];  // row-major 3x4 projection (K T C) at x and x + h_j e_j
  double inv_h[kNumParams];
  double cov[4];  // row-major 2 x 2
  double *partials;
};

// Affine-Jacobian basis for the moment finalisation: J(p) = J0 + px Jx + py Jy + pz Jz,
// each row-major 3 x 6, plus the covariance (row-major 3 x 3).
struct AffineBasis {
  double J[4][18];
  double cov[9];
};

// Small parametric models over per-element scalar data (planes data[p * stride + i]).
// (the *Marked kinds: the same models over observations some of which carry the NaN marker "not a
// residual" — chosen by the library when it finds one in the data, so that data without markers, the
// reference's own, run the kernels that do not look for them)
enum ScalarModelKind : int {
  kScalarExpCurve = 1,
  kScalarRational = 2,
  kScalarPowell = 3,
  kScalarExpCurveMarked = 4,
  kScalarRationalMarked = 5
};

template <typename S>
struct ScalarSweepArgs {
  const S *data;
  long long count;
  long long stride;
  int loss_kind;
  S loss_param;
  S x[kMaxParams];
  S h[kMaxParams];  // forward-difference steps (linearization.h:85-87)
  S cov[16];        // row-major m x m
  double *partials;
};

// Correspondence search (the model's update(x) step for ICP): targets binned into a uniform grid
// (`reach` cell edges >= the maximum correspondence distance, so the (2 reach + 1)^3 cells around a
// query hold every candidate; icp.cpp picks reach by the density of the targets), stored cell by
// cell as padded 4-vectors with a prefix table of cell starts.
template <typename S>
struct IcpMatchArgs {
  S *tiles;               // point2point tile layout: source planes read, target planes written
  long long count;
  int num_tiles;
  const S *sorted;        // [num_targets][4]  (x, y, z, unused), grouped by cell
  const int *cell_start;  // [cells + 1]
  S origin[3];
  S inv_cell;
  S cell;                 // cell edge: reach cells cover the maximum correspondence distance
  int reach;              // >= 1
  int dims[3];
  S max_dist2;
  S T[12];                // row-major [R | t] applied to the source before the search
  unsigned int *matched;  // optional: matched sources per wave, [workgroups x waves per workgroup]
  int lock_rings;         // second round: rings of rows up to this one in lock step, the rest row by row
};

// Where a sweep kernel is launched.
struct LaunchSite {
  hipStream_t stream = nullptr;
  bool streaming = false;  // non-temporal loads: data set larger than the Infinity Cache
  // when set, the sweep kernel's own dispatch is timestamped into these events (hipExtLaunch):
  // exactly the kernel's duration, as a kernel trace reports it
  hipEvent_t time_start = nullptr, time_stop = nullptr;
  // direct dispatch (aql.hpp): when `aql.queue` is set the launch helper writes the packet itself and
  // reports it through *aql_used; when that fails (kernel not found) it launches on `stream` instead
  mopt_detail::AqlSite aql;
  bool *aql_used = nullptr;
};

// Optional hand-over of a finalize kernel's 43 (or 1) results straight into mapped host memory:
// the values are stored, fenced at system scope, and then `*flag = sequence` is released, so a
// host thread polling the flag needs neither a copy nor a stream synchronisation.
struct HostPublish {
  double *host_result = nullptr;           // mapped, fine-grained host memory (device pointer)
  unsigned long long *host_flag = nullptr;
  unsigned long long sequence = 0;
  unsigned long long *host_status = nullptr;  // optional: receives kStatus* before the flag
};

// ---- shard combine without a collective launch ----------------------------------------------
// A sweep over sharded correspondences ends in n*n + n + 1 sums per rank that must be added over
// the ranks (the host accumulation of levenberg_marquadt_dyn.cpp:57-59, across shards).  The
// message is 344 bytes, so what matters is latency, not bandwidth: instead of a collective launch
// behind the finalize kernel, every rank owns a block of *slots* — one per (parity, rank) — in
// memory all ranks can store to (uncached device memory opened over IPC / peer access for the
// device-side form, shared pinned host memory for the host-side form), and the finalize kernel
// itself pushes its sums into its slot of every rank's block, write-through, followed by the
// sequence number of the sweep.  Slots alternate with the parity of the sequence number: a rank
// can only be one sweep ahead of the slowest reader (it needs that reader's flag of the sweep in
// between), so a slot is never overwritten while someone may still read it.
constexpr int kMaxPeers = 8;                    // one node: 8 GPUs
constexpr int kSlotData = 288;                  // >= n*n + n + 1 for n <= 16 (wide models: 273)
constexpr int kSlotFlag = kSlotData;            // sequence word (as unsigned long long)
constexpr int kSlotStatus = kSlotData + 1;      // 0 ok, kStatusPeerTimeout
constexpr int kSlotDoubles = kSlotData + 16;    // 2432 B: flag and status on their own 128-B line
constexpr unsigned long long kStatusPeerTimeout = 1;
__host__ __device__ inline size_t slotBlockDoubles(int num_ranks) { return size_t(2) * num_ranks * kSlotDoubles; }
__host__ __device__ inline size_t slotIndex(unsigned long long sequence, int num_ranks, int rank) {
  return (size_t(sequence & 1) * num_ranks + rank) * kSlotDoubles;
}

// Device-side combine, run by the (single-workgroup) finalize kernel: push this rank's sums into
// slot (parity, rank) of every rank's block, then wait for the other ranks' flags in the own block
// and add the G slots in rank order — every rank forms bit-identical totals.  num_ranks == 0: off.
struct PeerCombine {
  double *blocks[kMaxPeers] = {nullptr};  // [k]: rank k's slot block as THIS device addresses it
  int rank = 0;
  int num_ranks = 0;
  int offset = 0;  // first value of this sweep inside a slot (0: H | b | sum_sq; n*n + n: cost only)
  unsigned long long sequence = 0;
  unsigned long long timeout_ticks = 0;  // of the 100 MHz wall clock; the wait is always bounded
};

// ---- device-resident Levenberg-Marquardt (mopt_lm_minimize) ---------------------------------
// The LM loop of src/levenberg_marquadt_dyn.cpp:34-119 with the iteration taken on the device:
// after every sweep a one-workgroup step kernel reads the sums, solves the damped 6x6 system,
// tests the trial point and writes the per-x constants of the NEXT sweep (transforms, forward-
// difference steps, affine basis) into HBM, from where the "resident" forms of the sweep and
// finalize kernels read them — so sweeps can be queued ahead of time and the host only sees the
// final x.  A queued kernel that finds `done` set returns at once.
constexpr int kLmMaxCosts = 4;
enum LmModel : int { kLmPoint2Point = 1, kLmReprojection = 2, kLmScalar = 3, kLmJit = 4 };
// values of moptimizer::OptimizationStatus (include/moptimizer/types.h:6-12)
enum LmStatus : int {
  kLmConverged = 0,
  kLmMaxIterations = 1,
  kLmSmallDelta = 2,
  kLmNumericError = 3,
  kLmFatalError = 4,
  kLmRunning = -1
};

struct LmControl {  // device memory; written by the step kernel only
  int done;         // 0 while iterating; queued sweeps / finalizes return at once when set
  int trial;        // sweeps evaluated so far = offset of the next peer-combine sequence number
  int pad[2];       // [0]: a peer-combine status (kStatusPeerTimeout) seen by a finalize kernel
};                  // [1]: the next point is a linearization point whose correspondences are to
                    //      be re-searched first (ICP costs: the model's update(x))
// The control block of a minimisation is kLmControlBlocks of these.  [0] is the one above.
// [kLmGateMoments].done is ALL the sweep and the finalize kernel of a point2point cost read that
// differentiates numerically under MOPT_KERNEL_AUTO / _MOMENTS (LmProblem::fd_per_iterate; one word, one
// load): such kernels hold both forward-difference forms, and the step sets the word to kLmGateStopped
// once the loop has stopped, else to kLmGateLiteralForm where the next point has some 0 < |x_j| < 0.08
// (the reference's own cancellation noise is part of its answer there, linearization.h:85-105), else to
// kLmGateMomentsForm — so the sweep the blocking call would choose at that x is the one that runs.
constexpr int kLmControlBlocks = 2;
constexpr int kLmGateMoments = 1;
constexpr int kLmGateMomentsForm = 0, kLmGateLiteralForm = 1, kLmGateStopped = 2;

struct LmCostDesc {
  int model = 0;     // LmModel
  int jac_mode = 0;  // JacMode
  int n_out = 0;
  int moments = 0;   // point2point: the finalize kernel contracts moments with `basis`
  int x_offset = 0;  // scalar / run-time compiled models: byte offset of x[] | h[] inside `args`
  int x_slots = 8;   // length of each of the two (16 in a wide model's argument block)
  void *args = nullptr;           // device: P2PSweepArgs<S> / ReprojSweepArgs / ScalarSweepArgs<S>
  AffineBasis *basis = nullptr;   // device, point2point moments
  const double *result = nullptr; // device: this cost's H | b | sum_sq (summed over the ranks)
  double camera[12] = {0};        // reprojection constants (row-major 3x4, 4x4)
  double frame[16] = {0};
};

struct LmReport {  // mapped host memory (as doubles so that one layout serves float and double)
  double x[kMaxWideParams];
  double cost;        // sum of squares at the returned x
  double lambda;
  double status;      // LmStatus
  double iterations;  // executed outer iterations (Optimizer::getExecutedIterations)
  double trials;      // sweeps evaluated
  double peer_status; // kStatusPeerTimeout when a rank went missing
  double pad[2];
  unsigned long long flag;  // progress word: 2 * (step runs completed) + (1 once the loop has
};                          // stopped: only then is the payload above written, before the word)

struct LmProblem {
  int num_costs = 0;
  int n = 0;
  int max_iterations = 15;    // optimizer.h:19
  int lm_max_iterations = 3;  // levenberg_marquadt_dyn.cpp:9
  int manifold = 0;           // 1 / 2: x (+) delta composed on SE(3), on the left / right, instead of
                              //        added (n = 6 only)
  int rematch = 0;            // 1: some cost re-searches its correspondences in update(x): an accepted
                              //    point is re-linearized after the search instead of adopted
  int merged = 0;             // 1: the partial rows of all costs lie behind one another and the last
                              //    cost's finalize reduces them all: its result is the sum over the costs
  int fd_per_iterate = 0;     // 1: some point2point cost takes its forward differences by moments or
                              //    literally as each evaluated point asks (control[1], control[2])
  LmCostDesc cost[kLmMaxCosts];
  LmControl *control = nullptr;  // device
  void *state = nullptr;         // device, LmState<S> (lm_kernels.hip)
  LmReport *report = nullptr;    // mapped host memory as the device addresses it
};

// `init`: start a minimisation from x0 (control and state reset, constants of the first sweep
// written); otherwise: digest the sweep that has just finished and propose the next point.
template <typename S>
hipError_t launchLmStep(const LmProblem &problem, bool init, const S *x0, hipStream_t stream);
template <typename Args>
hipError_t launchStoreArgs(const Args &value, Args *d_dst, hipStream_t stream);

// A whole minimisation of one small point2point cost (moments sweep, <= solveSmallMaxTiles() tiles,
// 6 parameters) in one launch of one workgroup: sweep_kernels.hip p2pSolveSmallKernel.  The cost's
// resident blocks must hold its data / loss / covariance (residentPrepare); the report goes where
// problem.report says, as from the launch-per-point loop.  fd_cov: the cost's covariance form (kCov*) where
// problem.fd_per_iterate is set — the kernel then also holds the literal forward-difference form for that
// covariance and the step chooses per point —, ignored otherwise.
int solveSmallMaxTiles();
template <typename S>
hipError_t launchP2PSolveSmall(const S *tiles, int num_tiles, const P2PSweepArgs<S> *d_args,
                               const AffineBasis *d_basis, double *result, const LmProblem &problem,
                               const S *x0, int max_points, int fd_cov, hipStream_t stream);

// Resident forms: per-x constants read from HBM (`d_args`, `d_basis`), early exit on control->done,
// peer-combine sequence = peers.sequence + control->trial.
template <typename S>
hipError_t launchP2PMomentsResident(const S *tiles, int num_tiles, const P2PSweepArgs<S> *d_args,
                                    const LmControl *control, int grid, const LaunchSite &site);
template <typename S>
hipError_t launchP2PLiteralResident(const P2PSweepArgs<S> *d_args, const LmControl *control,
                                    int jac_mode, int cov_mode, int grid, const LaunchSite &site);
// several costs' resident sweeps in one launch (their partial rows behind one another)
struct ResidentSweepSet {
  const void *args[kLmMaxCosts] = {};   // device: each cost's resident argument block
  int first_block[kLmMaxCosts + 1] = {};  // workgroups [first_block[k], first_block[k+1]) -> cost k
  int num_costs = 0;
};
hipError_t launchReprojResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                   int cov_mode, const LaunchSite &site);
// the same for point2point costs evaluated literally (all in one Jacobian mode and covariance form)
// and for the built-in scalar models (all of one kind, mode and covariance form)
template <typename S>
hipError_t launchP2PLiteralResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                       int jac_mode, int cov_mode, const LaunchSite &site);
template <typename S>
hipError_t launchForwardDiffResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                        int cov_mode, const LaunchSite &site);
template <typename S>
hipError_t launchScalarModelResidentSet(const ResidentSweepSet &set, const LmControl *control,
                                        int model, int jac_mode, int cov_mode, hipStream_t stream);
hipError_t launchReprojResident(const ReprojSweepArgs *d_args, const LmControl *control,
                                int cov_mode, int grid, const LaunchSite &site);
template <typename S>
hipError_t launchScalarModelResident(const ScalarSweepArgs<S> *d_args, const LmControl *control,
                                     int model, int jac_mode, int cov_mode, int grid,
                                     hipStream_t stream);
// `step` (optional): this is the last cost of the problem — run the LM step in the same launch
// (`scalar_bytes` selects LevenbergMarquadtDynamic<float | double>, `own_index` = this cost's slot)
hipError_t launchFinalizeDenseResident(const double *partials, int grid, int nacc, int n,
                                       double *result, LmControl *control, hipStream_t stream,
                                       const PeerCombine *peers, const LmProblem *step = nullptr,
                                       int own_index = 0, int scalar_bytes = 8);
// A cost whose sweep kind is chosen per evaluated point (control[kLmGateMoments].done != 0: the literal
// sweep ran): rows of moments over `grid_moments` workgroups, or dense rows of `nacc` values over
// `grid_literal`; n = 6.
hipError_t launchFinalizeEitherResident(const double *partials, int grid_moments, int grid_literal,
                                        int nacc, const AffineBasis *d_basis, double *result,
                                        LmControl *control, hipStream_t stream,
                                        const PeerCombine *peers, const LmProblem *step,
                                        int own_index, int scalar_bytes);
hipError_t launchFinalizeMomentsResident(const double *partials, int grid,
                                         const AffineBasis *d_basis, double *result,
                                         LmControl *control, hipStream_t stream,
                                         const PeerCombine *peers, const LmProblem *step = nullptr,
                                         int own_index = 0, int scalar_bytes = 8);

// ---- launches (all asynchronous on `stream`) ------------------------------------------------
template <typename S>
hipError_t launchRelayoutP2P(const S *src_xyz, const S *tgt_xyz, long long count, S *tiles,
                             int num_tiles, hipStream_t stream);
hipError_t launchRelayoutReproj(const double *points_xyzw, const int32_t *pixels_uv,
                                long long count, unsigned char *tiles, int num_tiles,
                                hipStream_t stream);

// literal per-point evaluation; partials: [grid][kAccSym or kAccFull]
template <typename S>
hipError_t launchP2PLinearizeLiteral(const P2PSweepArgs<S> &args, int jac_mode, int cov_mode,
                                     int grid, const LaunchSite &site);
// forward differences evaluated as the reference does (fd_kernels.hip); rows as the literal sweep's
template <typename S>
hipError_t launchForwardDiff(const P2PSweepArgs<S> &args, int cov_mode, int grid,
                             const LaunchSite &site);
// one launch for a cost whose forward-difference sweep is chosen per evaluated point (kLmGateMoments)
template <typename S>
hipError_t launchForwardDiffEitherResident(const S *tiles, int num_tiles, const P2PSweepArgs<S> *d_args,
                                           const LmControl *control, int cov_mode, int grid,
                                           int moments_grid, const LaunchSite &site);
template <typename S>
hipError_t launchForwardDiffResident(const P2PSweepArgs<S> *d_args, const LmControl *control,
                                     int cov_mode, int grid, const LaunchSite &site);
// moment accumulation (analytic modes; numeric via the affine forward-difference basis)
template <typename S>
hipError_t launchP2PMoments(const P2PSweepArgs<S> &args, int grid, const LaunchSite &site);
template <typename S>
hipError_t launchP2PCost(const P2PSweepArgs<S> &args, int grid, const LaunchSite &site);

hipError_t launchReprojLinearize(const ReprojSweepArgs &args, int cov_mode, int grid,
                                 const LaunchSite &site);
hipError_t launchReprojCost(const ReprojSweepArgs &args, int grid, const LaunchSite &site);

// partials[grid][nacc] -> result[n*n + n + 1] (H column-major | b | sum_sq)
// `peers` (optional): add the sums over the ranks inside the same kernel (PeerCombine above)
// `aql` (optional): the queue the sweep before it went to (aql.hpp) — the finalize kernel must follow
// it there; the three kernels are looked up beforehand (aqlFinalizersLoaded), so this cannot fail over
hipError_t launchFinalizeDense(const double *partials, int grid, int nacc, int n, double *result,
                               const HostPublish &pub, hipStream_t stream,
                               const PeerCombine *peers = nullptr,
                               const mopt_detail::AqlSite *aql = nullptr);
hipError_t launchFinalizeMoments(const double *partials, int grid, const AffineBasis &basis,
                                 double *result, const HostPublish &pub, hipStream_t stream,
                                 const PeerCombine *peers = nullptr,
                                 const mopt_detail::AqlSite *aql = nullptr);
hipError_t launchFinalizeCost(const double *partials, int grid, double *result,
                              const HostPublish &pub, hipStream_t stream,
                              const PeerCombine *peers = nullptr,
                              const mopt_detail::AqlSite *aql = nullptr);
// true when the three finalize kernels above can be dispatched directly on `site`'s device
bool aqlFinalizersLoaded(const mopt_detail::AqlSite &site);
// Small parametric models: one launch computes the workgroup partial rows of the linearization
// (or of the cost when cost_only); finish with launchFinalizeDense(n) / launchFinalizeCost.
template <typename S>
hipError_t launchScalarModel(const ScalarSweepArgs<S> &args, int model, bool cost_only,
                             int jac_mode, int cov_mode, int grid, const LaunchSite &site);

// For every source point i: nearest target of T p_i within max_dist -> target planes of slot i
// (NaN marker when there is none).
template <typename S>
hipError_t launchIcpMatch(const IcpMatchArgs<S> &args, hipStream_t stream);
// the same search with the pose read from the cost's resident sweep constants, run only when the LM
// step kernel asked for it (control->pad[1])
template <typename S>
hipError_t launchIcpMatchResident(const IcpMatchArgs<S> &args, const P2PSweepArgs<S> *d_args,
                                  const LmControl *control, hipStream_t stream);
// target planes of the tile layout -> packed xyz (NaN where unmatched); for inspection / tests
template <typename S>
hipError_t launchGatherTargets(const S *tiles, long long count, const int *order /* or NULL */,
                               S *out_xyz, hipStream_t stream);

// Grid construction for the search (icp_grid.hip).  All pointers are device memory.
//   icpBoundingBox   min / max of m packed points (synchronises the stream)
//   icpSortByCell    d_perm[k] = original index of the k-th point in (cell id, original index)
//                    order; d_cell_start (optional) [cells + 1] offsets into that order
//                    (synchronises the stream)
//   icpGatherPoints  out[k] = xyz[d_perm[k]], packed (3) or padded to 4 scalars
template <typename S>
hipError_t icpBoundingBox(const S *d_xyz, long long m, double lo[3], double hi[3],
                          hipStream_t stream);
template <typename S>
hipError_t icpSortByCell(const S *d_xyz, long long m, const double origin[3], double cell,
                         const int dims[3], int *d_perm, int *d_cell_start, hipStream_t stream,
                         long long *num_parked = nullptr);  // non-NULL: points with a non-finite
                                                             // coordinate sort last and are counted
// cells of the grid that hold at least one point (synchronises the stream)
hipError_t icpCountOccupiedCells(const int *d_cell_start, long long ncells, long long *occupied,
                                 hipStream_t stream);
template <typename S>
hipError_t icpGatherPoints(const S *d_xyz, const int *d_perm, long long m, S *d_out, bool padded,
                           hipStream_t stream);

// sum of the search's per-wave matched counts -> mapped host memory (as one double) + flag
hipError_t launchPublishCounter(const unsigned int *d_wave_counts, long long num_waves,
                                const HostPublish &pub, hipStream_t stream);
// device result (count doubles) -> mapped host memory + flag (after a collective)
hipError_t launchPublish(const double *d_values, int count, const HostPublish &pub,
                         hipStream_t stream);

}  // namespace mopt
