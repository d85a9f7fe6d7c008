/*
 * moptimizer_hip.h — C ABI of the MI355X linearization library (libmoptimizer_hip.so).
 *
 * This is the drop-in boundary of the path: one `mopt_cost` object stands for one
 * moptimizer cost function (a model bound to its data arrays + a loss + a covariance) whose
 * per-residual sweeps run as HIP kernels on one gfx950 device.  Each entry point replaces one
 * piece of the reference's C++ interface (paths relative to /root/reference):
 *
 *   mopt_point2point_create      the `Point2Point` model object bound to its two clouds plus
 *                                the cost-class constructor      tst/point2point.cpp:24-29,82-83;
 *                                include/moptimizer/cost_function_analytical_dyn.h:17-18
 *   mopt_reprojection_create     `CameraModel` + its cost           tst/camera_calibration.cpp:12-31,91-92
 *   mopt_cost_set_covariance     CostFunctionBase::setCovariance    include/moptimizer/cost_function.h:38-40
 *   mopt_cost_set_loss           CostFunctionBase::setLossFunction  include/moptimizer/cost_function.h:37
 *   mopt_cost_linearize          CostFunctionBase::linearize        include/moptimizer/cost_function.h:50
 *                                (= CostComputation::computeHessian / computeHessianNumerical,
 *                                include/moptimizer/linearization.h:65-158)
 *   mopt_cost_compute            CostFunctionBase::computeCost      include/moptimizer/cost_function.h:49
 *                                (= CostComputation::parallelComputeCost, linearization.h:49-63)
 *   mopt_cost_destroy            ~CostFunctionBase                  include/moptimizer/cost_function.h:35
 *
 * The *_async forms and mopt_group_* are additions for sharded (multi-GPU) use: they leave the
 * n*n + n + 1 partial sums on the device so that a collective (RCCL all-reduce) can combine the
 * shards before anything crosses PCIe.
 *
 * Conventions (identical to the reference's buffers):
 *   - scalar_bytes is 4 (float) or 8 (double); `x`, `cov`, `hessian`, `b`, `sum_sq` of a call are
 *     in that scalar.  x has n = 6 entries (tx, ty, tz, wx, wy, wz).
 *   - hessian: n*n, column-major, fully overwritten.  b: n.  sum_sq: the UNWEIGHTED sum of
 *     squared residuals (linearization.h:152,157), also what mopt_cost_compute returns.
 *   - cov: m*m column-major (m = 3 point2point, m = 2 reprojection); NULL = identity.
 *   - device result layout of the async forms: double[n*n + n + 1] = H (column-major) | b | sum_sq,
 *     always fp64 whatever the cost's scalar.
 *   - a mopt_cost (or mopt_group) is used by one thread at a time and runs one sweep at a time
 *     (it owns one set of partial-sum buffers); different costs may be used concurrently.
 *   - every function returns MOPT_OK (0) or an error code; mopt_last_error() gives the text for
 *     the calling thread.  No call falls back to a CPU implementation: without a usable HIP
 *     device the create functions fail with MOPT_ERR_NO_DEVICE / MOPT_ERR_HIP.
 */
#ifndef MOPTIMIZER_HIP_H_
#define MOPTIMIZER_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MOPT_API __attribute__((visibility("default")))
#else
#define MOPT_API
#endif

typedef struct mopt_cost mopt_cost;
typedef struct mopt_group mopt_group;

enum mopt_status {
  MOPT_OK = 0,
  MOPT_ERR_INVALID_ARGUMENT = 1,
  MOPT_ERR_HIP = 2,
  MOPT_ERR_NO_DEVICE = 3,
  MOPT_ERR_RCCL = 4,
  MOPT_ERR_UNSUPPORTED = 5,
  MOPT_ERR_PEER_TIMEOUT = 6 /* a rank of a sharded cost did not deliver its sums in time */
};

/* How the per-residual Jacobian is obtained. */
enum mopt_jacobian_mode {
  /* model-supplied Jacobian, row-major m x n as the API contract states (model.h:35-42):
   * point2point J = [ I3 | -skew(p) ]. */
  MOPT_JAC_ANALYTIC = 0,
  /* bit-faithful to tst/point2point.cpp:71-75, whose f_df fills the buffer column-major while
   * linearization.h:17-18 reads it row-major. */
  MOPT_JAC_ANALYTIC_TST_LAYOUT = 1,
  /* forward differences with the reference's step rule (linearization.h:78-105). */
  MOPT_JAC_NUMERIC = 2,
  /* point2point, for the SE(3) manifold update the reference leaves as "TODO Manifold operation"
   * (src/levenberg_marquadt_dyn.cpp:82-83; include/moptimizer/manifold.h): the derivative of the
   * residual with respect to a LEFT perturbation of the pose, T <- [Exp(phi) R | Exp(phi) t + rho]:
   * J = [ I3 | -skew(R p + t) ], row-major.  Exact at every pose (the Euclidean-parameter form
   * [ I3 | -skew(p) ] is the derivative at R = I only); goes with mopt_lm_options.manifold = 1 /
   * mopt_se3_plus, which apply the step as that perturbation instead of adding it to x. */
  MOPT_JAC_ANALYTIC_LEFT = 3,
  /* the same for a RIGHT perturbation of the rotation with the translation added,
   * T <- [R Exp(phi) | t + rho] — the composition the reference's own sketches of a rotation update
   * use (`parameter_matrix * Exp(delta)`, tst/manifold.cpp:47; `rot_ * rhs_rot`, lin_ += delta,
   * tst/state_model.cpp:28-34): J = [ I3 | -R skew(p) ], row-major; goes with
   * mopt_lm_options.manifold = 2 / mopt_se3_plus_right. */
  MOPT_JAC_ANALYTIC_RIGHT = 4
};

enum mopt_loss_kind {
  MOPT_LOSS_NONE = 0,          /* loss::NoLoss          w = 1               */
  MOPT_LOSS_GEMAN_MCCLURE = 1  /* loss::GemmanMCClure   w = t^2 / (s + t)^2 */
};

enum mopt_create_flags {
  MOPT_INPUT_HOST = 0,   /* data pointers are host memory (copied once to HBM)             */
  MOPT_INPUT_DEVICE = 1  /* data pointers are already device memory on `device` (re-laid out
                            on the GPU, no PCIe traffic)                                     */
};

/* How a sweep evaluates the reference's per-residual arithmetic. */
enum mopt_kernel_variant {
  MOPT_KERNEL_AUTO = 0,    /* fastest variant that meets the parity bar (1e-6 on H, b, cost) for
                              the mode: moments, except forward differences at an x with some
                              0 < |x_j| < 0.08 (literal) — in the blocking and asynchronous
                              calls AND, since round 6, at every point the device-resident loop
                              (mopt_lm_minimize) evaluates: its sweep and finalize kernels hold
                              both forward-difference forms and the step kernel, which forms
                              the next x, names the one the rule asks for there
                              (mopt_cost_lm_choice_stats counts them); so does the one-launch
                              solve of small problems                                          */
  MOPT_KERNEL_LITERAL = 1, /* every residual and Jacobian entry formed per point, then
                              w * J^T * S * J accumulated entry by entry, as the reference does */
  MOPT_KERNEL_MOMENTS = 2, /* Jacobians that are affine in the source point (all point2point
                              modes) reduced through weighted point/residual moments — except
                              for forward differences with some 0 < |x_j| < 0.08, where that
                              evaluation leaves the 1e-6 bar (it lacks the reference's own
                              per-point cancellation noise eps |R p + t| / h_j) and the literal
                              evaluation is used: in every call and at every point of the
                              device-resident loop, as under AUTO                              */
  MOPT_KERNEL_MOMENTS_ALWAYS = 3 /* moments whatever the step size: for measurements; forward
                              differences then differ from the reference's by up to
                              2e-8 / min |x_j| relative (0.97 at |x_j| ~ 1e-8)                  */
};

MOPT_API int mopt_device_count(int *count);

/* x = (tx, ty, tz, wx, wy, wz) -> 4x4 column-major [ Exp(w) t ; 0 1 ]: exactly the transform the
 * sweeps of this library derive from x on the host, once per parameter vector, as the reference's
 * model->setup(x) does (so3::convert6DOFParameterToMatrix + so3::Exp, src/so3.cpp:7-19,43-57).
 * When T_plus_out / h_out are given, also the six forward-difference points of
 * linearization.h:78-92: h_j = sqrt(eps) |x_j| (sqrt(eps) when zero), T_plus_out[16 j ..] the
 * transform at x + h_j e_j.  Needs no device.  scalar_bytes 4 or 8 selects the type of all arrays. */
MOPT_API int mopt_se3_from_params(int scalar_bytes, const void *x, void *T_out /* 16 */,
                                  void *T_plus_out /* 6 * 16 or NULL */, void *h_out /* 6 or NULL */);
/* x (+) delta on SE(3), the manifold form of the reference's Euclidean `xi_ = x0_map_ + delta_`
 * (levenberg_marquadt_dyn.cpp:82-83): with x = (t, w), delta = (rho, phi):
 * R' = Exp(phi) Exp(w), t' = Exp(phi) t + rho, x' = (t', Log(R')) — Exp / Log of src/so3.cpp:43-57,
 * :96-105.  6 scalars each; needs no device. */
MOPT_API int mopt_se3_plus(int scalar_bytes, const void *x, const void *delta, void *x_out);
/* The same composed on the right: R' = Exp(w) Exp(phi), t' = t + rho (tst/manifold.cpp:47,
 * tst/state_model.cpp:28-34). */
MOPT_API int mopt_se3_plus_right(int scalar_bytes, const void *x, const void *delta, void *x_out);
MOPT_API const char *mopt_last_error(void);
MOPT_API const char *mopt_version(void);

/* ---- cost objects ------------------------------------------------------------------------ */

/* Point-to-point ICP cost over `count` index-aligned correspondences.  src_xyz / tgt_xyz: packed
 * xyz triples (24 B per point in fp64 — the memory of a std::vector<Eigen::Vector3d>). */
MOPT_API int mopt_point2point_create(mopt_cost **out, int device, int scalar_bytes,
                                     const void *src_xyz, const void *tgt_xyz, int64_t count,
                                     unsigned flags);

/* Replace the correspondences of an existing point2point cost (same scalar type; any count) without
 * rebuilding it — the service a model's `update(x)` needs when it re-matches correspondences every
 * outer iteration (model.h:24-26, cost_function.h:42-44; the hook is empty in the reference's own
 * models).  Re-lays the new arrays into the resident tiles, growing them if needed. */
MOPT_API int mopt_point2point_set_data(mopt_cost *cost, const void *src_xyz, const void *tgt_xyz,
                                       int64_t count, unsigned flags);

/* ---- ICP with correspondence search on the GPU (the model's update(x) step) ------------------
 * The reference calls cost->update(x) -> model->update(x) at the top of every outer LM iteration
 * (src/levenberg_marquadt_dyn.cpp:54, include/moptimizer/cost_function.h:42-44) "i.e registration
 * correspondences" (model.h:24-26), but ships no model that implements it.  Semantics here: for each
 * source point p, the target nearest to R(x) p + t(x) in Euclidean distance, if within max_distance;
 * sources without one are skipped by the sweeps exactly as an index whose f() returns false
 * (linearization.h:102,144).  mopt_icp_create builds a point2point cost over `num_src` sources
 * whose targets are (re)chosen from the `num_tgt`-point target cloud by every mopt_icp_update; all
 * cost calls (linearize / compute / covariance / loss) apply unchanged.  Host arrays, packed xyz.
 * Points with a NaN or infinite coordinate (the invalid pixels of a depth image) are left out: such
 * a target does not shape the grid and is never matched; such a source is not part of the cost (no
 * residual, no sum sees it) and mopt_icp_get_matches reports a NaN triple for it. */
MOPT_API int mopt_icp_create(mopt_cost **out, int device, int scalar_bytes, const void *src_xyz,
                             int64_t num_src, const void *tgt_xyz, int64_t num_tgt,
                             double max_distance);
/* The same with creation flags: MOPT_INPUT_DEVICE takes both clouds from device memory on `device`
 * (complete when the call is made: synchronise the stream that produced them) — a registration
 * pipeline that keeps its scans on the GPU builds a cost without PCIe traffic. */
MOPT_API int mopt_icp_create_from(mopt_cost **out, int device, int scalar_bytes, const void *src_xyz,
                                  int64_t num_src, const void *tgt_xyz, int64_t num_tgt,
                                  double max_distance, unsigned flags /* mopt_create_flags */);
MOPT_API int mopt_icp_update(mopt_cost *cost, const void *x, int64_t *num_matched /* may be NULL */);
/* current target of every source as packed xyz (NaN triple where unmatched), in the order of the
 * array handed to mopt_icp_create; host buffer of num_src * 3 scalars.  Works for any point2point
 * cost. */
MOPT_API int mopt_icp_get_matches(mopt_cost *cost, void *tgt_out_xyz);
/* The uniform grid an ICP cost searches: cell edge, cells to the search radius (`reach`: 1 where the
 * radius holds about one target, up to 8 where it holds many — the cells are then finer than the
 * radius and the search covers (2 reach + 1)^3 of them), cells per axis and the grid's origin.
 * Ties in distance go to the target stored first: smallest (cell z, cell y, cell x, index in the
 * caller's array) with cell = floor((q - origin) / cell_edge).  Any pointer may be NULL. */
MOPT_API int mopt_icp_grid(const mopt_cost *cost, double *cell_edge, int *reach, int dims[3],
                           double origin[3]);

/* Reprojection (camera-calibration) cost, fp64, numeric Jacobian only.  points_xyzw: packed
 * 4-vectors (32 B); pixels_uv: packed int32 pairs (8 B).  camera_3x4 / frame_4x4: row-major
 * constants (tst/camera_calibration.cpp:22-30); NULL selects the reference's values. */
MOPT_API int mopt_reprojection_create(mopt_cost **out, int device, const double *points_xyzw,
                                      const int32_t *pixels_uv, int64_t count,
                                      const double *camera_3x4, const double *frame_4x4,
                                      unsigned flags);

/* The other parametric models the reference's tests drive through the same cost classes (n != 6):
 *   MOPT_MODEL_EXP_CURVE  y - exp(x0 t + x1)       n = 2, m = 1   tst/curve_fitting.cpp:81-98
 *   MOPT_MODEL_RATIONAL   y - x0 t / (x1 + t)      n = 2, m = 1   tst/test_models.h:7-20 (Jacobian:
 *                                                                 tst/differentiation.cpp:26-38)
 *   MOPT_MODEL_POWELL     Powell's function        n = 4, m = 4   tst/powell.cpp:21-60 (count = 1)
 * t / y: arrays of `count` scalars read with a stride of `stride_scalars` (the curve-fitting data are
 * interleaved pairs: t = data, y = data + 1, stride 2); ignored for MOPT_MODEL_POWELL.  All the cost
 * calls above apply; x, hessian, b have n, n*n, n entries; async results n*n + n + 1 doubles.
 * An observation whose y is NaN is not a residual: the model's f / f_df return false for it
 * (model.h:32,43) and every sweep skips the index (linearization.h:102,144) — the marker a
 * point2point slot without a correspondence carries. */
enum mopt_scalar_model { MOPT_MODEL_EXP_CURVE = 1, MOPT_MODEL_RATIONAL = 2, MOPT_MODEL_POWELL = 3 };
MOPT_API int mopt_scalar_model_create(mopt_cost **out, int device, int scalar_bytes, int model_kind,
                                      const void *t, const void *y, int64_t stride_scalars,
                                      int64_t count);

/* A user-defined model: the device counterpart of subclassing IBaseModel (model.h:11-47).  The
 * reference calls the user's setup(x) once per parameter vector and f(x, f_x, index) /
 * f_df(x, f_x, jacobian, index) per index through virtuals; a GPU sweep cannot, so the model is
 * handed over as the *bodies* of those functions in HIP C++ and compiled for gfx950 at run time
 * (hipRTC) into the per-element sweep:
 *
 *   setup_body     statements of   void setup(const S *x, S *a)
 *                  fills the n_aux per-x values `a` (e.g. x -> rotation matrix); runs once per
 *                  sweep for x and, for forward differences, once per x + h_j e_j - the
 *                  reference sets up one model clone per perturbed vector (linearization.h:91-95).
 *                  NULL with n_aux = 0 when the model needs none
 *   residual_body  statements of   void residual(const S *x, const S *a, const S *d, S *r)
 *                  x = the n parameters, a = setup's output for this x, d = the element's n_planes
 *                  data values, r = m outputs
 *   jacobian_body  statements of   void jacobian(const S *x, const S *a, const S *d, S *J)
 *                  J = m x n row-major (J[i*n + j] = d r_i / d x_j), as IBaseModel::f_df fills it;
 *                  NULL or "" -> only MOPT_JAC_NUMERIC is available (BaseModel without f_df,
 *                  model.h:29-33: the reference throws on f_df, so does mopt_cost_linearize)
 *
 * f and f_df return bool — false: "this index is not a residual", and the loops skip it
 * (model.h:32,43; linearization.h:102,144).  A body says so by clearing the reserved local
 * `bool valid` (true on entry), e.g. "valid = d[2] > 0;": the element then enters no sum, whatever
 * the body leaves in r or J (NaN included).  With a supplied Jacobian the element is skipped when
 * either body clears it (f_df's one verdict); with forward differences only the residual at x
 * decides — the perturbed evaluations' verdicts are ignored, as linearization.h:104 ignores them.
 *
 * S is `double` or `float` per scalar_bytes; 1 <= n <= 16, 1 <= m <= 16, 0 <= n_planes <= 16,
 * 0 <= n_aux <= 64.  Models with n <= 8 and m <= 4 get the per-lane sweep and everything the
 * built-in models have; wider ones (tst/state_model.cpp: n = m = 15) get a sweep that spreads an
 * element's Jacobian columns over 16 lanes — every call of the narrow ones works for them too
 * (blocking, async, mopt_lm_minimize, all three shard combines); only costs of one
 * mopt_lm_minimize problem must share n.  data: n_planes arrays of `count` scalars, plane p at data + p * plane_stride
 * (host memory, or device memory with MOPT_INPUT_DEVICE); copied once.  Everything else (loss,
 * covariance, numeric differentiation with the reference's step, the returned unweighted cost,
 * speculation, async calls, communicators) is that of the built-in models.  A body that does not
 * compile returns MOPT_ERR_INVALID_ARGUMENT with the compiler log in mopt_last_error().  With
 * MOPT_JIT_DUMP_DIR=<dir> in the environment every compiled code object is also written to
 * <dir>/mopt_jit_n<n>_m<m>_s<bytes>_mode<k>_cov<c>[_wide].co (for llvm-objdump / llvm-readelf
 * --notes: the registers, spills and scratch a model's sweep came out with). */
MOPT_API int mopt_jit_model_create(mopt_cost **out, int device, int scalar_bytes, int n_params,
                                   int n_outputs, int n_planes, int n_aux, const char *setup_body,
                                   const char *residual_body, const char *jacobian_body,
                                   const void *data, int64_t plane_stride, int64_t count,
                                   unsigned flags);

MOPT_API int mopt_cost_destroy(mopt_cost *cost);

MOPT_API int mopt_cost_set_covariance(mopt_cost *cost, const void *cov_colmajor);
MOPT_API int mopt_cost_set_loss(mopt_cost *cost, int loss_kind, double parameter);
MOPT_API int mopt_cost_set_kernel_variant(mopt_cost *cost, int variant);

/* count, n (parameters), m (outputs per residual), scalar_bytes, device; any pointer may be NULL */
MOPT_API int mopt_cost_info(const mopt_cost *cost, int64_t *count, int *n, int *m,
                            int *scalar_bytes, int *device);

/* ---- blocking sweeps (what LevenbergMarquadtDynamic::minimize calls) ----------------------- */

MOPT_API int mopt_cost_linearize(mopt_cost *cost, int jacobian_mode, const void *x, void *hessian,
                                 void *b, void *sum_sq);
MOPT_API int mopt_cost_compute(mopt_cost *cost, const void *x, void *sum_sq);

/* Speculation (on by default).  The LM loop evaluates computeCost(xi) for a trial point and, when the
 * step is accepted, linearize(xi) at the very same point in the next outer iteration
 * (src/levenberg_marquadt_dyn.cpp:86,112 then :55).  The linearization sweep reads the same bytes as
 * the cost sweep and also yields sum_sq, so with speculation mopt_cost_compute runs the
 * linearization sweep (in the mode of the most recent mopt_cost_linearize) and keeps H | b | sum_sq;
 * a following mopt_cost_linearize with bit-identical x, mode, loss and covariance returns them
 * without touching HBM — one sweep per accepted LM iteration instead of two.  Values equal those of
 * the un-speculated calls up to the summation order of sum_sq (1e-15 relative).
 * That trade is free for the point2point moments sweep over fixed correspondences.  For the other
 * sweeps (forward differences of the reprojection / scalar / user models, literal evaluation, ICP
 * costs whose update(x) re-matches before the kept result can be used) the library keeps
 * speculating only while it pays: once more kept results have gone unused than used,
 * mopt_cost_compute goes back to the cost-only sweep. */
MOPT_API int mopt_cost_set_speculation(mopt_cost *cost, int enabled);
/* sweeps launched and calls answered from the kept result since creation */
MOPT_API int mopt_cost_stats(const mopt_cost *cost, int64_t *sweeps, int64_t *cache_hits);
/* How many of those sweeps the library dispatched itself — AQL packets with agent-scope fences written
 * into an HSA queue of its own instead of a launch on the cost's HIP stream (blocking sweeps of
 * point2point, reprojection and built-in scalar-model costs; csrc/aql.hpp says why: 1-3 us per call).  Same kernels, same
 * numbers.  MOPT_AQL=0 in the environment keeps everything on HIP streams. */
MOPT_API int mopt_cost_direct_dispatches(const mopt_cost *cost, int64_t *sweeps);
/* Gives back the HSA queues the direct path holds on `device` (each is created when a cost first
 * takes that path — costs with a correspondence search, run-time compiled models and RCCL-combined
 * shards never do — and otherwise kept for the life of the process; the next such cost creates one
 * again, a few milliseconds).  Each is a hardware queue of the GPU, and processes that share one GPU can run out
 * of them — a parent that has finished its own costs calls this before it starts workers on the
 * same GPU.  MOPT_ERR_INVALID_ARGUMENT while a cost of this process lives on the device. */
MOPT_API int mopt_device_trim(int device);

/* Costs of one problem.  The optimizer asks the costs it holds one after the other at the same x:
 * `for (cost : costs_) { cost->update(x0); y0 += cost->linearize(x0, H, b); ... }` and the same for
 * the trial cost (src/levenberg_marquadt_dyn.cpp:52-59, :86) — with blocking calls that is one launch
 * path per cost, back to back.  After mopt_costs_link(costs, n) the first of them asked at some x also
 * queues the sweep each of the others is going to be asked for at that x (its linearization in the
 * mode it was last linearized in, or what its mopt_cost_compute would run), each on its own stream;
 * the others' calls find their sweep in flight and only wait for it.  Results are those of the
 * unlinked calls, bit for bit (the same kernels on the same inputs); a cost asked at a different x
 * than was guessed simply runs its sweep, the queued one is discarded.  Linked costs are used from
 * one thread.  Not queued ahead: sharded costs (their sums need every rank's call) and ICP costs
 * (their update(x) must run first).  num_costs <= 1 (or costs == NULL with 0) unlinks.
 * BASELINE config 5 (two reprojection costs): 49.3 -> 32.7 us per linearization of both. */
MOPT_API int mopt_costs_link(mopt_cost *const *costs, int num_costs);
/* blocking calls of this cost that were answered by a sweep a linked cost had queued */
MOPT_API int mopt_cost_link_stats(const mopt_cost *cost, int64_t *answered_ahead);

/* ---- asynchronous sweeps (shard partials stay in HBM) -------------------------------------- */

/* Enqueue on `hip_stream` and return at once.  hip_stream is a hipStream_t passed as a pointer;
 * NULL is HIP's null (legacy default) stream, as everywhere in HIP — mopt_cost_stream gives the
 * cost's own stream when that is wanted.  d_result: device double[n*n + n + 1].  d_sum_sq: device
 * double[1].  The caller orders later work after the sweep by stream order or a synchronisation.
 * Sweeps of ONE cost must be stream-ordered with respect to each other (a cost owns one set of
 * partial-sum buffers): enqueue them on one stream, or order the streams with events.  The library
 * records an event behind a sweep enqueued on a caller's stream, and mopt_cost_destroy /
 * mopt_point2point_set_data wait for it before the cost's device buffers are reused. */
MOPT_API int mopt_cost_linearize_async(mopt_cost *cost, int jacobian_mode, const void *x,
                                       double *d_result, void *hip_stream);
MOPT_API int mopt_cost_compute_async(mopt_cost *cost, const void *x, double *d_sum_sq,
                                     void *hip_stream);
/* The cost's own stream (hipStream_t) and a wait for it. */
MOPT_API int mopt_cost_stream(mopt_cost *cost, void **hip_stream);
MOPT_API int mopt_cost_synchronize(mopt_cost *cost);

/* ---- multi-process shard group: one rank (process) per GPU, RCCL over xGMI ----------------- */

/* Rank 0 obtains an id (MOPT_COMM_ID_BYTES bytes) and hands it to every rank by any means
 * (torch.distributed broadcast, MPI, a file).  After mopt_cost_comm_init_rank, the blocking
 * mopt_cost_linearize / mopt_cost_compute of this cost sum the n*n + n + 1 (or 1) partial results
 * of all ranks with one ncclAllReduce(sum, fp64) on the cost's stream before publishing them to
 * the host, so every rank returns the whole-data-set H, b, sum_sq.  Collective: all ranks must
 * make the same calls in the same order.  Replaces the host accumulation
 * `hessian_ += cost_hessian_` (src/levenberg_marquadt_dyn.cpp:57-59) across shards. */
#define MOPT_COMM_ID_BYTES 128
MOPT_API int mopt_comm_unique_id(void *id_out, int id_bytes);
MOPT_API int mopt_cost_comm_init_rank(mopt_cost *cost, const void *id, int rank, int num_ranks);
/* What the attached communicator itself reports (ncclCommCount / ncclCommUserRank): the number of
 * ranks the all-reduce of this cost spans and this rank's place among them; 0 / -1 when no
 * communicator is attached.  Either pointer may be NULL. */
MOPT_API int mopt_cost_comm_info(const mopt_cost *cost, int *num_ranks, int *rank);

/* ---- latency-optimised shard combine (replaces the all-reduce launch) -----------------------
 *
 * The message of the all-reduce is 344 bytes, so its cost is launches and hops, not bandwidth
 * (an ncclAllReduce adds a kernel launch behind the finalize kernel, and a publish kernel behind
 * that).  Two fused forms; in both the finalize kernel of the sweep itself hands the sums over and
 * every rank adds the num_ranks contributions in rank order (bit-identical totals on every rank):
 *
 *   MOPT_COMBINE_HOST  the finalize kernel of every rank publishes its sums (write-through stores
 *                      + sequence word) into its slot of one block of pinned host memory shared by
 *                      all ranks (POSIX shared memory registered with HIP); each rank's host
 *                      thread waits for the num_ranks sequence words and adds the slots.  One hop
 *                      (GPU -> host memory), no collective launch, no device-side wait.  Results
 *                      exist on the host only: for the blocking calls.
 *   MOPT_COMBINE_PEER  every rank owns a slot block in uncached device memory which all ranks
 *                      open (hipIpcMemHandle between processes: xGMI stores); the finalize kernel
 *                      pushes its sums into its slot of every rank's block, waits (bounded) for the
 *                      other ranks' sequence words in its own block, adds, and publishes.  The sums
 *                      of all ranks are then also in this rank's HBM: for the *_async calls and
 *                      mopt_lm_minimize, where the next step is taken on the device.
 *   MOPT_COMBINE_RCCL  ncclAllReduce on the cost's stream (mopt_cost_comm_init_rank), kept for
 *                      comparison.
 *
 * Attach (any subset), then pick with mopt_cost_set_combine; the last attached transport is
 * selected by default.  All of this is collective: every rank makes the same calls in the same
 * order, and the caller puts a barrier of its own between attaching and the first sweep, and
 * before destroying (a rank must not push into memory a peer has already released).
 *
 *   mopt_cost_hostcomm_attach   shm_name: a name unique to this job and cost ("/mopt-<pid>-<k>",
 *                               same on every rank); created by whoever comes first, unlinked by
 *                               mopt_hostcomm_unlink (any one rank, after the barrier) or at
 *                               destroy
 *   mopt_cost_peer_export       allocates this rank's slot block and returns its IPC handle
 *                               (MOPT_PEER_HANDLE_BYTES); the caller gathers the handles of all
 *                               ranks, in rank order, by any means
 *   mopt_cost_peer_attach       opens them (handles: num_ranks * MOPT_PEER_HANDLE_BYTES); at most
 *                               8 ranks (one node).  Ranks need not be processes: a handle that
 *                               was exported by the calling process itself (several ranks' costs
 *                               in one process, a thread each) is attached through the block's own
 *                               pointer, with peer access enabled when it lives on another GPU
 *
 * A peer that never arrives ends the device-side wait after MOPT_PEER_TIMEOUT_MS (default 5000)
 * and the call returns MOPT_ERR_PEER_TIMEOUT; the host-side wait is bounded the same way. */
enum mopt_combine_mode {
  MOPT_COMBINE_NONE = 0, /* this rank's own sums (also: detach for a comparison run) */
  MOPT_COMBINE_RCCL = 1,
  MOPT_COMBINE_HOST = 2,
  MOPT_COMBINE_PEER = 3
};
#define MOPT_PEER_HANDLE_BYTES 64
MOPT_API int mopt_cost_hostcomm_attach(mopt_cost *cost, const char *shm_name, int rank,
                                       int num_ranks);
MOPT_API int mopt_hostcomm_unlink(const char *shm_name);
MOPT_API int mopt_cost_peer_export(mopt_cost *cost, int num_ranks, void *handle_out);
MOPT_API int mopt_cost_peer_attach(mopt_cost *cost, const void *handles, int rank, int num_ranks);
MOPT_API int mopt_cost_set_combine(mopt_cost *cost, int combine_mode);
MOPT_API int mopt_cost_get_combine(const mopt_cost *cost, int *combine_mode, int *rank,
                                   int *num_ranks);

/* ---- device-resident Levenberg-Marquardt ----------------------------------------------------
 *
 * An additive entry point next to the boundary above (which stays what the reference's LM loop
 * calls): the whole of LevenbergMarquadtDynamic<Scalar>::minimize
 * (src/levenberg_marquadt_dyn.cpp:34-119) with the iteration taken on the device.  Behind every
 * sweep a one-workgroup kernel adds the costs' H | b | sum_sq (:48-60), solves
 * (H + lambda diag H) delta = -b by pivoted LDL^T (:78-80), forms the trial point (:83), evaluates
 * the gain ratio and the accept / reject / stop logic (:86-115) and writes the transforms and
 * forward-difference constants of the next sweep into HBM; sweeps are queued ahead and read their
 * constants from there.  The host takes no decision and sees only the final x.  Each point is
 * evaluated once, with the linearization sweep (it also yields sum r^T r).
 * A single point2point cost of at most four tiles (2048 fp64 / 4096 fp32 correspondences — the
 * reference's own test sizes) with fixed correspondences is minimised by ONE launch of one workgroup
 * that keeps the correspondences in registers and never leaves the kernel between points (same sums,
 * added in another order; MOPT_LM_ONE_LAUNCH_TILES=0 in the environment keeps the launch-per-point
 * form; `window` does not apply); where the forward-difference form is chosen per point
 * (MOPT_KERNEL_AUTO / _MOMENTS with MOPT_JAC_NUMERIC) that kernel holds both forms.
 *
 * costs / jacobian_modes: the costs of Optimizer::addCost (optimizer.h:58) with the Jacobian mode
 * of each (the cost class the caller would have used), at most 4, same device / scalar type / n;
 * point2point (also ICP costs: their update(x), the correspondence search, then runs inside the
 * loop, at the top of every outer iteration as in :54, and an accepted point is searched and
 * linearized again), reprojection, the built-in scalar models and run-time compiled models.
 * x: n scalars, in: x0, out: the result.
 * For sharded costs select MOPT_COMBINE_PEER first: every rank then runs the identical loop on
 * bit-identical sums (collective call).
 * report->status takes the values of moptimizer::OptimizationStatus (types.h:6-12). */
enum mopt_lm_status {
  MOPT_LM_CONVERGED = 0,
  MOPT_LM_MAXIMUM_ITERATIONS_REACHED = 1,
  MOPT_LM_SMALL_DELTA = 2,
  MOPT_LM_NUMERIC_ERROR = 3,
  MOPT_LM_FATAL_ERROR = 4
};
typedef struct mopt_lm_options {
  int max_iterations;    /* Optimizer::setMaximumIterations, default 15 (optimizer.h:19)        */
  int lm_max_iterations; /* setLevenbergMarquadtIterations, default 3 (levenberg_marquadt_dyn.cpp:9) */
  int manifold;          /* 0: xi = x0 + delta as the reference (:83); 1: xi = x0 (+) delta on SE(3)
                            (mopt_se3_plus; n = 6; use MOPT_JAC_ANALYTIC_LEFT costs); 2: composed
                            on the right (mopt_se3_plus_right; MOPT_JAC_ANALYTIC_RIGHT costs)     */
  int window;            /* trial points queued ahead of the device; 0 = default (3)             */
} mopt_lm_options;
typedef struct mopt_lm_report {
  int status;     /* mopt_lm_status */
  int iterations; /* Optimizer::getExecutedIterations */
  int64_t sweeps; /* points evaluated (one sweep per cost each) */
  double cost;    /* sum of squares at the returned x */
  double lambda;  /* final damping */
} mopt_lm_report;
MOPT_API int mopt_lm_minimize(mopt_cost *const *costs, int num_costs, const int *jacobian_modes,
                              void *x, const mopt_lm_options *options /* NULL = defaults */,
                              mopt_lm_report *report /* may be NULL */);
/* A point2point cost that differentiates numerically under MOPT_KERNEL_AUTO / _MOMENTS has the sweep of
 * every point mopt_lm_minimize evaluates chosen on the device as mopt_cost_linearize would choose it at
 * that x (literal forward differences where some 0 < |x_j| < 0.08, the moments elsewhere).  Since the
 * cost was created: the points so evaluated, and how many of them took the literal sweep. */
MOPT_API int mopt_cost_lm_choice_stats(const mopt_cost *cost, int64_t *points, int64_t *literal_points);

/* ---- measurement -------------------------------------------------------------------------- */

/* With profiling on, sweep kernel launches are bracketed by HIP events recorded on their stream:
 * every launch when `enabled` is 1, every N-th launch when it is N > 1 (recording a pair costs the
 * caller ~5 us, so sampling keeps a timed loop close to its un-instrumented speed).
 * mopt_cost_profile synchronises and reports the accumulated time of the bracketed dominant (sweep)
 * kernels and how many of them there were since profiling was last switched on. */
MOPT_API int mopt_cost_set_profiling(mopt_cost *cost, int enabled);
MOPT_API int mopt_cost_profile(mopt_cost *cost, double *sweep_ms_total, int64_t *sweep_launches);

/* ---- single-process multi-GPU group ------------------------------------------------------- */

/* Shard `count` correspondences contiguously over `num_devices` GPUs of this node — device k
 * holds [k count / G, (k + 1) count / G) — one mopt_cost and one host worker thread per device.
 * Every sweep runs on all devices at once; its n*n + n + 1 partial sums per device are added
 *   - by DEFAULT on the host, in shard order: each device's finalize kernel publishes its sums
 *     into mapped host memory and the calling thread adds the G rows (344 bytes per device: the
 *     exchange is latency, and this is the shortest path — no collective launch);
 *   - with MOPT_GROUP_COLLECTIVE=rccl in the environment when the group is created, by one
 *     ncclAllReduce(sum, fp64) over xGMI on each shard's stream (ncclCommInitAll over `devices`),
 *     the result then read from device 0.  A device list that names a GPU more than once (several
 *     shards on one GPU, to rehearse the sharding on a smaller machine) cannot form an RCCL
 *     communicator and always takes the host sum.
 * Both give the same sums up to the association of G fp64 terms.  The blocking calls below have
 * the semantics of mopt_cost_linearize / mopt_cost_compute on the whole data set.  For one process
 * per GPU (torch.distributed, MPI) use mopt_cost_set_combine on per-rank costs instead. */
MOPT_API int mopt_group_point2point_create(mopt_group **out, const int *devices, int num_devices,
                                           int scalar_bytes, const void *src_xyz,
                                           const void *tgt_xyz, int64_t count);
MOPT_API int mopt_group_destroy(mopt_group *group);
MOPT_API int mopt_group_set_covariance(mopt_group *group, const void *cov_colmajor);
MOPT_API int mopt_group_set_loss(mopt_group *group, int loss_kind, double parameter);
MOPT_API int mopt_group_linearize(mopt_group *group, int jacobian_mode, const void *x,
                                  void *hessian, void *b, void *sum_sq);
MOPT_API int mopt_group_compute(mopt_group *group, const void *x, void *sum_sq);
MOPT_API int mopt_group_size(const mopt_group *group, int *num_devices);

#ifdef __cplusplus
}
#endif
#endif /* MOPTIMIZER_HIP_H_ */
