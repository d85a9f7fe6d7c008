"""ctypes binding of libmoptimizer_hip.so (C ABI: include/moptimizer_hip.h).

Plumbing for tests/ and bench.py only — the host side of the product is C++
(include/moptimizer_amd/*.hpp).  There is no Python or CPU implementation behind these
calls: if the shared library is missing, or no HIP device can run the work, they raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOPT_LIBRARY: another build of the same library (e.g. the -DMOPT_LM_TIMING build of the LM step)
LIB_PATH = os.environ.get("MOPT_LIBRARY") or os.path.join(_HERE, "lib", "libmoptimizer_hip.so")

MOPT_OK = 0
JAC_ANALYTIC, JAC_ANALYTIC_TST_LAYOUT, JAC_NUMERIC, JAC_ANALYTIC_LEFT, JAC_ANALYTIC_RIGHT = 0, 1, 2, 3, 4
LOSS_NONE, LOSS_GEMAN_MCCLURE = 0, 1
INPUT_HOST, INPUT_DEVICE = 0, 1
KERNEL_AUTO, KERNEL_LITERAL, KERNEL_MOMENTS, KERNEL_MOMENTS_ALWAYS = 0, 1, 2, 3
COMBINE_NONE, COMBINE_RCCL, COMBINE_HOST, COMBINE_PEER = 0, 1, 2, 3
COMBINE_NAMES = {COMBINE_NONE: "none", COMBINE_RCCL: "rccl", COMBINE_HOST: "host", COMBINE_PEER: "peer"}
PEER_HANDLE_BYTES = 64
ERR_PEER_TIMEOUT = 6
RESULT_DOUBLES = 43

_lib = None

LM_CONVERGED, LM_MAXIMUM_ITERATIONS_REACHED, LM_SMALL_DELTA, LM_NUMERIC_ERROR, LM_FATAL_ERROR = range(5)


class LmOptions(ctypes.Structure):
    _fields_ = [("max_iterations", ctypes.c_int), ("lm_max_iterations", ctypes.c_int),
                ("manifold", ctypes.c_int), ("window", ctypes.c_int)]


class LmReport(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("iterations", ctypes.c_int), ("sweeps", ctypes.c_int64),
                ("cost", ctypes.c_double), ("lambda_", ctypes.c_double)]


class MoptError(RuntimeError):
    pass


def load():
    """Load the library once.  torch (when present) is imported first so that the process ends
    up with a single HIP runtime: torch wheels bundle libamdhip64.so.7 and the loader shares it
    by soname only if it is already mapped."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MoptError(
            "%s not found: build it with `make` (or __graft_entry__.build()); there is no "
            "fallback implementation" % LIB_PATH)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    c_void_pp = ctypes.POINTER(ctypes.c_void_p)
    sigs = {
        "mopt_device_count": [ctypes.POINTER(ctypes.c_int)],
        "mopt_se3_from_params": [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p],
        "mopt_se3_plus": [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
        "mopt_se3_plus_right": [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
        "mopt_point2point_create": [c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint],
        "mopt_point2point_set_data": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_uint],
        "mopt_icp_create": [c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_double],
        "mopt_icp_create_from": [c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_uint],
        "mopt_icp_update": [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
        "mopt_icp_get_matches": [ctypes.c_void_p, ctypes.c_void_p],
        "mopt_icp_grid": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                          ctypes.POINTER(ctypes.c_double)],
        "mopt_reprojection_create": [c_void_pp, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_uint],
        "mopt_scalar_model_create": [c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_int64],
        "mopt_jit_model_create": [c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p,
                                  ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.c_uint],
        "mopt_cost_destroy": [ctypes.c_void_p],
        "mopt_cost_set_covariance": [ctypes.c_void_p, ctypes.c_void_p],
        "mopt_cost_set_loss": [ctypes.c_void_p, ctypes.c_int, ctypes.c_double],
        "mopt_cost_set_kernel_variant": [ctypes.c_void_p, ctypes.c_int],
        "mopt_cost_info": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                           ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                           ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
        "mopt_cost_linearize": [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_void_p],
        "mopt_cost_compute": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
        "mopt_cost_linearize_async": [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p],
        "mopt_cost_compute_async": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p],
        "mopt_cost_stream": [ctypes.c_void_p, c_void_pp],
        "mopt_cost_synchronize": [ctypes.c_void_p],
        "mopt_cost_set_speculation": [ctypes.c_void_p, ctypes.c_int],
        "mopt_costs_link": [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int],
        "mopt_cost_link_stats": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
        "mopt_cost_direct_dispatches": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
        "mopt_cost_lm_choice_stats": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_int64)],
        "mopt_device_trim": [ctypes.c_int],
        "mopt_cost_stats": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_int64)],
        "mopt_comm_unique_id": [ctypes.c_void_p, ctypes.c_int],
        "mopt_cost_comm_init_rank": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int],
        "mopt_cost_comm_info": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
        "mopt_cost_hostcomm_attach": [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int],
        "mopt_hostcomm_unlink": [ctypes.c_char_p],
        "mopt_cost_peer_export": [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
        "mopt_cost_peer_attach": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int],
        "mopt_cost_set_combine": [ctypes.c_void_p, ctypes.c_int],
        "mopt_cost_get_combine": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
        "mopt_lm_minimize": [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                             ctypes.POINTER(ctypes.c_int), ctypes.c_void_p,
                             ctypes.POINTER(LmOptions), ctypes.POINTER(LmReport)],
        "mopt_cost_set_profiling": [ctypes.c_void_p, ctypes.c_int],
        "mopt_cost_profile": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(ctypes.c_int64)],
        "mopt_group_point2point_create": [c_void_pp, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int64],
        "mopt_group_destroy": [ctypes.c_void_p],
        "mopt_group_set_covariance": [ctypes.c_void_p, ctypes.c_void_p],
        "mopt_group_set_loss": [ctypes.c_void_p, ctypes.c_int, ctypes.c_double],
        "mopt_group_linearize": [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p],
        "mopt_group_compute": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
        "mopt_group_size": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.mopt_last_error.restype = ctypes.c_char_p
    lib.mopt_last_error.argtypes = []
    lib.mopt_version.restype = ctypes.c_char_p
    lib.mopt_version.argtypes = []
    _lib = lib
    return lib


def check(rc):
    if rc != MOPT_OK:
        raise MoptError("moptimizer_hip error %d: %s" % (rc, load().mopt_last_error().decode()))


COMM_ID_BYTES = 128


def hostcomm_unlink(shm_name):
    check(load().mopt_hostcomm_unlink(shm_name.encode()))


def comm_unique_id():
    """RCCL unique id (bytes) for mopt_cost_comm_init_rank; call on one rank and broadcast."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    check(load().mopt_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


def device_trim(device=0):
    """Gives back the hardware queues the library's direct dispatch holds on `device`
    (mopt_device_trim): before workers that share the GPU start.  Raises while a cost of this
    process lives on the device."""
    check(load().mopt_device_trim(int(device)))


def device_count():
    n = ctypes.c_int(0)
    check(load().mopt_device_count(ctypes.byref(n)))
    return n.value


def se3_from_params(x, with_steps=False, dtype=np.float64):
    """The transform(s) the library derives from x on the host (no device needed): 4x4 matrix, and
    with_steps also the six forward-difference transforms and steps."""
    x = np.ascontiguousarray(x, dtype=dtype)
    T = np.zeros(16, dtype=dtype)
    if not with_steps:
        check(load().mopt_se3_from_params(x.itemsize, _ptr(x), _ptr(T), None, None))
        return T.reshape(4, 4, order="F")
    Tp = np.zeros(96, dtype=dtype)
    h = np.zeros(6, dtype=dtype)
    check(load().mopt_se3_from_params(x.itemsize, _ptr(x), _ptr(T), _ptr(Tp), _ptr(h)))
    return (T.reshape(4, 4, order="F"),
            [Tp[16 * j:16 * j + 16].reshape(4, 4, order="F") for j in range(6)], h)


def se3_plus(x, delta, dtype=np.float64):
    """x (+) delta on SE(3) (mopt_se3_plus)."""
    x = np.ascontiguousarray(x, dtype=dtype)
    delta = np.ascontiguousarray(delta, dtype=dtype)
    out = np.zeros(6, dtype=dtype)
    check(load().mopt_se3_plus(x.itemsize, _ptr(x), _ptr(delta), _ptr(out)))
    return out


def se3_plus_right(x, delta, dtype=np.float64):
    """x (+) delta composed on the right: R <- R Exp(delta_w), t <- t + delta_t (mopt_se3_plus_right)."""
    x = np.ascontiguousarray(x, dtype=dtype)
    delta = np.ascontiguousarray(delta, dtype=dtype)
    out = np.zeros(6, dtype=dtype)
    check(load().mopt_se3_plus_right(x.itemsize, _ptr(x), _ptr(delta), _ptr(out)))
    return out


def link_costs(costs):
    """mopt_costs_link: the costs of one problem, asked one after the other at the same x by the
    optimizer's loop — the first one asked queues the others' sweeps too.  [] or one cost unlinks."""
    handles = (ctypes.c_void_p * max(len(costs), 1))(*[c._h for c in costs])
    check(load().mopt_costs_link(handles, len(costs)))


def lm_minimize(costs, jac_modes, x0, max_iterations=15, lm_max_iterations=3, window=0,
                manifold=False):
    """Device-resident LevenbergMarquadtDynamic::minimize over `costs` (mopt_lm_minimize).
    `manifold`: False / 0 Euclidean update (the reference), True / 1 / "left" x (+) delta composed on
    the left, 2 / "right" on the right.  Returns (x, report dict)."""
    manifold = {"left": 1, "right": 2}.get(manifold, manifold)
    costs = list(costs)
    dt = _dtype_of(costs[0].scalar_bytes)
    x = np.array(x0, dtype=dt).copy()
    handles = (ctypes.c_void_p * len(costs))(*[c._h for c in costs])
    modes = (ctypes.c_int * len(costs))(*[int(m) for m in jac_modes])
    opt = LmOptions(int(max_iterations), int(lm_max_iterations), int(manifold), int(window))
    rep = LmReport()
    check(load().mopt_lm_minimize(handles, len(costs), modes, _ptr(x), ctypes.byref(opt),
                                  ctypes.byref(rep)))
    return x, dict(status=rep.status, iterations=rep.iterations, sweeps=rep.sweeps, cost=rep.cost,
                   lambda_=rep.lambda_)


def _dtype_of(scalar_bytes):
    return np.float64 if scalar_bytes == 8 else np.float32


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _CostBase:
    """Mirror of the CostFunctionBase surface over one mopt_cost handle."""

    def __init__(self):
        self._h = ctypes.c_void_p()
        self.scalar_bytes = 8
        self.n_out = 3
        self.n_params = 6

    # -- CostFunctionBase::setCovariance / setLossFunction ---------------------------------
    def set_covariance(self, cov):
        if cov is None:
            check(load().mopt_cost_set_covariance(self._h, None))
            return
        cov = np.asfortranarray(np.asarray(cov, dtype=_dtype_of(self.scalar_bytes)))
        assert cov.shape == (self.n_out, self.n_out)
        check(load().mopt_cost_set_covariance(self._h, _ptr(cov)))

    def set_loss(self, kind, parameter=0.0):
        check(load().mopt_cost_set_loss(self._h, int(kind), float(parameter)))

    def set_kernel_variant(self, variant):
        check(load().mopt_cost_set_kernel_variant(self._h, int(variant)))

    def set_speculation(self, enabled):
        check(load().mopt_cost_set_speculation(self._h, 1 if enabled else 0))

    def answered_ahead(self):
        """blocking calls answered by a sweep a linked cost had queued (mopt_costs_link)"""
        hits = ctypes.c_int64(0)
        check(load().mopt_cost_link_stats(self._h, ctypes.byref(hits)))
        return hits.value

    def direct_dispatches(self):
        """sweeps the library dispatched itself (AQL packets into its own HSA queue) instead of on the HIP stream"""
        n = ctypes.c_int64(0)
        check(load().mopt_cost_direct_dispatches(self._h, ctypes.byref(n)))
        return n.value

    def lm_choice_stats(self):
        """(points of device-resident solves whose forward-difference sweep was chosen per point, how many
        of them took the literal sweep) since creation."""
        points, literal = ctypes.c_int64(0), ctypes.c_int64(0)
        check(load().mopt_cost_lm_choice_stats(self._h, ctypes.byref(points), ctypes.byref(literal)))
        return points.value, literal.value

    def stats(self):
        sweeps, hits = ctypes.c_int64(0), ctypes.c_int64(0)
        check(load().mopt_cost_stats(self._h, ctypes.byref(sweeps), ctypes.byref(hits)))
        return sweeps.value, hits.value

    # -- CostFunctionBase::linearize / computeCost ------------------------------------------
    def linearize(self, x, jac_mode):
        dt = _dtype_of(self.scalar_bytes)
        x = np.ascontiguousarray(x, dtype=dt)
        n = self.n_params
        assert x.shape == (n,)
        H = np.zeros((n, n), dtype=dt, order="F")
        b = np.zeros(n, dtype=dt)
        s = np.zeros(1, dtype=dt)
        check(load().mopt_cost_linearize(self._h, int(jac_mode), _ptr(x), _ptr(H), _ptr(b), _ptr(s)))
        return H, b, s[0]

    def bound_linearize(self, jac_mode):
        """Pre-bound form for tight loops (bench): returns (call, x, H, b, s) where x is a reusable
        input array, H/b/s are reused outputs and call() runs one blocking linearize."""
        dt = _dtype_of(self.scalar_bytes)
        n = self.n_params
        x = np.zeros(n, dtype=dt)
        H = np.zeros((n, n), dtype=dt, order="F")
        b = np.zeros(n, dtype=dt)
        s = np.zeros(1, dtype=dt)
        fn = load().mopt_cost_linearize
        args = (self._h, int(jac_mode), _ptr(x), _ptr(H), _ptr(b), _ptr(s))

        def call():
            rc = fn(*args)
            if rc != MOPT_OK:
                check(rc)

        return call, x, H, b, s

    def compute_cost(self, x):
        dt = _dtype_of(self.scalar_bytes)
        x = np.ascontiguousarray(x, dtype=dt)
        s = np.zeros(1, dtype=dt)
        check(load().mopt_cost_compute(self._h, _ptr(x), _ptr(s)))
        return s[0]

    # -- asynchronous forms: results stay in HBM --------------------------------------------
    def own_stream(self):
        s = ctypes.c_void_p()
        check(load().mopt_cost_stream(self._h, ctypes.byref(s)))
        return s.value or 0

    def linearize_async(self, x, jac_mode, d_result_ptr, stream_ptr=None):
        """stream_ptr: a hipStream_t value (0 = HIP's null stream, torch's default); None = the
        cost's own stream."""
        x = np.ascontiguousarray(x, dtype=_dtype_of(self.scalar_bytes))
        stream = self.own_stream() if stream_ptr is None else stream_ptr
        check(load().mopt_cost_linearize_async(self._h, int(jac_mode), _ptr(x),
                                               ctypes.c_void_p(d_result_ptr),
                                               ctypes.c_void_p(stream)))

    def compute_cost_async(self, x, d_sum_ptr, stream_ptr=None):
        x = np.ascontiguousarray(x, dtype=_dtype_of(self.scalar_bytes))
        stream = self.own_stream() if stream_ptr is None else stream_ptr
        check(load().mopt_cost_compute_async(self._h, _ptr(x), ctypes.c_void_p(d_sum_ptr),
                                             ctypes.c_void_p(stream)))

    def comm_init_rank(self, unique_id, rank, num_ranks):
        """Attach this cost (one shard) to the multi-process RCCL group: afterwards the blocking
        linearize / compute_cost return the sums over all ranks."""
        buf = ctypes.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        check(load().mopt_cost_comm_init_rank(self._h, buf, int(rank), int(num_ranks)))

    # -- shard combine without a collective launch (include/moptimizer_hip.h) ---------------
    def comm_info(self):
        """(ranks, rank) as the attached RCCL communicator reports them (ncclCommCount,
        ncclCommUserRank); (0, -1) without a communicator."""
        n, r = ctypes.c_int(0), ctypes.c_int(-1)
        check(load().mopt_cost_comm_info(self._h, ctypes.byref(n), ctypes.byref(r)))
        return n.value, r.value

    def hostcomm_attach(self, shm_name, rank, num_ranks):
        """MOPT_COMBINE_HOST: every rank's finalize kernel publishes into one shared host block."""
        check(load().mopt_cost_hostcomm_attach(self._h, shm_name.encode(), int(rank),
                                               int(num_ranks)))

    def peer_export(self, num_ranks):
        """Allocate this rank's device slot block; returns its IPC handle (bytes)."""
        buf = ctypes.create_string_buffer(PEER_HANDLE_BYTES)
        check(load().mopt_cost_peer_export(self._h, int(num_ranks), buf))
        return buf.raw

    def peer_attach(self, handles, rank, num_ranks):
        """MOPT_COMBINE_PEER: open the slot blocks of all ranks (handles in rank order)."""
        blob = b"".join(bytes(h) for h in handles)
        assert len(blob) == PEER_HANDLE_BYTES * num_ranks
        buf = ctypes.create_string_buffer(blob, len(blob))
        check(load().mopt_cost_peer_attach(self._h, buf, int(rank), int(num_ranks)))

    def set_combine(self, mode):
        check(load().mopt_cost_set_combine(self._h, int(mode)))

    def get_combine(self):
        mode, rank, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(load().mopt_cost_get_combine(self._h, ctypes.byref(mode), ctypes.byref(rank),
                                           ctypes.byref(n)))
        return mode.value, rank.value, n.value

    def synchronize(self):
        check(load().mopt_cost_synchronize(self._h))

    def set_profiling(self, enabled):
        """False/0 off; True/1 every sweep launch; N > 1 every N-th launch."""
        check(load().mopt_cost_set_profiling(self._h, int(enabled)))

    def profile(self):
        ms = ctypes.c_double(0)
        n = ctypes.c_int64(0)
        check(load().mopt_cost_profile(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def info(self):
        count = ctypes.c_int64()
        n, m, sb, dev = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(load().mopt_cost_info(self._h, ctypes.byref(count), ctypes.byref(n), ctypes.byref(m),
                                    ctypes.byref(sb), ctypes.byref(dev)))
        return dict(count=count.value, n=n.value, m=m.value, scalar_bytes=sb.value,
                    device=dev.value)

    def close(self):
        if self._h:
            load().mopt_cost_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Point2PointCost(_CostBase):
    """Point-to-point ICP cost on one GPU.  src / tgt: [N, 3] arrays (host), or raw device
    pointers with `count` when device_ptrs=True."""

    def __init__(self, src, tgt, device=0, dtype=np.float64, device_ptrs=False, count=None):
        super().__init__()
        self.scalar_bytes = np.dtype(dtype).itemsize
        self.n_out = 3
        if device_ptrs:
            n = int(count)
            check(load().mopt_point2point_create(ctypes.byref(self._h), device, self.scalar_bytes,
                                                 ctypes.c_void_p(int(src)),
                                                 ctypes.c_void_p(int(tgt)), n, INPUT_DEVICE))
        else:
            src = np.ascontiguousarray(src, dtype=dtype).reshape(-1, 3)
            tgt = np.ascontiguousarray(tgt, dtype=dtype).reshape(-1, 3)
            assert src.shape == tgt.shape
            n = src.shape[0]
            check(load().mopt_point2point_create(ctypes.byref(self._h), device, self.scalar_bytes,
                                                 _ptr(src), _ptr(tgt), n, INPUT_HOST))
        self.count = n


    def set_data(self, src, tgt):
        """Replace the correspondences (host arrays) in place."""
        dt = _dtype_of(self.scalar_bytes)
        src = np.ascontiguousarray(src, dtype=dt).reshape(-1, 3)
        tgt = np.ascontiguousarray(tgt, dtype=dt).reshape(-1, 3)
        assert src.shape == tgt.shape
        check(load().mopt_point2point_set_data(self._h, _ptr(src), _ptr(tgt), src.shape[0],
                                               INPUT_HOST))
        self.count = src.shape[0]


class IcpCost(Point2PointCost):
    """Point-to-point cost whose correspondences are re-searched on the GPU: update(x) is the
    model's update step (nearest target of the warped source within max_distance)."""

    def __init__(self, src, tgt, max_distance, device=0, dtype=np.float64):
        """src, tgt: host arrays (n, 3), or — both — contiguous torch tensors on `device`, taken from
        device memory (mopt_icp_create_from with MOPT_INPUT_DEVICE; their stream is synchronised)."""
        _CostBase.__init__(self)
        self.n_out = 3
        if hasattr(src, "data_ptr") and hasattr(tgt, "data_ptr"):
            import torch
            assert src.is_cuda and tgt.is_cuda and src.dtype == tgt.dtype
            if src.dtype not in (torch.float32, torch.float64):
                raise TypeError("IcpCost takes float32 or float64 clouds, not %s" % src.dtype)
            assert src.is_contiguous() and tgt.is_contiguous()
            self.scalar_bytes = src.element_size()
            torch.cuda.synchronize(src.device)
            n_src, n_tgt = src.numel() // 3, tgt.numel() // 3
            check(load().mopt_icp_create_from(ctypes.byref(self._h), src.device.index or 0,
                                              self.scalar_bytes, ctypes.c_void_p(src.data_ptr()), n_src,
                                              ctypes.c_void_p(tgt.data_ptr()), n_tgt, float(max_distance),
                                              1))
            self.count = n_src
            return
        self.scalar_bytes = np.dtype(dtype).itemsize
        src = np.ascontiguousarray(src, dtype=dtype).reshape(-1, 3)
        tgt = np.ascontiguousarray(tgt, dtype=dtype).reshape(-1, 3)
        check(load().mopt_icp_create(ctypes.byref(self._h), device, self.scalar_bytes, _ptr(src),
                                     src.shape[0], _ptr(tgt), tgt.shape[0], float(max_distance)))
        self.count = src.shape[0]

    def update(self, x, count_matches=True):
        x = np.ascontiguousarray(x, dtype=_dtype_of(self.scalar_bytes))
        n = ctypes.c_int64(-1)
        check(load().mopt_icp_update(self._h, _ptr(x), ctypes.byref(n) if count_matches else None))
        return n.value

    def grid(self):
        """(cell edge, reach, dims[3], origin[3]) of the grid the search walks."""
        edge, reach = ctypes.c_double(), ctypes.c_int()
        dims, origin = (ctypes.c_int * 3)(), (ctypes.c_double * 3)()
        check(load().mopt_icp_grid(self._h, ctypes.byref(edge), ctypes.byref(reach), dims, origin))
        return edge.value, reach.value, np.array(dims[:]), np.array(origin[:])

    def matches(self):
        out = np.zeros((self.count, 3), dtype=_dtype_of(self.scalar_bytes))
        check(load().mopt_icp_get_matches(self._h, _ptr(out)))
        return out


class ReprojectionCost(_CostBase):
    """Camera-calibration reprojection cost (fp64, numeric Jacobian) on one GPU."""

    def __init__(self, points_xyzw, pixels_uv, device=0, camera=None, frame=None):
        super().__init__()
        self.scalar_bytes = 8
        self.n_out = 2
        pts = np.ascontiguousarray(points_xyzw, dtype=np.float64).reshape(-1, 4)
        pix = np.ascontiguousarray(pixels_uv, dtype=np.int32).reshape(-1, 2)
        assert pts.shape[0] == pix.shape[0]
        cam = None if camera is None else np.ascontiguousarray(camera, dtype=np.float64)
        frm = None if frame is None else np.ascontiguousarray(frame, dtype=np.float64)
        check(load().mopt_reprojection_create(
            ctypes.byref(self._h), device, _ptr(pts), _ptr(pix), pts.shape[0],
            None if cam is None else _ptr(cam), None if frm is None else _ptr(frm), INPUT_HOST))
        self.count = pts.shape[0]


MODEL_EXP_CURVE, MODEL_RATIONAL, MODEL_POWELL = 1, 2, 3


class ScalarModelCost(_CostBase):
    """One of the reference tests' small parametric models (exp curve, rational, Powell) on the GPU."""

    _SHAPES = {MODEL_EXP_CURVE: (2, 1), MODEL_RATIONAL: (2, 1), MODEL_POWELL: (4, 4)}

    def __init__(self, kind, t=None, y=None, device=0, dtype=np.float64):
        super().__init__()
        self.scalar_bytes = np.dtype(dtype).itemsize
        self.n_params, self.n_out = self._SHAPES[kind]
        if kind == MODEL_POWELL:
            check(load().mopt_scalar_model_create(ctypes.byref(self._h), device, self.scalar_bytes,
                                                  kind, None, None, 1, 1))
            self.count = 1
        else:
            t = np.ascontiguousarray(t, dtype=dtype)
            y = np.ascontiguousarray(y, dtype=dtype)
            assert t.shape == y.shape and t.ndim == 1
            check(load().mopt_scalar_model_create(ctypes.byref(self._h), device, self.scalar_bytes,
                                                  kind, _ptr(t), _ptr(y), 1, t.shape[0]))
            self.count = t.shape[0]


class JitModelCost(_CostBase):
    """A user-defined model given as HIP source for its residual (and optionally setup and
    Jacobian) bodies."""

    def __init__(self, n_params, n_outputs, residual_body, jacobian_body=None, planes=None,
                 device=0, dtype=np.float64, n_aux=0, setup_body=None):
        super().__init__()
        self.scalar_bytes = np.dtype(dtype).itemsize
        self.n_params, self.n_out = int(n_params), int(n_outputs)
        if planes is None:
            data, n_planes, count = None, 0, 1
        else:
            data = np.ascontiguousarray(planes, dtype=dtype)
            assert data.ndim == 2
            n_planes, count = data.shape
        jac = None if not jacobian_body else jacobian_body.encode()
        setup = None if not setup_body else setup_body.encode()
        check(load().mopt_jit_model_create(ctypes.byref(self._h), device, self.scalar_bytes,
                                           self.n_params, self.n_out, n_planes, int(n_aux), setup,
                                           residual_body.encode(), jac,
                                           None if data is None else _ptr(data), count, count, 0))
        self.count = count


class Point2PointGroup:
    """Single-process multi-GPU point-to-point cost (RCCL all-reduce inside the library)."""

    def __init__(self, src, tgt, devices, dtype=np.float64):
        self._h = ctypes.c_void_p()
        self.scalar_bytes = np.dtype(dtype).itemsize
        src = np.ascontiguousarray(src, dtype=dtype).reshape(-1, 3)
        tgt = np.ascontiguousarray(tgt, dtype=dtype).reshape(-1, 3)
        devs = (ctypes.c_int * len(devices))(*devices)
        check(load().mopt_group_point2point_create(ctypes.byref(self._h), devs, len(devices),
                                                   self.scalar_bytes, _ptr(src), _ptr(tgt),
                                                   src.shape[0]))

    def set_covariance(self, cov):
        cov = None if cov is None else np.asfortranarray(
            np.asarray(cov, dtype=_dtype_of(self.scalar_bytes)))
        check(load().mopt_group_set_covariance(self._h, None if cov is None else _ptr(cov)))

    def set_loss(self, kind, parameter=0.0):
        check(load().mopt_group_set_loss(self._h, int(kind), float(parameter)))

    def linearize(self, x, jac_mode):
        dt = _dtype_of(self.scalar_bytes)
        x = np.ascontiguousarray(x, dtype=dt)
        H = np.zeros((6, 6), dtype=dt, order="F")
        b = np.zeros(6, dtype=dt)
        s = np.zeros(1, dtype=dt)
        check(load().mopt_group_linearize(self._h, int(jac_mode), _ptr(x), _ptr(H), _ptr(b),
                                          _ptr(s)))
        return H, b, s[0]

    def bound_linearize(self, jac_mode):
        """Pre-bound form for tight loops (bench): returns (call, x, H, b, s) where x is a reusable
        input array, H/b/s are reused outputs and call() runs one blocking linearize."""
        dt = _dtype_of(self.scalar_bytes)
        x = np.zeros(6, dtype=dt)
        H = np.zeros((6, 6), dtype=dt, order="F")
        b = np.zeros(6, dtype=dt)
        s = np.zeros(1, dtype=dt)
        fn = load().mopt_group_linearize
        args = (self._h, int(jac_mode), _ptr(x), _ptr(H), _ptr(b), _ptr(s))

        def call():
            rc = fn(*args)
            if rc != MOPT_OK:
                check(rc)

        return call, x, H, b, s

    def compute_cost(self, x):
        dt = _dtype_of(self.scalar_bytes)
        x = np.ascontiguousarray(x, dtype=dt)
        s = np.zeros(1, dtype=dt)
        check(load().mopt_group_compute(self._h, _ptr(x), _ptr(s)))
        return s[0]

    def close(self):
        if self._h:
            load().mopt_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
