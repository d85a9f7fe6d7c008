"""ctypes access to oracle/_build/liboracle.so — the CPU restatement of the reference path.
Test infrastructure: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only, never by the product."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
REPLAY = os.path.join(ORACLE_DIR, "_build", "replay_reference_tests")

ANALYTIC_DYN, NUMERIC_DYN, ANALYTIC_STATIC, NUMERIC_STATIC = 0, 1, 2, 3
LAYOUT_ROW_MAJOR, LAYOUT_TST, LAYOUT_LEFT, LAYOUT_RIGHT = 0, 1, 2, 3
MANIFOLD_UPDATE = 4  # OR-ed into `layout` of p2p_minimize: xi = x0 (+) delta on SE(3), left composition
MANIFOLD_UPDATE_RIGHT = 8  # the same composed on the right (R <- R Exp(delta_w), t <- t + delta_t)

_lib = None


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not (os.path.exists(LIB) and os.path.exists(REPLAY)):
            build()
        _lib = Oracle(ctypes.CDLL(LIB))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_p2p_linearize_threads.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_p2p_linearize.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_p2p_cost.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p]
        lib.oracle_p2p_minimize.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_int),
            ctypes.POINTER(ctypes.c_int)]
        lib.oracle_camera_linearize.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_camera_cost.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_camera_minimize.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        lib.oracle_se3_from_x.argtypes = [ctypes.c_void_p] * 4

    def build_flags(self):
        """Compiler and flags the checker was built with (oracle/Makefile)."""
        self.lib.oracle_build_flags.restype = ctypes.c_char_p
        return self.lib.oracle_build_flags().decode()

    @staticmethod
    def _prep(src, tgt, x, cov, dtype):
        src = np.ascontiguousarray(src, dtype=dtype).reshape(-1, 3)
        tgt = np.ascontiguousarray(tgt, dtype=dtype).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=dtype)
        cov = None if cov is None else np.asfortranarray(np.asarray(cov, dtype=dtype))
        return src, tgt, x, cov

    def p2p_linearize(self, src, tgt, x, cost_class=ANALYTIC_DYN, layout=LAYOUT_ROW_MAJOR,
                      cov=None, loss_kind=0, loss_param=0.0, dtype=np.float64, threads=0):
        src, tgt, x, cov = self._prep(src, tgt, x, cov, dtype)
        sb = np.dtype(dtype).itemsize
        H = np.zeros((6, 6), dtype=dtype, order="F")
        b = np.zeros(6, dtype=dtype)
        s = np.zeros(1, dtype=dtype)
        if threads and threads > 0:
            rc = self.lib.oracle_p2p_linearize_threads(
                sb, cost_class, layout, _p(src), _p(tgt), src.shape[0], _p(x), _p(cov), loss_kind,
                loss_param, threads, _p(H), _p(b), _p(s))
        else:
            rc = self.lib.oracle_p2p_linearize(sb, cost_class, layout, _p(src), _p(tgt),
                                               src.shape[0], _p(x), _p(cov), loss_kind,
                                               loss_param, _p(H), _p(b), _p(s))
        assert rc == 0, rc
        return H, b, s[0]

    def p2p_cost(self, src, tgt, x, dtype=np.float64, threads=0):
        src, tgt, x, _ = self._prep(src, tgt, x, None, dtype)
        s = np.zeros(1, dtype=dtype)
        rc = self.lib.oracle_p2p_cost(np.dtype(dtype).itemsize, _p(src), _p(tgt), src.shape[0],
                                      _p(x), threads, _p(s))
        assert rc == 0, rc
        return s[0]

    def p2p_minimize(self, src, tgt, x0, cost_class=NUMERIC_DYN, layout=LAYOUT_TST, max_iter=15,
                     lm_iter=0, cov=None, loss_kind=0, loss_param=0.0, dtype=np.float64):
        src, tgt, x, cov = self._prep(src, tgt, x0, cov, dtype)
        x = x.copy()
        status, iters = ctypes.c_int(), ctypes.c_int()
        rc = self.lib.oracle_p2p_minimize(np.dtype(dtype).itemsize, cost_class, layout, _p(src),
                                          _p(tgt), src.shape[0], _p(x), max_iter, lm_iter, _p(cov),
                                          loss_kind, loss_param, ctypes.byref(status),
                                          ctypes.byref(iters))
        assert rc == 0, rc
        return x, status.value, iters.value

    def camera_linearize(self, pts, pix, x, cov=None, loss_kind=0, loss_param=0.0):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 4)
        pix = np.ascontiguousarray(pix, dtype=np.int32).reshape(-1, 2)
        x = np.ascontiguousarray(x, dtype=np.float64)
        cov = None if cov is None else np.asfortranarray(np.asarray(cov, dtype=np.float64))
        H = np.zeros((6, 6), order="F")
        b = np.zeros(6)
        s = np.zeros(1)
        rc = self.lib.oracle_camera_linearize(_p(pts), _p(pix), pts.shape[0], _p(x), _p(cov),
                                              loss_kind, loss_param, _p(H), _p(b), _p(s))
        assert rc == 0, rc
        return H, b, s[0]

    def camera_cost(self, pts, pix, x):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 4)
        pix = np.ascontiguousarray(pix, dtype=np.int32).reshape(-1, 2)
        x = np.ascontiguousarray(x, dtype=np.float64)
        s = np.zeros(1)
        rc = self.lib.oracle_camera_cost(_p(pts), _p(pix), pts.shape[0], _p(x), _p(s))
        assert rc == 0, rc
        return s[0]

    def camera_minimize(self, pts, pix, counts, x0, max_iter=15, loss_kind=0, loss_param=0.0):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 4)
        pix = np.ascontiguousarray(pix, dtype=np.int32).reshape(-1, 2)
        x = np.ascontiguousarray(x0, dtype=np.float64).copy()
        cnt = (ctypes.c_int * len(counts))(*counts)
        status, iters = ctypes.c_int(), ctypes.c_int()
        rc = self.lib.oracle_camera_minimize(_p(pts), _p(pix), cnt, len(counts), _p(x), max_iter,
                                             loss_kind, loss_param, ctypes.byref(status),
                                             ctypes.byref(iters))
        assert rc == 0, rc
        return x, status.value, iters.value

    def scalar_linearize(self, kind, t, y, x, numeric=True, cov=None, loss_kind=0, loss_param=0.0,
                         dtype=np.float64):
        # kind: 1 exp curve, 2 rational, 3 Powell; 4 / 5: rational / exp curve whose f / f_df return
        # false for an observation with a NaN y
        n, m = {1: (2, 1), 2: (2, 1), 3: (4, 4), 4: (2, 1), 5: (2, 1)}[kind]
        count = 1 if kind == 3 else len(t)
        t = None if t is None else np.ascontiguousarray(t, dtype=dtype)
        y = None if y is None else np.ascontiguousarray(y, dtype=dtype)
        x = np.ascontiguousarray(x, dtype=dtype)
        cov = None if cov is None else np.asfortranarray(np.asarray(cov, dtype=dtype))
        H = np.zeros((n, n), dtype=dtype, order="F")
        b = np.zeros(n, dtype=dtype)
        s = np.zeros(1, dtype=dtype)
        self.lib.oracle_scalar_linearize.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p]
        rc = self.lib.oracle_scalar_linearize(np.dtype(dtype).itemsize, kind, 1 if numeric else 0,
                                              _p(t), _p(y), count, _p(x), _p(cov), loss_kind,
                                              loss_param, _p(H), _p(b), _p(s))
        assert rc == 0, rc
        return H, b, s[0]

    def state_linearize(self, x_init, x, cov=None, loss_kind=0, loss_param=0.0):
        """tst/state_model.cpp: StateModel(x_init), n = m = 15, one residual block, numeric cost class."""
        x_init = np.ascontiguousarray(x_init, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        cov = None if cov is None else np.asfortranarray(np.asarray(cov, dtype=np.float64))
        H = np.zeros((15, 15), order="F")
        b = np.zeros(15)
        s = np.zeros(1)
        self.lib.oracle_state_linearize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p]
        rc = self.lib.oracle_state_linearize(_p(x_init), _p(x), _p(cov), loss_kind, loss_param,
                                             _p(H), _p(b), _p(s))
        assert rc == 0, rc
        return H, b, s[0]

    def state_cost(self, x_init, x):
        x_init = np.ascontiguousarray(x_init, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        s = np.zeros(1)
        self.lib.oracle_state_cost.argtypes = [ctypes.c_void_p] * 3
        assert self.lib.oracle_state_cost(_p(x_init), _p(x), _p(s)) == 0
        return s[0]

    def state_minimize(self, x_init, x0, max_iter=15):
        x_init = np.ascontiguousarray(x_init, dtype=np.float64)
        x = np.array(x0, dtype=np.float64)
        status, iters = ctypes.c_int(0), ctypes.c_int(0)
        self.lib.oracle_state_minimize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_void_p, ctypes.c_void_p]
        rc = self.lib.oracle_state_minimize(_p(x_init), _p(x), max_iter, ctypes.byref(status),
                                            ctypes.byref(iters))
        assert rc == 0, rc
        return x, status.value, iters.value

    def se3_from_x(self, x, with_steps=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        T = np.zeros(16)
        if with_steps:
            Tp = np.zeros(96)
            h = np.zeros(6)
            self.lib.oracle_se3_from_x(_p(x), _p(T), _p(Tp), _p(h))
            return (T.reshape(4, 4, order="F"),
                    [Tp[16 * j:16 * j + 16].reshape(4, 4, order="F") for j in range(6)], h)
        self.lib.oracle_se3_from_x(_p(x), _p(T), None, None)
        return T.reshape(4, 4, order="F")
