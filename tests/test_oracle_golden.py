"""The oracle (CPU restatement of the reference path) against what pins it:
  * the reference's own known-answer tests, replayed by oracle/replay_reference_tests.cpp;
  * the reference's point2point tests on its façade fixture (tst/point2point.cpp:142-217);
  * the committed golden vectors (tests/golden/p2p_1k_golden.npz).
Runs without a GPU."""
import os
import subprocess

import numpy as np

from tests import datasets as ds
from tests import oracle_binding as ob

GOLDEN = os.path.join(ds.GOLDEN, "p2p_1k_golden.npz")


def test_reference_known_answer_tests_replay(oracle):
    out = subprocess.run([ob.REPLAY], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:]
    summary = [l for l in out.stdout.splitlines() if l.startswith("SUMMARY")][0]
    assert summary.endswith("0 failures"), summary
    assert int(summary.split()[1]) >= 104


def test_facade_consistency_over_cost_classes(oracle, facade):
    """tst/point2point.cpp:142-184: the four cost classes agree on the cost (1e-7) and the
    static/dynamic twins agree on H (1e-7) at x0 = 0."""
    src, tgt = facade
    assert src.shape == (29310, 3)
    x0 = np.zeros(6)
    res = {cc: oracle.p2p_linearize(src, tgt, x0, cost_class=cc, layout=ob.LAYOUT_TST)
           for cc in (ob.ANALYTIC_STATIC, ob.ANALYTIC_DYN, ob.NUMERIC_STATIC, ob.NUMERIC_DYN)}
    s = res[ob.ANALYTIC_STATIC][2]
    for cc in res:
        assert abs(res[cc][2] - s) < 1e-7
    flat = lambda H: H.reshape(-1, order="F")[:35]
    assert np.abs(flat(res[ob.ANALYTIC_STATIC][0]) - flat(res[ob.ANALYTIC_DYN][0])).max() < 1e-7
    assert np.abs(flat(res[ob.NUMERIC_STATIC][0]) - flat(res[ob.NUMERIC_DYN][0])).max() < 1e-7


def test_facade_anchors(oracle, facade):
    """Independent anchors from the survey's scratch restatement."""
    src, tgt = facade
    x0 = np.zeros(6)
    Hn, bn, sn = oracle.p2p_linearize(src, tgt, x0, cost_class=ob.NUMERIC_DYN)
    assert abs(sn - 11726562.6975) < 1e-3
    assert np.allclose(np.diag(Hn)[:3], 29310.0, atol=1e-4)
    want_b = np.array([-296979.219, -484616.599, 95968.259, -491316.656, -1130305.845, -6340301.958])
    assert np.abs(bn - want_b).max() < 2e-3
    # row-major analytic Jacobian agrees with forward differences at w = 0
    Ha, ba, sa = oracle.p2p_linearize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN, layout=ob.LAYOUT_ROW_MAJOR)
    assert np.abs(Ha - Hn).max() / np.abs(Hn).max() < 1e-8
    # the Jacobian as written in the test gives a singular H (zero row/column 1)
    Ht, _, _ = oracle.p2p_linearize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN, layout=ob.LAYOUT_TST)
    assert not Ht[1, :].any() and not Ht[:, 1].any()


def test_facade_lm_reaches_fixture_pose(oracle, facade):
    """tst/point2point.cpp:192-217 asserts nothing; here the numerical-cost LM must land on
    (t, log R) of the fixture transform, in 5 outer iterations, status CONVERGED."""
    src, tgt = facade
    x, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.NUMERIC_DYN, max_iter=50)
    assert status == 0 and iters == 5
    assert np.abs(x - ds.FIXTURE_X).max() < 1e-7
    x, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.NUMERIC_STATIC, max_iter=50)
    assert np.abs(x - ds.FIXTURE_X).max() < 1e-7


def test_golden_vectors(oracle):
    from tests.golden import make_p2p_golden as mk
    g = np.load(GOLDEN)
    src, tgt = ds.synthetic_pair(1000, seed=42, noise=0.01)
    for m, xn, ln, cn in mk.cases():
        cc, layout = mk.MODES[m]
        lk, lp = mk.LOSSES[ln]
        H, b, s = oracle.p2p_linearize(src, tgt, mk.XS[xn], cost_class=cc, layout=layout,
                                       cov=mk.COVS[cn], loss_kind=lk, loss_param=lp)
        key = "%s/%s/%s/%s" % (m, xn, ln, cn)
        # numeric mode amplifies last-bit libm differences by 1/h; analytic is exact arithmetic
        tol = 1e-7 if m == "numeric" else 1e-12
        assert np.abs(H - g[key + "/H"]).max() <= tol * np.abs(g[key + "/H"]).max(), key
        assert np.abs(b - g[key + "/b"]).max() <= tol * np.abs(g[key + "/b"]).max(), key
        assert abs(s - g[key + "/cost"]) <= 1e-12 * g[key + "/cost"], key
    x, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.NUMERIC_DYN, max_iter=50)
    assert np.abs(x - g["lm/x"]).max() < 1e-8 and status == int(g["lm/status"])
    for xn, xv in (("zero", np.zeros(6)), ("bad", np.array([0.5, 0.5, 0.5, 0.2, 0.5, 0.5]))):
        H, b, s = oracle.camera_linearize(mk.CAMERA_PTS, mk.CAMERA_PIX, xv)
        assert np.abs(H - g["camera/%s/H" % xn]).max() <= 1e-6 * np.abs(H).max()
        assert abs(s - g["camera/%s/cost" % xn]) <= 1e-12 * s


def test_threaded_baseline_equals_sequential(oracle):
    src, tgt = ds.synthetic_pair(50_000, seed=9, noise=0.02)
    H1, b1, s1 = oracle.p2p_linearize(src, tgt, ds.X_GENERIC)
    H4, b4, s4 = oracle.p2p_linearize(src, tgt, ds.X_GENERIC, threads=4)
    assert np.abs(H1 - H4).max() <= 1e-12 * np.abs(H1).max()
    assert np.abs(b1 - b4).max() <= 1e-12 * np.abs(b1).max()
    assert abs(s1 - s4) <= 1e-12 * s1
    assert abs(oracle.p2p_cost(src, tgt, ds.X_GENERIC, threads=3) - s1) <= 1e-12 * s1


def test_product_so3_agrees_with_oracle_so3(oracle):
    """The product's closed-form SE(3) map (include/moptimizer_amd/so3.hpp) and the oracle's
    matrix-form restatement are written independently; they must agree to rounding."""
    import ctypes
    import tempfile
    src = r'''
    #include "moptimizer_amd/so3.hpp"
    extern "C" void product_se3(const double* x, double* T16) {
      moptimizer::so3::convert6DOFParameterToMatrix<double>(x, T16); }
    '''
    with tempfile.TemporaryDirectory() as d:
        cpp = os.path.join(d, "p.cpp")
        so = os.path.join(d, "p.so")
        open(cpp, "w").write(src)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I",
                               os.path.join(ds.ROOT, "include"), cpp, "-o", so])
        lib = ctypes.CDLL(so)
        rng = np.random.default_rng(0)
        for _ in range(50):
            x = rng.normal(size=6) * np.array([10, 10, 10, 1, 1, 1])
            T = np.zeros(16)
            lib.product_se3(x.ctypes.data_as(ctypes.c_void_p), T.ctypes.data_as(ctypes.c_void_p))
            assert np.abs(T.reshape(4, 4, order="F") - oracle.se3_from_x(x)).max() < 1e-14
        T = np.zeros(16)
        x = np.zeros(6)
        lib.product_se3(x.ctypes.data_as(ctypes.c_void_p), T.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(T.reshape(4, 4, order="F"), np.eye(4))


def test_text_cloud_loader_round_trip(facade, tmp_path):
    """include/moptimizer_amd/cloud_io.hpp reads the reference's `x y z r g b` text format
    (tst/point2point.cpp:125-138): written with 8 decimals like tst/data/fachada.txt, the façade
    comes back bit for bit."""
    import ctypes
    src, _ = facade
    txt = os.path.join(tmp_path, "cloud.txt")
    with open(txt, "w") as f:
        for x, y, z in src[:2000]:
            f.write("%.8f %.8f %.8f 255 218 0\n" % (x, y, z))
        f.write("not a number\n7 8 9 1 2 3\n")   # parsing stops here, as the reference's loop does
    code = r'''
    #include "moptimizer_amd/cloud_io.hpp"
    extern "C" long load_cloud(const char* path, double* out, long cap) {
      auto v = moptimizer::io::loadXyzRgbText<double>(path);
      for (long i = 0; i < (long)v.size() && i < cap; ++i) out[i] = v[i];
      return (long)v.size(); }
    '''
    cpp = os.path.join(tmp_path, "l.cpp")
    so = os.path.join(tmp_path, "l.so")
    open(cpp, "w").write(code)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I",
                           os.path.join(ds.ROOT, "include"), cpp, "-o", so])
    lib = ctypes.CDLL(so)
    lib.load_cloud.restype = ctypes.c_long
    out = np.zeros(3 * 4000)
    n = lib.load_cloud(txt.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(out.size))
    assert n == 3 * 2000
    assert np.array_equal(out[:n].reshape(-1, 3), src[:2000])
