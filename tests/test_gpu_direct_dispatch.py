"""The blocking sweeps dispatched by the library itself (csrc/aql.hpp: AQL packets with agent-scope
fences into HSA queues of its own) against the same calls on the HIP stream (MOPT_AQL=0): the same
kernels with the same arguments, so the same bits — every sweep kind the direct path serves, a path
switch in the middle (profiling, a device-resident solve, an asynchronous call on the cost's stream),
linked costs, and many costs sharing the two queues."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import datasets as ds

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import moptimizer_0_amd as mo
from tests import datasets as ds

out = {}
src, tgt = ds.synthetic_pair(200_003, seed=3, noise=0.02)
x = ds.X_GENERIC
cov = np.array([[2.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 1.5]])
c = mo.Point2PointCost(src, tgt)
c.set_speculation(False)
rows = []
for variant in (mo.KERNEL_AUTO, mo.KERNEL_LITERAL):
    c.set_kernel_variant(variant)
    for mode in (mo.JAC_ANALYTIC, mo.JAC_ANALYTIC_TST_LAYOUT, mo.JAC_NUMERIC, mo.JAC_ANALYTIC_LEFT, mo.JAC_ANALYTIC_RIGHT):
        for cv, loss in ((None, 0), (cov, 1)):
            c.set_covariance(cv)
            c.set_loss(loss, 100.0)
            H, b, s = c.linearize(x, mode)
            rows.append(np.concatenate([H.ravel(), b, [s, c.compute_cost(x)]]))
c.set_covariance(None); c.set_loss(0, 0.0); c.set_kernel_variant(mo.KERNEL_AUTO)
direct_before = c.direct_dispatches()
# path switches: profiled sweeps, a device-resident solve on the stream, then blocking sweeps again
c.set_profiling(1)
H, b, s = c.linearize(x, mo.JAC_ANALYTIC); rows.append(np.concatenate([H.ravel(), b, [s, 0.0]]))
ms, launches = c.profile(); c.set_profiling(0)
xs, rep = mo.capi.lm_minimize([c], [mo.JAC_ANALYTIC], np.zeros(6), max_iterations=3)
rows.append(np.concatenate([xs, np.zeros(38)]))
H, b, s = c.linearize(xs, mo.JAC_NUMERIC); rows.append(np.concatenate([H.ravel(), b, [s, c.compute_cost(xs)]]))
out["profiled_launches"] = int(launches)
out["profiled_ms"] = float(ms)
# float32, and the reprojection costs of config 5 (linked)
f = mo.Point2PointCost(src.astype(np.float32), tgt.astype(np.float32), dtype=np.float32)
H, b, s = f.linearize(x.astype(np.float32), mo.JAC_NUMERIC)
rows.append(np.concatenate([H.ravel(), b, [s, f.compute_cost(x.astype(np.float32))]]).astype(np.float64))
pts, pix = ds.synthetic_camera(30_000, seed=5)
cams = [mo.ReprojectionCost(pts[:12_000], pix[:12_000]), mo.ReprojectionCost(pts[12_000:], pix[12_000:])]
for k in cams: k.set_loss(mo.LOSS_GEMAN_MCCLURE, 100.0)
mo.capi.link_costs(cams)
for it in range(3):
    xc = np.full(6, 1e-3 * it)
    for k in cams:
        H, b, s = k.linearize(xc, mo.JAC_NUMERIC)
        rows.append(np.concatenate([H.ravel(), b, [s, k.compute_cost(xc)]]))
mo.capi.link_costs([])
# a built-in scalar model (tst/test_models.h rational model, tst/curve_fitting.cpp exp curve)
t = np.linspace(0.0, 4.95, 5003)
curves = [mo.ScalarModelCost(mo.capi.MODEL_RATIONAL, t, 0.36 * t / (0.56 + t)),
          mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t, np.exp(0.3 * t + 0.1))]
for sc, modes in zip(curves, ((mo.JAC_ANALYTIC, mo.JAC_NUMERIC), (mo.JAC_NUMERIC,))):
    for mode in modes:
        xs2 = np.array([0.31, 0.52])
        H, b, s = sc.linearize(xs2, mode)
        row = np.zeros(44); v = np.concatenate([H.ravel(), b, [s, sc.compute_cost(xs2)]]); row[:v.size] = v
        rows.append(row)
# A profiled sweep of a built-in scalar model is timed with a recorded event pair, so its launch leaves the
# direct path it was offered for the HIP stream; the call must then BE a stream call (wait, fault check,
# ordering of the next direct sweep): same numbers, counted launches, and direct again afterwards
sc = curves[0]
d0 = sc.direct_dispatches()
sc.set_profiling(1)
for k in range(3):
    H, b, s = sc.linearize(np.array([0.31, 0.52]) + 1e-3 * k, mo.JAC_ANALYTIC)
    row = np.zeros(44); v = np.concatenate([H.ravel(), b, [s, 0.0]]); row[:v.size] = v; rows.append(row)
ms_sc, n_sc = sc.profile(); sc.set_profiling(0)
out["scalar_profiled"] = [int(n_sc), float(ms_sc), sc.direct_dispatches() - d0]
for k in range(2):
    H, b, s = sc.linearize(np.array([0.4, 0.5]) + 1e-3 * k, mo.JAC_ANALYTIC)
    row = np.zeros(44); v = np.concatenate([H.ravel(), b, [s, 0.0]]); row[:v.size] = v; rows.append(row)
out["scalar_direct_after"] = sc.direct_dispatches() - d0
# profiling a sharded cost (host-slot combine, one rank): the direct path's own dispatch stamps are collected
h = mo.Point2PointCost(src[:50_000], tgt[:50_000])
h.set_speculation(False)
h.hostcomm_attach("/mopt-test-prof-%%d" %% os.getpid(), 0, 1)
h.set_combine(mo.COMBINE_HOST)
h.set_profiling(1)
for k in range(4):
    H, b, s = h.linearize(x + 1e-3 * k, mo.JAC_ANALYTIC)
    rows.append(np.concatenate([H.ravel(), b, [s, 0.0]]))
ms_h, n_h = h.profile(); h.set_profiling(0)
out["host_profiled"] = [int(n_h), float(ms_h), h.direct_dispatches()]
h.close()
# many costs over the two queues, interleaved
many = [mo.Point2PointCost(src[k * 1000:(k + 1) * 1000 + 37], tgt[k * 1000:(k + 1) * 1000 + 37]) for k in range(12)]
for rnd in range(3):
    for k, m in enumerate(many):
        H, b, s = m.linearize(x + 1e-3 * rnd, mo.JAC_ANALYTIC)
        rows.append(np.concatenate([H.ravel(), b, [s, float(k)]]))
out["rows"] = np.array(rows).tolist()
out["direct"] = [c.direct_dispatches(), f.direct_dispatches(), cams[0].direct_dispatches(), many[0].direct_dispatches(),
                 curves[0].direct_dispatches(), curves[1].direct_dispatches()]
out["direct_before"] = direct_before
out["sweeps"] = c.stats()[0]
print("RESULT " + json.dumps(out))
"""


def _run(aql):
    env = dict(os.environ, MOPT_AQL=aql)
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ds.ROOT}], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_direct_dispatch_gives_the_same_bits_as_the_hip_stream(hip_lib):
    direct, stream = _run("1"), _run("0")
    assert stream["direct"] == [0, 0, 0, 0, 0, 0] and stream["direct_before"] == 0
    # the direct path was taken: by the point2point cost for (nearly) every blocking sweep, by the
    # float cost, the reprojection costs, the small costs sharing the queues and the built-in scalar models
    assert direct["direct_before"] >= 0.9 * 40 and direct["direct"][0] > direct["direct_before"]
    assert all(n > 0 for n in direct["direct"][1:])
    a, b = np.array(direct["rows"]), np.array(stream["rows"])
    assert a.shape == b.shape and a.shape[0] > 60
    assert np.array_equal(a, b), np.abs(a - b).max()
    # a profiled sweep is timed on the path it takes, with a plausible duration either way
    for run in (direct, stream):
        assert run["profiled_launches"] == 1 and 1e-3 < run["profiled_ms"] < 1.0, run["profiled_ms"]
        # the scalar model's profiled sweeps: three launches timed, none of them direct
        n_sc, ms_sc, went_direct = run["scalar_profiled"]
        assert n_sc == 3 and 1e-3 < ms_sc < 1.0 and went_direct == 0, run["scalar_profiled"]
        n_h, ms_h, _ = run["host_profiled"]
        assert n_h == 4 and 1e-3 < ms_h < 1.0, run["host_profiled"]
    assert direct["scalar_direct_after"] == 2 and stream["scalar_direct_after"] == 0
    assert direct["host_profiled"][2] == 4 and stream["host_profiled"][2] == 0


TRIM_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import moptimizer_0_amd as mo
from tests import datasets as ds

src, tgt = ds.synthetic_pair(50_001, seed=4, noise=0.02)
out = {"refused": False, "rows": [], "direct": []}
mo.capi.device_trim(0)  # nothing held yet: fine
for life in range(3):
    c = mo.Point2PointCost(src, tgt)
    d = mo.Point2PointCost(src[:999], tgt[:999])
    for k in range(40):
        H, b, s = c.linearize(ds.X_GENERIC + 1e-4 * k, mo.JAC_ANALYTIC)  # (a repeated x is answered from the kept result)
        Hd, bd, sd = d.linearize(ds.X_GENERIC + 1e-4 * k, mo.JAC_NUMERIC)
    out["rows"].append(np.concatenate([H.ravel(), b, [s], Hd.ravel(), bd, [sd]]).tolist())
    out["direct"].append(c.direct_dispatches())
    c.close()
    try:
        mo.capi.device_trim(0)
    except mo.MoptError:
        out["refused"] = True  # d is alive
    d.close()
    mo.capi.device_trim(0)
print("RESULT " + json.dumps(out))
"""


def test_queues_given_back_and_created_again(hip_lib):
    """mopt_device_trim: refused while a cost lives on the device; after it, the next cost creates
    the queues again and its sweeps go the direct way with the same results."""
    out = subprocess.run([sys.executable, "-c", TRIM_SCRIPT % {"root": ds.ROOT}], env=dict(os.environ, MOPT_AQL="1"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    assert got["refused"] is True
    assert all(n >= 36 for n in got["direct"]), got["direct"]
    assert got["rows"][0] == got["rows"][1] == got["rows"][2]
