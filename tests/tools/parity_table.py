#!/usr/bin/env python3
"""Measured parity (norm-wise relative error of H, b; relative error of the cost) of the HIP path
against the CPU restatement: per Jacobian mode / kernel variant / size, and — for forward
differences, whose step is h_j = sqrt(eps) |x_j| (linearization.h:85) — per decade of |x_j|."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds, oracle_binding as ob
o = ob.load()
def rel(a, b): return np.abs(np.asarray(a, float) - b).max() / np.abs(b).max()
print("| N | mode | kernel | x | max|dH|/max|H| | max|db|/max|b| | |dcost|/cost |")
print("|---|---|---|---|---|---|---|")
for n in (1000, 100_000, 1_000_000):
    src, tgt = ds.synthetic_pair(n, seed=42, noise=0.01)
    cost = mo.Point2PointCost(src, tgt)
    for mode, mname in ((0, "analytic"), (1, "analytic, tst layout"), (3, "analytic, left perturbation"),
                        (4, "analytic, right perturbation"), (2, "forward differences")):
        cc = ob.NUMERIC_DYN if mode == 2 else ob.ANALYTIC_DYN
        layout = {1: ob.LAYOUT_TST, 3: ob.LAYOUT_LEFT, 4: ob.LAYOUT_RIGHT}.get(mode, ob.LAYOUT_ROW_MAJOR)
        for variant, vname in ((3, "moments"), (1, "literal")):
            cost.set_kernel_variant(variant)
            for x, xname in ((ds.X_ZERO, "0"), (ds.X_GENERIC, "generic")):
                H, b, s = cost.linearize(x, mode)
                Hr, br, sr = o.p2p_linearize(src, tgt, x, cost_class=cc, layout=layout)
                print("| %d | %s | %s | %s | %.1e | %.1e | %.1e |" % (n, mname, vname, xname, rel(H, Hr), rel(b, br), abs(s - sr) / sr), flush=True)
    cost.close()

print()
print("Forward differences by decade of |x_j| (20 random x per decade, worst case; N = 100 000):")
print()
print("| |x_j| ~ | h_j ~ | kernel | worst max|dH|/max|H| | worst max|db|/max|b| |")
print("|---|---|---|---|---|")
rng = np.random.default_rng(5)
src, tgt = ds.synthetic_pair(100_000, seed=42, noise=0.01)
cost = mo.Point2PointCost(src, tgt)
COVS = (("", None), (", symmetric covariance", np.array([[2.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 1.5]])),
        (", general covariance + Geman-McClure(100)", np.array([[2.0, 0.7, -0.1], [0.3, 0.5, 0.9], [-0.4, 0.2, 1.5]])))
for scale in (1e-8, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1, 1.0):
    for variant, vname in ((3, "moments (MOMENTS_ALWAYS)"), (1, "literal"), (0, "AUTO")):
        for cname, cov in COVS:
            if cov is not None and variant != 1:
                continue
            loss = (1, 100.0) if "Geman" in cname else (0, 0.0)
            cost.set_kernel_variant(variant)
            cost.set_covariance(cov)
            cost.set_loss(*loss)
            wH = wb = 0.0
            for _ in range(20):
                x = rng.choice([-1.0, 1.0], 6) * scale * rng.uniform(0.3, 3.0, 6)
                H, b, s = cost.linearize(x, 2)
                Hr, br, sr = o.p2p_linearize(src, tgt, x, cost_class=ob.NUMERIC_DYN, layout=ob.LAYOUT_ROW_MAJOR,
                                             cov=cov, loss_kind=loss[0], loss_param=loss[1])
                wH, wb = max(wH, rel(H, Hr)), max(wb, rel(b, br))
            print("| %.0e | %.1e | %s%s | %.1e | %.1e |" % (scale, 1.49e-8 * scale, vname, cname, wH, wb), flush=True)
cost.close()

print()
print("Other models against the CPU restatement (forward differences; the device evaluates exp / sin / cos / acos")
print("with its own libm, so a last-bit difference in a residual is amplified by eps / h_j):")
print()
print("| model | n, m | elements | x | max|dH|/max|H| | max|db|/max|b| | |dcost|/cost |")
print("|---|---|---|---|---|---|---|")
pts, pix = ds.synthetic_camera(100_000, seed=17)
cam = mo.ReprojectionCost(pts, pix)
for x, xname in ((np.zeros(6), "0"), (np.array([-0.01, 0.02, -0.058, 0.018, -0.0013, 0.027]), "near the optimum")):
    H, b, s = cam.linearize(x, 2)
    Hr, br, sr = o.camera_linearize(pts, pix, x)
    print("| reprojection (tst/camera_calibration.cpp) | 6, 2 | 100000 | %s | %.1e | %.1e | %.1e |" % (xname, rel(H, Hr), rel(b, br), abs(s - sr) / sr))
cam.close()
t = np.linspace(0.0, 5.0, 50_000)
y = np.exp(0.3 * t + 0.1) + 0.01 * np.random.default_rng(2).standard_normal(t.size)
curve = mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t, y)
for x in (np.zeros(2), np.array([0.29, 0.13])):
    H, b, s = curve.linearize(x, 2)
    Hr, br, sr = o.scalar_linearize(1, t, y, x, numeric=True)
    print("| exp curve (tst/curve_fitting.cpp) | 2, 1 | 50000 | %s | %.1e | %.1e | %.1e |" % (np.array2string(x), rel(H, Hr), rel(b, br), abs(s - sr) / sr))
curve.close()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import STATE_RESIDUAL  # the 15-state model's residual as HIP source
x_init = np.zeros(15); x_init[:6] = [0.6, 0.8, 0.3, -0.4, 0.11, -0.9]
state = mo.JitModelCost(15, 15, STATE_RESIDUAL, planes=x_init.reshape(15, 1))
xs = np.zeros(15); xs[:6] = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6]
for x, xname in ((xs, "the test's start"), (x_init + 0.01, "near the optimum")):
    H, b, s = state.linearize(x, 2)
    Hr, br, sr = o.state_linearize(x_init, x)
    print("| state model (tst/state_model.cpp), wide sweep | 15, 15 | 1 | %s | %.1e | %.1e | %.1e |" % (xname, rel(H, Hr), rel(b, br), abs(s - sr) / sr))
state.close()
xr = np.array([0.038, 0.194, 0.425, 0.626, 1.253, 2.5, 3.70]); yr = np.array([0.05, 0.127, 0.094, 0.2122, 0.2729, 0.2665, 0.3317])
rat = mo.ScalarModelCost(mo.capi.MODEL_RATIONAL, xr, yr)
for x in (np.array([0.9, 0.2]), np.array([0.362, 0.556])):
    H, b, s = rat.linearize(x, 2)
    Hr, br, sr = o.scalar_linearize(2, xr, yr, x, numeric=True)
    print("| rational model (tst/test_models.h) | 2, 1 | 7 | %s | %.1e | %.1e | %.1e |" % (np.array2string(x), rel(H, Hr), rel(b, br), abs(s - sr) / sr))
rat.close()
pw = mo.ScalarModelCost(mo.capi.MODEL_POWELL)
for x in (np.array([3.0, -1.0, 0.0, 4.0]), np.array([0.3, -0.1, 0.07, 0.4])):
    H, b, s = pw.linearize(x, 2)
    Hr, br, sr = o.scalar_linearize(3, None, None, x, numeric=True)
    print("| Powell (tst/powell.cpp) | 4, 4 | 1 | %s | %.1e | %.1e | %.1e |" % (np.array2string(x), rel(H, Hr), rel(b, br), abs(s - sr) / sr))
pw.close()
