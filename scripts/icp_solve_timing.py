#!/usr/bin/env python3
"""(--surface [--shift-radii 0.5]: both scans on a wavy sheet instead of filling the box)
A whole ICP solve at the size of BASELINE configs 2 / 3: N sources against N targets (the moved
sources plus noise, shuffled), correspondences searched again at the top of every outer iteration
(cost->update(x), levenberg_marquadt_dyn.cpp:54), point2point forward differences, under the
device-resident loop (search, sweep and LM step all queued on the GPU).
Usage: python scripts/icp_solve_timing.py [--n 1000000] [--max-dist-spacings 1.0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo

def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default

n = int(arg("--n", "1000000"))
side = 100.0
spacing = side / n ** (1.0 / 3.0)
max_dist = float(arg("--max-dist-spacings", "1.0")) * spacing
rng = np.random.default_rng(9)
src = rng.random((n, 3)) * side
x_true = np.array([0.2 * spacing, -0.15 * spacing, 0.1 * spacing, 0.0008, -0.0005, 0.0011])
if "--surface" in sys.argv:
    # a scanned surface (a wavy sheet through the box), the radius in point spacings ON THE SHEET, the
    # second scan moved by --shift-radii of the radius plus a small rotation
    spacing = side / np.sqrt(n)
    max_dist = float(arg("--max-dist-spacings", "4.0")) * spacing
    uv = rng.random((n, 2)) * side
    src = np.column_stack([uv, 10.0 * np.sin(uv[:, 0] / 10.0) * np.cos(uv[:, 1] / 15.0) + 50.0])
    shift = float(arg("--shift-radii", "0.5")) * max_dist
    x_true = np.array([0.6 * shift, -0.5 * shift, 0.62 * shift, 0.0008, -0.0005, 0.0011])
th = np.linalg.norm(x_true[3:]); a = x_true[3:] / th
K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
tgt = (src @ R.T + x_true[:3] + rng.normal(0, 0.01 * spacing, src.shape))[rng.permutation(n)]
mo.capi.device_count()
t0 = time.perf_counter()
cost = mo.IcpCost(src, tgt, max_dist)
build = time.perf_counter() - t0
best, rep, x = 1e9, None, None
for _ in range(5):
    t0 = time.perf_counter()
    x, rep = mo.capi.lm_minimize([cost], [2], np.zeros(6), max_iterations=50)
    best = min(best, time.perf_counter() - t0)
print("ICP solve, %d x %d, radius %.2f spacings (grid reach %d): construction %.1f ms; device-resident solve %.3f ms, "
      "%d outer iterations, %d sweeps, status %d, max |x - x_true| %.2e"
      % (n, n, max_dist / spacing, cost.grid()[1], build * 1e3, best * 1e3, rep["iterations"], rep["sweeps"],
         rep["status"], np.abs(x - x_true).max()), flush=True)
cost.close()
