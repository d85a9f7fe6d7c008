#!/usr/bin/env python3
"""Device-resident vs host-driven ICP solve, iteration by iteration (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests.test_gpu_device_lm import host_lm
rng = np.random.default_rng(4)
tgt = rng.random((40_000, 3)) * 20.0
x_true = np.array([0.15, -0.1, 0.2, 0.02, -0.03, 0.025])
th = np.linalg.norm(x_true[3:]); a = x_true[3:] / th
K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
src = (tgt - x_true[:3]) @ R + rng.normal(0, 0.002, tgt.shape)
src = src[rng.permutation(len(src))][:30_000]
for k in (1, 2, 3, 5, 8, 12, 20, 27, 40):
    dev = mo.IcpCost(src, tgt, max_distance=0.6); ref = mo.IcpCost(src, tgt, max_distance=0.6)
    xd, rep = mo.capi.lm_minimize([dev], [0], np.zeros(6), max_iterations=k)
    xh, sh, ih = host_lm(ref, 0, np.zeros(6), max_iter=k)
    print(k, "dev", rep["status"], rep["iterations"], rep["sweeps"], "host", sh, ih, "max|dx| %.2e" % np.abs(xd - xh).max(), "err %.2e" % np.abs(xd - x_true).max(), flush=True)
    dev.close(); ref.close()

import time
dev = mo.IcpCost(src, tgt, max_distance=0.6); ref = mo.IcpCost(src, tgt, max_distance=0.6)
for name, run in (("device-resident", lambda: mo.capi.lm_minimize([dev], [0], np.zeros(6), max_iterations=27)),
                  ("host loop (python)", lambda: host_lm(ref, 0, np.zeros(6), max_iter=27))):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
    print("%s: 27 outer iterations (54 sweeps + 27 searches over 30 k sources / 40 k targets) in %.3f ms" % (name, best * 1e3))
