#!/usr/bin/env python3
"""Construction of a 1 M x 1 M ICP cost (grid build, sources ordered, first search) from host arrays
and from torch tensors in device memory (mopt_icp_create_from, MOPT_INPUT_DEVICE)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import moptimizer_0_amd as mo
n = 1_000_000
rng = np.random.default_rng(1)
tgt = rng.random((n, 3)) * 100.0
src = tgt[rng.permutation(n)] + rng.normal(0, 0.01, (n, 3))
d_src, d_tgt = torch.tensor(src, device="cuda:0"), torch.tensor(tgt, device="cuda:0")
for name, a, b in (("host arrays", src, tgt), ("device tensors", d_src, d_tgt)):
    mo.IcpCost(a, b, 1.0).close()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); c = mo.IcpCost(a, b, 1.0); ts.append(time.perf_counter() - t0); c.close()
    print("construction of a 1 M x 1 M ICP cost from %s: median %.2f ms, min %.2f ms" % (name, np.median(ts) * 1e3, min(ts) * 1e3))
