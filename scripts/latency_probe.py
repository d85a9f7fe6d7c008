#!/usr/bin/env python3
"""Wall time per blocking C-ABI call (what LM sees) for several N, with profiling off and on."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import moptimizer_0_amd as mo
from bench import make_shard_on_gpu
from tests import datasets as ds

torch.cuda.set_device(0)
x = ds.X_GENERIC
for n in (1000, 1_000_000, 10_000_000):
    src, tgt = make_shard_on_gpu(torch, n, 0, torch.float64)
    torch.cuda.synchronize()
    cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), device_ptrs=True, count=n)
    cost.set_speculation(False)
    for prof in (False, True):
        cost.set_profiling(prof)
        for what in ("linearize", "cost"):
            f = (lambda: cost.linearize(x, 0)) if what == "linearize" else (lambda: cost.compute_cost(x))
            for _ in range(20):
                f()
            iters = 300
            ts = []
            for _ in range(iters):
                t0 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t0)
            dt = float(np.median(ts))
            ms, cnt = cost.profile() if prof else (0.0, 1)
            print("n=%9d profiling=%-5s %-9s median wall %8.2f us/call   kernel %8.2f us" %
                  (n, prof, what, dt * 1e6, ms / max(cnt, 1) * 1e3), flush=True)
            if prof:
                cost.set_profiling(True)
    cost.close()
