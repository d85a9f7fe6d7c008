#!/usr/bin/env python3
"""Per-call cost of the collective code path on one GPU: a 1-rank RCCL communicator runs the same
sequence N ranks run (sweep, finalize, ncclAllReduce, publish kernel)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import moptimizer_0_amd as mo
from bench import make_shard_on_gpu, quiesce_python_gc
from tests import datasets as ds
torch.cuda.set_device(0)
quiesce_python_gc()
for n in (1_000_000, 10_000_000):
    src, tgt = make_shard_on_gpu(torch, n, 0, torch.float64)
    torch.cuda.synchronize()
    for with_comm in (False, True):
        cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), device_ptrs=True, count=n)
        cost.set_speculation(False)
        if with_comm:
            cost.comm_init_rank(mo.capi.comm_unique_id(), 0, 1)
        x = ds.X_GENERIC
        for _ in range(30): cost.linearize(x, 0)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter(); cost.linearize(x, 0); ts.append(time.perf_counter() - t0)
        print("n=%9d comm=%-5s median %.1f us p90 %.1f us" % (n, with_comm, np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6), flush=True)
        cost.close()
