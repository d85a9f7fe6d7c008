#!/usr/bin/env python3
"""Time-to-resident for host arrays: cost construction (PCIe copy + relayout) at several N."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds
for n in (1_000_000, 10_000_000):
    src, tgt = ds.synthetic_pair(n, seed=2)
    for rep in range(3):
        t0 = time.perf_counter()
        c = mo.Point2PointCost(src, tgt)
        dt = time.perf_counter() - t0
        c.close()
        print("n=%d create from host arrays: %.1f ms (%.1f GB/s)" % (n, dt * 1e3, n * 48 / dt / 1e9), flush=True)
