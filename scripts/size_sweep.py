#!/usr/bin/env python3
"""Blocking-call time, sweep-kernel time and algorithmic bandwidth over problem size, for the
three sweeps of the path (analytic linearize, forward-difference linearize, cost only), fp64 and
fp32.  Writes profiles/<tag>_size_sweep.json and prints a markdown table.
Usage: python scripts/size_sweep.py [--tag r1] [--max-n 100000000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(cost, x, mode, iters):
    fn = (lambda: cost.compute_cost(x)) if mode is None else (lambda: cost.linearize(x, mode))
    for _ in range(5):
        fn()
    cost.set_profiling(False)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    call_us = (time.perf_counter() - t0) / iters * 1e6
    cost.set_profiling(True)
    for _ in range(max(8, iters // 4)):
        fn()
    ms, cnt = cost.profile()
    cost.set_profiling(False)
    return call_us, ms / cnt * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r1")
    ap.add_argument("--max-n", type=int, default=300_000_000)
    args = ap.parse_args()
    import torch
    import moptimizer_0_amd as mo
    from bench import make_shard_on_gpu, quiesce_python_gc
    from tests import datasets as ds

    torch.cuda.set_device(0)
    quiesce_python_gc()
    rows = []
    for dtype, tdt, ndt, bpp in (("f64", torch.float64, np.float64, 48), ("f32", torch.float32, np.float32, 24)):
        x = ds.X_GENERIC.astype(ndt)
        for n in (1_000, 10_000, 100_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000, 300_000_000):
            if n > args.max_n:
                continue
            src, tgt = make_shard_on_gpu(torch, n, 0, tdt)
            torch.cuda.synchronize()
            cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), dtype=ndt, device_ptrs=True, count=n)
            cost.set_speculation(False)
            del src, tgt
            torch.cuda.empty_cache()
            iters = 400 if n <= 1_000_000 else (100 if n <= 10_000_000 else 20)
            for name, mode in (("analytic", 0), ("numeric", 2), ("cost", None)):
                call_us, kernel_us = measure(cost, x, mode, iters)
                gbs = n * bpp / (kernel_us * 1e-6) / 1e9
                rows.append({"dtype": dtype, "n": n, "sweep": name, "call_us": call_us,
                             "kernel_us": kernel_us, "kernel_GBps": gbs, "frac_of_8TBps": gbs / 8000.0,
                             "correspondences_per_s": n / (call_us * 1e-6)})
                print("| %s | %11d | %-8s | %9.1f | %9.1f | %7.0f | %5.1f %% | %.3g |" %
                      (dtype, n, name, call_us, kernel_us, gbs, gbs / 80.0, n / (call_us * 1e-6)), flush=True)
            cost.close()
    out = os.path.join(ROOT, "gpurun_out", "%s_size_sweep.json" % args.tag)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
