#!/usr/bin/env python3
"""Kernel-time survey on one GPU: for each (N, mode, variant, workgroups per CU) run a few sweeps
and print the HIP-event time of the dominant kernel and the algorithmic GB/s (48 B/correspondence).
Usage: python scripts/sweep_timing.py [--n 10000000 1000000] [--bpc 2 4 8]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[10_000_000, 1_000_000])
    ap.add_argument("--bpc", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dtype", default="f64")
    args = ap.parse_args()
    import torch
    import moptimizer_0_amd as mo
    from bench import make_shard_on_gpu
    from tests import datasets as ds

    torch.cuda.set_device(0)
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    ndt = np.float64 if args.dtype == "f64" else np.float32
    bpp = 48 if args.dtype == "f64" else 24
    x = ds.X_GENERIC.astype(ndt)
    for n in args.n:
        src, tgt = make_shard_on_gpu(torch, n, 0, tdt)
        torch.cuda.synchronize()
        cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), dtype=ndt, device_ptrs=True,
                                  count=n)
        cost.set_speculation(False)  # time real sweeps, not answers from the kept result
        del src, tgt
        for bpc in args.bpc:
            os.environ["MOPT_BLOCKS_PER_CU"] = str(bpc)
            for name, mode, variant in (("analytic/moments", 0, 2), ("analytic/literal", 0, 1),
                                        ("numeric/literal", 2, 1), ("numeric/moments", 2, 2),
                                        ("cost", None, 0)):
                cost.set_kernel_variant(variant)
                for _ in range(3):
                    cost.compute_cost(x) if mode is None else cost.linearize(x, mode)
                cost.set_profiling(True)
                for _ in range(args.iters):
                    cost.compute_cost(x) if mode is None else cost.linearize(x, mode)
                ms, cnt = cost.profile()
                cost.set_profiling(False)
                k = ms / cnt
                print("n=%9d bpc=%d %-17s kernel %8.2f us  %7.1f GB/s  %5.1f%% of 8 TB/s" %
                      (n, bpc, name, k * 1e3, n * bpp / (k * 1e-3) / 1e9,
                       n * bpp / (k * 1e-3) / 1e9 / 80.0), flush=True)
        cost.close()


if __name__ == "__main__":
    main()
