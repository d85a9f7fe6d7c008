#!/usr/bin/env python3
"""Kernel time of the forward-difference sweep evaluated as the reference does (MOPT_KERNEL_LITERAL,
numeric mode) under identity, symmetric and general covariances, with and without the robust loss,
against the moments sweep: HIP-event (dispatch-timestamp) time and the fraction of the 8 TB/s roof.
Usage: python scripts/fd_cov_timing.py [--n 10000000 1000000] [--dtype f64]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[10_000_000, 1_000_000])
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    args = ap.parse_args()
    import torch

    import moptimizer_0_amd as mo
    from tests import datasets as ds

    dt = np.float64 if args.dtype == "f64" else np.float32
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    bpc = 48 if args.dtype == "f64" else 24
    covs = {"identity": np.eye(3),
            "symmetric": np.array([[2.0, 0.3, -0.1], [0.3, 1.5, 0.2], [-0.1, 0.2, 0.8]]),
            "general": np.array([[2.0, 0.5, -0.1], [0.3, 1.5, 0.4], [-0.3, 0.2, 0.8]])}
    x = ds.X_GENERIC.astype(dt)
    for n in args.n:
        g = torch.Generator(device="cuda"); g.manual_seed(7)
        src = (torch.rand((n, 3), generator=g, device="cuda", dtype=torch.float64) * 10).to(tdt).contiguous()
        tgt = (src + 0.01 * torch.randn((n, 3), generator=g, device="cuda", dtype=tdt)).contiguous()
        torch.cuda.synchronize()
        cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), device=0, dtype=dt, device_ptrs=True, count=n)
        cost.set_speculation(False)
        for variant, vname in ((mo.KERNEL_LITERAL, "literal"), (mo.KERNEL_MOMENTS, "moments")):
            cost.set_kernel_variant(variant)
            for loss in (False, True):
                cost.set_loss(mo.LOSS_GEMAN_MCCLURE if loss else mo.LOSS_NONE, 100.0)
                for cname, cov in covs.items():
                    if vname == "moments" and (loss or cname != "identity"):
                        continue
                    cost.set_covariance(cov)
                    for _ in range(5):
                        cost.linearize(x, mo.JAC_NUMERIC)
                    cost.set_profiling(1)
                    t0 = time.perf_counter()
                    for _ in range(args.iters):
                        cost.linearize(x, mo.JAC_NUMERIC)
                    wall = (time.perf_counter() - t0) / args.iters
                    ms, cnt = cost.profile()
                    cost.set_profiling(False)
                    k = ms / cnt * 1e3
                    print("n=%9d %s numeric %-8s cov=%-9s loss=%-5s kernel %8.2f us  %6.0f GB/s  frac %.3f  call %8.2f us"
                          % (n, args.dtype, vname, cname, "GM" if loss else "none", k,
                             n * bpc / (k * 1e-6) / 1e9, n * bpc / (k * 1e-6) / 8e12, wall * 1e6), flush=True)
        cost.close()
        del src, tgt


if __name__ == "__main__":
    main()
