#!/usr/bin/env python3
"""Throughput of the generic sweeps (built-in scalar models and run-time compiled user models) at
data-set sizes far beyond the reference's tests: HIP-event time of the sweep kernel and the
algorithmic GB/s (planes x sizeof(S) per element).
Usage: python scripts/jit_timing.py [--n 10000000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def quiesce_python_gc():
    """As bench.py: a full collection of the interpreter's cyclic collector (~36-40 ms over what the
    imports leave behind) is triggered by the small arrays a loop of calls allocates — in round 4's
    file it landed in one row's 30 calls (1 901 us per call for a 9.4 us kernel).  Freeze what is
    alive now; the collector stays on."""
    import gc
    gc.collect()
    gc.freeze()


def timed(cost, x, mode, iters):
    """kernel: dispatch-stamped events, a pass of its own (a profiled launch costs the host ~10 us);
    call: median wall time of an unprofiled blocking call, a second pass."""
    cost.set_speculation(False)
    for _ in range(3):
        cost.linearize(x, mode)
    cost.set_profiling(True)
    for _ in range(iters):
        cost.linearize(x, mode)
    ms, cnt = cost.profile()
    cost.set_profiling(False)
    call, xin, _, _, _ = cost.bound_linearize(mode)
    xin[:] = x
    walls = []
    for _ in range(iters):
        t0 = time.perf_counter()
        call()
        walls.append(time.perf_counter() - t0)
    return ms / cnt * 1e3, float(np.median(walls)) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[10_000_000, 1_000_000])
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    import moptimizer_0_amd as mo

    quiesce_python_gc()
    rng = np.random.default_rng(3)
    for n in args.n:
        t = np.linspace(0.0, 4.95, n)
        y = np.exp(0.3 * t + 0.1) + rng.normal(0, 0.2, n)
        x = np.array([0.29, 0.13])
        t0 = time.perf_counter()
        jit = mo.JitModelCost(2, 1, "r[0] = d[1] - exp(x[0] * d[0] + x[1]);",
                              "const S e = exp(x[0] * d[0] + x[1]); J[0] = -d[0] * e; J[1] = -e;",
                              planes=np.stack([t, y]))
        create_ms = (time.perf_counter() - t0) * 1e3
        builtin = mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t, y)
        for name, cost, mode in (("built-in exp curve, numeric", builtin, 2),
                                 ("jit exp curve, numeric", jit, 2),
                                 ("jit exp curve, analytic", jit, 0)):
            k, wall = timed(cost, x, mode, args.iters)
            print("n=%9d %-30s kernel %8.2f us  %7.1f GB/s  call %8.2f us" %
                  (n, name, k, n * 16 / (k * 1e-6) / 1e9, wall), flush=True)
        print("n=%9d jit create (compile + upload) %.1f ms" % (n, create_ms), flush=True)
        # the built-in model with a Jacobian of its own (tst/test_models.h:7-20) and its run-time compiled twin
        yr = 0.36 * t / (0.56 + t) + rng.normal(0, 0.02, n)
        rational = mo.ScalarModelCost(mo.capi.MODEL_RATIONAL, t, yr)
        jit_rational = mo.JitModelCost(2, 1, "r[0] = d[1] - (x[0] * d[0]) / (x[1] + d[0]);",
                                       "const S q = x[1] + d[0]; J[0] = -d[0] / q; J[1] = (x[0] * d[0]) / (q * q);",
                                       planes=np.stack([t, yr]))
        for name, cost, mode in (("built-in rational, analytic", rational, 0),
                                 ("built-in rational, numeric", rational, 2),
                                 ("jit rational, analytic", jit_rational, 0),
                                 ("jit rational, numeric", jit_rational, 2)):
            k, wall = timed(cost, np.array([0.3, 0.5]), mode, args.iters)
            print("n=%9d %-30s kernel %8.2f us  %7.1f GB/s  call %8.2f us" %
                  (n, name, k, n * 16 / (k * 1e-6) / 1e9, wall), flush=True)
        planes = rng.normal(0, 1.0, (3, n))
        planes[0] = np.linspace(0.0, 6.0, n)
        osc = mo.JitModelCost(
            3, 2,
            "const S e = exp(-x[1] * d[0]); r[0] = d[1] - x[0] * e * cos(x[2] * d[0]);"
            "r[1] = d[2] - x[0] * e * sin(x[2] * d[0]);", planes=planes)
        k, wall = timed(osc, np.array([1.3, 0.4, 2.1]), 2, args.iters)
        print("n=%9d %-30s kernel %8.2f us  %7.1f GB/s  call %8.2f us" %
              (n, "jit oscillation n=3 m=2, numeric", k, n * 24 / (k * 1e-6) / 1e9, wall), flush=True)


    # the path's own model written as source (setup + residual + Jacobian), against the built-in sweep
    import torch
    from bench import make_shard_on_gpu
    from tests import datasets as ds
    from tests.test_gpu_parity import P2P_JIT_SETUP, P2P_JIT_RESIDUAL, P2P_JIT_JACOBIAN
    for n in args.n:
        src, tgt = make_shard_on_gpu(torch, n, 0, torch.float64)
        planes = torch.cat([src.T, tgt.T]).contiguous().cpu().numpy()
        jit = mo.JitModelCost(6, 3, P2P_JIT_RESIDUAL, P2P_JIT_JACOBIAN, planes=planes, n_aux=12,
                              setup_body=P2P_JIT_SETUP)
        for name, mode in (("jit point2point, analytic", 0), ("jit point2point, numeric", 2)):
            k, wall = timed(jit, ds.X_GENERIC, mode, args.iters)
            print("n=%9d %-30s kernel %8.2f us  %7.1f GB/s  call %8.2f us" %
                  (n, name, k, n * 48 / (k * 1e-6) / 1e9, wall), flush=True)


if __name__ == "__main__":
    main()
