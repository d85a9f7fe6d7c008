#!/usr/bin/env python3
"""End-to-end Levenberg-Marquardt solve (the caller of the path, levenberg_marquadt_dyn.cpp:34-119)
over the HIP cost, with and without speculative linearization, next to the CPU restatement.
    python tests/tools/lm_demo.py [--n 10000000] [--cpu-n 1000000]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def lm(linearize, compute_cost, x0, max_iter=50, lm_iter=3):
    x = np.array(x0, dtype=np.float64); lam = -1.0; eps = np.finfo(np.float64).eps
    for it in range(max_iter):
        H, b, y0 = linearize(x)
        if abs(y0) < 8 * eps: return x, it
        D = np.diag(np.diag(H))
        if lam < 0: lam = 1e-9 * np.abs(np.diag(H)).max()
        nu = 2.0
        for k in range(lm_iter):
            delta = np.linalg.solve(H + lam * D, -b); xi = x + delta; yi = compute_cost(xi)
            rho = (y0 - yi) / delta.dot(lam * delta - b)
            if rho < 0:
                if np.abs(delta).max() < np.sqrt(eps): return x, it
                lam *= nu; nu *= 2; continue
            x = xi; lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3); break
    return x, max_iter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--cpu-n", type=int, default=1_000_000)
    ap.add_argument("--mode", type=int, default=2, help="0 analytic, 2 numeric (as tst/point2point.cpp:192-217)")
    args = ap.parse_args()
    import torch
    import moptimizer_0_amd as mo
    from bench import make_shard_on_gpu, quiesce_python_gc
    from tests import datasets as ds, oracle_binding as ob
    torch.cuda.set_device(0)
    quiesce_python_gc()
    src, tgt = make_shard_on_gpu(torch, args.n, 0, torch.float64)
    torch.cuda.synchronize()
    for spec in (False, True):
        cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), device_ptrs=True, count=args.n)
        cost.set_speculation(spec)
        t0 = time.perf_counter()
        x, iters = lm(lambda x: cost.linearize(x, args.mode), cost.compute_cost, np.zeros(6))
        dt = time.perf_counter() - t0
        sweeps, hits = cost.stats()
        print("GPU  n=%d speculation=%-5s: %2d outer iterations, %2d sweeps (+%d answered from the kept "
              "linearization), %.2f ms, |x - truth| = %.1e" %
              (args.n, spec, iters, sweeps, hits, dt * 1e3, np.abs(x - ds.FIXTURE_X).max()), flush=True)
        cost.close()
    m = min(args.cpu_n, args.n)
    sh, th = src[:m].cpu().numpy(), tgt[:m].cpu().numpy()
    o = ob.load()
    t0 = time.perf_counter()
    xc, st, itc = o.p2p_minimize(sh, th, np.zeros(6), cost_class=ob.NUMERIC_DYN if args.mode == 2 else ob.ANALYTIC_DYN,
                                 layout=ob.LAYOUT_ROW_MAJOR, max_iter=50)
    dt = time.perf_counter() - t0
    print("CPU  n=%d (restatement; linearize 1 thread, cost sweep all cores): %d outer iterations, %.1f ms "
          "(= %.1f ms per million correspondences), |x - truth| = %.1e" %
          (m, itc, dt * 1e3, dt * 1e3 / (m / 1e6), np.abs(xc - ds.FIXTURE_X).max()))


if __name__ == "__main__":
    main()
