"""What does the pause before a timed region cost the steps after it?

bench.py brackets its K timed steps with barrier + torch.cuda.synchronize() (the driver's contract).
Round 5's per-step stamps showed the steps right after that bracket running slower than the same steps
a few hundred calls later (config 5 at K = 20: ~40 us per step against 23 us at K = 200).  This probe
runs one long sequence of blocking steps and, every `period` steps, does one thing between two steps —
nothing | torch.cuda.synchronize() | hipStreamSynchronize of the cost's stream | a host sleep of
20 us / 200 us / 2 ms — then prints the mean of step +1, +2..+5, +6..+20, +21..+60 after the event against the
steady state.

    python scripts/probe_sync_effect.py [p2p10m|p2p1m|camera] ...
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import moptimizer_0_amd as mo  # noqa: E402
from tests import datasets as ds  # noqa: E402

import bench  # noqa: E402


def busy_sleep(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


def make_step(kind):
    if kind == "camera":
        n, costs, step, _ = bench.camera_problem(mo, ds)
        mo.capi.link_costs(costs)
        return step, costs[0], costs
    n = 10_000_000 if kind == "p2p10m" else 1_000_000
    src, tgt = bench.make_shard_on_gpu(torch, n, 0, torch.float64)
    torch.cuda.synchronize()
    cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), device=0, dtype=np.float64, device_ptrs=True,
                              count=n)
    call, x_in, H, b, s = cost.bound_linearize(mo.JAC_ANALYTIC)
    xs = [ds.X_GENERIC + 1e-4 * k for k in range(16)]

    def step(k):
        x_in[:] = xs[k % 16]
        call()
        return H, b, s[0]
    return step, cost, [cost]


def main():
    bench.quiesce_python_gc()
    kinds = sys.argv[1:] or ["p2p10m", "p2p1m", "camera"]
    clock = time.perf_counter
    for kind in kinds:
        step, first, keep = make_step(kind)
        for k in range(3000 if kind != "p2p10m" else 1500):
            step(k)
        events = [("nothing", lambda: None),
                  ("torch.cuda.synchronize()", torch.cuda.synchronize),
                  ("hipStreamSynchronize(cost stream)", first.synchronize),
                  ("host spins 20 us", lambda: busy_sleep(20)),
                  ("host spins 200 us", lambda: busy_sleep(200)),
                  ("host spins 2 ms", lambda: busy_sleep(2000)),
                  ("time.sleep(2 ms)", lambda: time.sleep(0.002)),
                  ("host spins 20 ms", lambda: busy_sleep(20000))]
        def both(a, b):
            return lambda: (a(), b())
        if os.environ.get("PROBE_SET", "1") == "2":
            events = [("nothing", lambda: None),
                      ("sync", torch.cuda.synchronize),
                      ("sync, sync", both(torch.cuda.synchronize, torch.cuda.synchronize)),
                      ("sync, host spins 200 us", both(torch.cuda.synchronize, lambda: busy_sleep(200))),
                      ("sync, host spins 2 ms", both(torch.cuda.synchronize, lambda: busy_sleep(2000))),
                      ("sync, host spins 10 ms", both(torch.cuda.synchronize, lambda: busy_sleep(10000))),
                      ("sync, time.sleep(10 ms)", both(torch.cuda.synchronize, lambda: time.sleep(0.01)))]
        if os.environ.get("PROBE_SET", "1") == "3":
            events = [("nothing", lambda: None), ("sync", torch.cuda.synchronize)]
        period, reps = int(os.environ.get("PROBE_PERIOD", "150")), 6
        print("== %s: steady state and the steps after an event (us per step; mean over %d repetitions)"
              % (kind, reps), flush=True)
        for name, fn in events:
            after = np.zeros((reps, period))
            for r in range(reps):
                fn()
                t = clock()
                for k in range(period):
                    step(k)
                    t2 = clock()
                    after[r, k] = (t2 - t) * 1e6
                    t = t2
            m = after.mean(axis=0)
            print("  %-36s +1: %6.1f  +2..5: %6.1f  +6..20: %6.1f  +21..60: %6.1f  +61..150: %6.1f   (all: median %6.1f mean %6.1f)"
                  % (name, m[0], m[1:5].mean(), m[5:20].mean(), m[20:60].mean(), m[60:150].mean(),
                     float(np.median(after)), float(after.mean())), flush=True)
        if kind == "camera":
            mo.capi.link_costs([])
        for c in keep:
            c.close()


if __name__ == "__main__":
    main()
