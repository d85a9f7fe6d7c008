#!/usr/bin/env python3
"""Benchmark of the linearization hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one LM linearization sweep as the optimizer sees it: for the current parameter
vector x, every rank runs the point-to-point analytic linearize kernels over its resident shard,
the 43 partial sums (H 6x6 | b | cost) are combined with one RCCL all-reduce when N > 1, and the
result is brought to the host (LM needs it there before it can choose the next x).  x changes
every step.  Inputs are synthetic (seed 42) and already resident in HBM when timing starts.

Workload: BASELINE.json quotes its metric at N = 10 M correspondences, which fits one GPU, so
each GPU holds 10 M correspondences (480 MB fp64, larger than the 256 MiB Infinity Cache so the
sweep really streams from HBM).  Per-GPU work is fixed as N grows ("weak" scaling).

`python bench.py --gpus N` with N > 1 and no launcher starts its own N rank processes (fresh
children, before anything touches the GPU) and relays rank 0's line; under torch.distributed.run
(WORLD_SIZE set) it is one of the ranks.  MOPT_BENCH_BACKEND=gloo rehearses N ranks on fewer GPUs.

Timed region: settle | W warm-up steps | barrier + torch.cuda.synchronize() | t0 | K blocking steps |
barrier + synchronize | t1.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  timing       — the K individual step times of rank 0 (per_step_us, median, min, first)
  check.vs_oracle — one GPU linearize over rank 0's shard against the CPU restatement's H, b, cost
                 on the same correspondences (the sums the cpu_baseline sweeps compute anyway), bar 1e-6
  check.timed_region — what the library counted between t0 and t1 (mopt_cost_stats /
                 mopt_cost_direct_dispatches deltas): K sweeps, 0 calls answered from a kept result,
                 K direct dispatches (0 on HIP streams) — K steps were K sweeps over HBM
  configs      — N = 1: BASELINE configs 1, 2, 3 (default and literal evaluation) and 5, timed the same
                 way, plus the two figures the 256 MiB Infinity Cache cannot have helped:
                 hbm_check (the same analytic sweep over 100 M correspondences = 4.8 GB, 19 x the
                 cache) and rotating (several distinct costs of the headline's size swept round-robin:
                 every line a sweep reads was last touched >= 3 x 480 MB ago); roofline.frac_rotating
                 and roofline.hbm_check_frac repeat their fractions beside roofline.frac
  roofline     — achieved algorithmic GB/s of the dominant (sweep) kernel from HIP events that
                 carry the dispatch's own timestamps, taken in a pass of their own AFTER the K
                 timed wall-clock steps (which run uninstrumented), vs the 8 TB/s HBM3E peak
  cpu_baseline — the CPU restatement of the reference's single-threaded linearize
                 (oracle/, kind "port") timed on rank 0's host cores on a bounded sample, at
                 every world size
  rccl         — N > 1: the same K steps with the sums added by ncclAllReduce over the
                 communicator attached to the cost; "ranks" is what ncclCommCount reports
  by_collective — N > 1: step time, whole-job rate and fraction of the HBM roof AS A STEP SEES IT
                 for every transport measured (config.collective names the headline's)

Order for N > 1: attach every transport (host slots, peer slots, RCCL) | headline pass | kernel-
time pass | CPU baseline — from there the line can stand — | RCCL pass | the comparison passes
(other transports, no combine, the 10 M strong-scaling split).  A watchdog covers it all: firing
before the line can stand it ends the rank with exit code 3 and no line; after, it prints the line
as it is and ends the rank.  Two top-level booleans, always present for N > 1, say at a glance whether
the line is whole: "rccl_incomplete" (the RCCL timed pass was started and did not finish) and
"extras_incomplete" (the comparison passes did not all finish; "note" names the stage).  A transport
asked for with --collective that cannot be attached (RCCL between ranks that share a GPU) falls back to
the automatic order; config.collective_requested / collective_fell_back record it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_CORRESPONDENCE = {8: 48, 4: 24}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=10_000_000, help="correspondences PER GPU")
    ap.add_argument("--total-n", type=int, default=0,
                    help="strong scaling: shard this many correspondences over the GPUs "
                         "(BASELINE config 4: 10000000 over 8); overrides --n")
    ap.add_argument("--workload", choices=["point2point", "camera"], default="point2point",
                    help="camera = BASELINE config 5: reprojection cost, 100k elements as two costs "
                         "(40k + 60k) with Geman-McClure, numeric Jacobian, 1 GPU")
    ap.add_argument("--mode", choices=["analytic", "analytic_tst", "numeric"], default="analytic")
    ap.add_argument("--variant", choices=["auto", "literal", "moments"], default="auto")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--cov", choices=["identity", "symmetric", "general"], default="identity",
                    help="covariance of the cost (CostFunctionBase::setCovariance): the reference's "
                         "default identity, or a fixed symmetric / non-symmetric 3x3 one")
    ap.add_argument("--loss", choices=["none", "gm"], default="none",
                    help="gm = loss::GemmanMCClure(100)")
    ap.add_argument("--collective", choices=["auto", "host", "peer", "rccl", "torch"], default="auto",
                    help="N>1, how the 43 sums are added over the ranks: host = every finalize kernel "
                         "publishes into one shared pinned host block, the host adds (one hop, no "
                         "collective launch); peer = finalize kernels push into each other's HBM "
                         "slots over xGMI/IPC and add on the device; rccl = ncclAllReduce on the "
                         "cost's stream; torch = torch.distributed.all_reduce on the async result; "
                         "auto = the first of host, peer, rccl, torch that every rank could attach")
    ap.add_argument("--no-compare", action="store_true",
                    help="N>1: skip the extra passes that time every other attached transport, the "
                         "sweep without any combine, and the strong-scaling (config 4) split")
    ap.add_argument("--kernel-steps", type=int, default=0,
                    help="steps of the separate pass that times the sweep kernel with HIP events "
                         "(0 = min(steps, 100); the K wall-clock steps run with profiling off)")
    ap.add_argument("--settle-ms", type=float, default=500.0,
                    help="untimed sweeps before the warm-up steps, about this many milliseconds")
    ap.add_argument("--pause-after-sync-ms", type=float, default=0.0,
                    help="host spin between the barrier + synchronise that precedes the timed steps and "
                         "t0 (the runtime releases its queued commands after a synchronisation; 0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="N=1: skip the configs block (BASELINE configs 2, 3 and 5 timed after the headline)")
    ap.add_argument("--hbm-check-n", type=int, default=100_000_000,
                    help="N=1: correspondences of configs.hbm_check, the sweep whose input (48 B each) is "
                         "many times the 256 MiB Infinity Cache (0 = skip)")
    ap.add_argument("--rotating-costs", type=int, default=4,
                    help="N=1: distinct costs of the headline's size swept round-robin for "
                         "roofline.frac_rotating (0 = skip)")
    ap.add_argument("--rotating-launches", type=int, default=0,
                    help="timed kernel launches per rotating cost (0 = min(kernel steps, 30))")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--deadline-s", type=float,
                    default=float(os.environ.get("MOPT_BENCH_DEADLINE_S", "330")),
                    help="N>1: seconds after the ranks have their data by which the run ends "
                         "whatever has or has not finished (watchdog)")
    return ap.parse_args()


def make_shard_on_gpu(torch, n, rank, dtype):
    """src ~ U[0,10]^3, tgt = R src + t + N(0, 0.01^2) with the fixture pose of
    tst/point2point.cpp:93-101; generated on the device so nothing crosses PCIe."""
    from tests import datasets as ds
    g = torch.Generator(device="cuda")
    g.manual_seed(42 + rank)
    src = torch.rand((n, 3), generator=g, device="cuda", dtype=torch.float64) * 10.0
    R = torch.tensor(ds.fixture_rotation(), device="cuda", dtype=torch.float64)
    t = torch.tensor(ds.FIXTURE_T, device="cuda", dtype=torch.float64)
    tgt = src @ R.T + t + 0.01 * torch.randn((n, 3), generator=g, device="cuda",
                                              dtype=torch.float64)
    return src.to(dtype).contiguous(), tgt.to(dtype).contiguous()


def quiesce_python_gc():
    """Keep the interpreter's cyclic garbage collector out of the timed region.  A full collection
    over the ~170 k objects that `import torch` leaves behind takes ~36 ms and is triggered by the
    small arrays each step allocates (measured: always the 335th blocking call of a fresh process,
    and never with the collector frozen - `scripts/probe_stall.py`; a C++ caller does not see it).
    Everything alive now is moved to the permanent generation; the collector stays enabled."""
    import gc
    gc.collect()
    gc.freeze()


def cpu_baseline(src_host, tgt_host, x, jac_mode, target_seconds):
    """Single-threaded restatement of the reference's linearize (the reference loop is
    single-threaded: linearization.h:97,142) on a bounded prefix of the same workload.  Returns the
    baseline record and the oracle's own (H, b, sum) at x over that prefix — the timed sweeps compute
    them anyway — for the line's check.vs_oracle."""
    from tests import oracle_binding as ob
    oracle = ob.load()
    cost_class = ob.NUMERIC_DYN if jac_mode == 2 else ob.ANALYTIC_DYN
    layout = ob.LAYOUT_TST if jac_mode == 1 else ob.LAYOUT_ROW_MAJOR
    sample = src_host.shape[0]
    sweeps, dt = 0, 0.0
    sums = None
    t0 = time.perf_counter()
    while dt < target_seconds and sweeps < 1000:
        sums = oracle.p2p_linearize(src_host, tgt_host, x, cost_class=cost_class, layout=layout)
        sweeps += 1
        dt = time.perf_counter() - t0
    out = {
        "value": sample * sweeps / dt,
        "unit": "correspondences/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d sweeps over the first %d correspondences of rank 0's shard, %.1f s of CPU "
                  "work (single thread, as the reference's linearize loop)" % (sweeps, sample, dt),
        # the reference builds -O3 -march=native (CMakeLists.txt:19-22); the checker's objects travel
        # to a host other than the one that compiled them, hence a named micro-architecture level
        "compiler": oracle.build_flags(),
    }
    # for orientation only: the same sweep split over every host core (the reference parallelises
    # only its cost-only sweep, linearization.h:52); the median of three, since a 256-thread sweep
    # of a few tens of milliseconds swings with thread start-up
    cores = os.cpu_count() or 1
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.p2p_linearize(src_host, tgt_host, x, cost_class=cost_class, layout=layout, threads=cores)
        rates.append(sample / (time.perf_counter() - t0))
    out["all_cores"] = {"value": sorted(rates)[1], "cores": cores, "of": "median of 3 sweeps",
                        "min": min(rates), "max": max(rates)}
    return out, sums


def vs_oracle(got, want, bar):
    """check.vs_oracle: norm-wise distance of the GPU's H, b (max |d| / max |want|) and of the cost
    from the CPU restatement's on the same correspondences at the same x."""
    H, b, c = got
    Hr, br, cr = want
    Hr, br = np.asarray(Hr, dtype=np.float64), np.asarray(br, dtype=np.float64)
    h_rel = float(np.abs(np.asarray(H, dtype=np.float64) - Hr).max() / np.abs(Hr).max())
    b_rel = float(np.abs(np.asarray(b, dtype=np.float64) - br).max() / np.abs(br).max())
    c_rel = float(abs(float(c) - float(cr)) / abs(float(cr)))
    return {"H_rel": h_rel, "b_rel": b_rel, "cost_rel": c_rel, "bar": bar,
            "ok": bool(max(h_rel, b_rel, c_rel) <= bar)}


def camera_problem(mo, ds):
    """BASELINE config 5: 100 000 reprojection elements as two costs (40 k + 60 k, the split of
    tst/multiple_objectives.cpp:110-117), Geman-McClure(100) on each (tst/loss_function.cpp:31-32),
    forward differences.  Returns the costs and step(k): the optimizer's loop over its costs
    (levenberg_marquadt_dyn.cpp:48-60) through pre-bound calls and reused buffers, as the point2point
    workload is measured: what is timed is the library, not allocations."""
    n, split = 100_000, 40_000
    pts, pix = ds.synthetic_camera(n, seed=17)
    costs = [mo.ReprojectionCost(pts[:split], pix[:split]), mo.ReprojectionCost(pts[split:], pix[split:])]
    for c in costs:
        c.set_loss(mo.LOSS_GEMAN_MCCLURE, 100.0)
    x = np.zeros(6)
    bound = [c.bound_linearize(mo.JAC_NUMERIC) for c in costs]
    H, b = np.zeros((6, 6), order="F"), np.zeros(6)

    def step(k):
        xs = x + 1e-4 * (k % 16)
        H.fill(0.0)
        b.fill(0.0)
        y = 0.0
        for call, x_in, Hc, bc, sc in bound:
            x_in[:] = xs
            call()  # blocking C-ABI call: kernels, the 43 results on the host
            np.add(H, Hc, out=H)
            np.add(b, bc, out=b)
            y += sc[0]
        return H, b, y

    def kernel_ms(steps):
        """both costs' sweep kernels per step, from a pass of its own (a profiled launch carries two events)"""
        for c in costs:
            c.set_profiling(True)
        for k in range(steps):
            step(k)
        prof = [c.profile() for c in costs]
        for c in costs:
            c.set_profiling(False)
        return sum(p[0] for p in prof) / max(prof[0][1], 1)

    return n, costs, step, kernel_ms


def camera_main(args):
    """BASELINE config 5 (latency-dominated: 4 MB of input): one step = the multi-objective
    linearization of levenberg_marquadt_dyn.cpp:48-60 — linearize both costs, add H, b, cost on the
    host."""
    import moptimizer_0_amd as mo
    from tests import datasets as ds
    quiesce_python_gc()
    n, costs, step, kernel_pass_ms = camera_problem(mo, ds)

    def timed():
        for k in range(args.warmup):
            step(k)
        t0 = time.perf_counter()
        for k in range(args.steps):
            out = step(k)
        return time.perf_counter() - t0, out

    elapsed_unlinked, _ = timed()
    # the costs of one problem, as an optimizer holds them: the first one asked at an x queues the
    # other's sweep too (mopt_costs_link); the loop above is unchanged
    mo.capi.link_costs(costs)
    elapsed, (H, b, y) = timed()
    mo.capi.link_costs([])
    kernel_ms = kernel_pass_ms(args.steps)
    achieved = n * 40 / (kernel_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "point-correspondences/sec per LM linearization sweep; % HBM peak",
        "value": n * args.steps / elapsed, "unit": "residual-blocks/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "camera-calibration reprojection cost (config 5): 100000 elements as "
                               "two costs (40k + 60k), Geman-McClure(100), forward differences, "
                               "both linearized and summed per step; the two costs linked "
                               "(mopt_costs_link)"},
        "ms_per_step_unlinked": elapsed_unlinked / args.steps * 1e3,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kernel_ms,
                     "note": "4 MB of input: launch-latency-bound, not bandwidth-bound"},
        "check": {"sum_sq": float(y)}, "cpu_baseline": None}), flush=True)


def ensure_built():
    """The HIP library is a build product kept out of the git history; a tree without it is built
    here (by rank 0) rather than benchmarked through anything else."""
    lib = os.path.join(ROOT, "moptimizer_0_amd", "lib", "libmoptimizer_hip.so")
    if os.path.exists(lib):
        return
    if int(os.environ.get("RANK", "0")) == 0:
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "-j4", "all"], stdout=sys.stderr)
    else:
        for _ in range(600):
            if os.path.exists(lib):
                break
            time.sleep(1.0)
        time.sleep(2.0)  # let the linker finish writing


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script
    (the parent has not imported torch or touched HIP, and never replaces itself), give each its
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, relay rank 0's JSON line, and exit with the first
    non-zero rank exit code."""
    import socket
    import subprocess
    ensure_built()  # once, before the ranks race for it
    env_base = dict(os.environ)
    env_base.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env_base:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            env_base["MASTER_PORT"] = str(sock.getsockname()[1])
    env_base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL, slot blocks)
    procs = []
    for r in range(args.gpus):
        env = dict(env_base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines[-1:]:
            print(ln, file=sys.stderr)
    if json_lines:
        print(json_lines[-1], flush=True)
    bad = [c for c in codes if c != 0]
    if bad or not json_lines:
        raise SystemExit(bad[0] if bad else 1)


def refuse_more_processes_than_the_gpu_admits(world):
    """A rehearsal (MOPT_BENCH_BACKEND=gloo: ranks share GPUs) on a machine that limits the processes per
    GPU — this pool ends a run that puts more than 6 on one, the caller included — is refused HERE, by name,
    before any rank exists, when MOPT_MAX_PROCESSES_PER_GPU says what the limit is: eight rank processes on
    one GPU are not a time-out or a killed run but exit code 4 and a sentence.  (Ranks as threads reach
    world size 8 on such a machine: tests/test_gpu_multirank.py test_eight_ranks_combine_through_the_library.)"""
    limit = int(os.environ.get("MOPT_MAX_PROCESSES_PER_GPU", "0"))
    if limit <= 0 or os.environ.get("MOPT_BENCH_BACKEND", "nccl") == "nccl":
        return
    import subprocess
    try:  # (counting devices does not initialise the GPU in this process)
        ndev = int(subprocess.check_output([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                           stderr=subprocess.DEVNULL).split()[-1])
    except Exception:  # noqa: BLE001
        ndev = 1
    per_gpu = -(-world // max(ndev, 1))
    if per_gpu + 1 > limit:
        print(json.dumps({"error": "bench.py: %d ranks on %d GPU(s) are %d rank processes on one GPU, %d with "
                                   "their caller; MOPT_MAX_PROCESSES_PER_GPU=%d (this machine's limit on processes "
                                   "per GPU) admits %d ranks per GPU — not started" %
                                   (world, ndev, per_gpu, per_gpu + 1, limit, limit - 1),
                          "resource": "processes per GPU", "ranks": world, "gpus": ndev, "limit": limit}),
              file=sys.stderr, flush=True)
        raise SystemExit(4)


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        refuse_more_processes_than_the_gpu_admits(args.gpus)
        return spawn_ranks(args)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch one rank per GPU "
                         "(torch.distributed.run --nproc-per-node %d), or run without a launcher"
                         % (world, args.gpus, args.gpus))
    ensure_built()
    if args.workload == "camera":
        return camera_main(args)
    import torch
    import torch.distributed as dist

    import moptimizer_0_amd as mo
    from moptimizer_0_amd.sharded import attach_combines, gpu_point2point_sweep
    from tests import datasets as ds

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MOPT_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks: ranks share
    # devices and torch.distributed (barriers, handle exchange, max-over-ranks) runs over gloo.  The
    # host / peer combines work between ranks that share a GPU; RCCL refuses to.  Everything else —
    # shards, barriers, timing, the JSON line — is the real code path.
    backend = os.environ.get("MOPT_BENCH_BACKEND", "nccl")
    if world > 1 and backend != "nccl":
        # ranks sharing a GPU: no extra hardware queues per rank (the library's direct-dispatch queues;
        # with one rank per GPU — the real thing — sharded costs use them like any other)
        os.environ.setdefault("MOPT_AQL_SHARDED", "0")
    if not torch.cuda.is_available():
        raise SystemExit("rank %d of %d: no HIP device is visible — bench.py measures the HIP path "
                         "and has no CPU substitute for it" % (rank, world))
    ndev = max(torch.cuda.device_count(), 1)
    if world > ndev and backend == "nccl":
        raise SystemExit("%d ranks but %d GPUs: one rank per GPU (MOPT_BENCH_BACKEND=gloo rehearses "
                         "more ranks than GPUs)" % (world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    ctl = "cuda" if (world > 1 and backend == "nccl") else "cpu"  # where control tensors live

    def log(msg):
        print(msg, file=sys.stderr, flush=True)

    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    t_dtype = torch.float64 if args.dtype == "f64" else torch.float32
    scalar_bytes = np.dtype(np_dtype).itemsize
    jac_mode = {"analytic": mo.JAC_ANALYTIC, "analytic_tst": mo.JAC_ANALYTIC_TST_LAYOUT,
                "numeric": mo.JAC_NUMERIC}[args.mode]
    variant = {"auto": mo.KERNEL_AUTO, "literal": mo.KERNEL_LITERAL,
               "moments": mo.KERNEL_MOMENTS}[args.variant]
    quiesce_python_gc()

    def make_cost(n):
        src, tgt = make_shard_on_gpu(torch, n, rank, t_dtype)
        torch.cuda.synchronize()
        cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), device=local_rank, dtype=np_dtype,
                                  device_ptrs=True, count=n)
        cost.set_kernel_variant(variant)
        if args.cov != "identity":
            cost.set_covariance({"symmetric": [[2.0, 0.3, -0.1], [0.3, 1.5, 0.2], [-0.1, 0.2, 0.8]],
                                 "general": [[2.0, 0.5, -0.1], [0.3, 1.5, 0.4], [-0.3, 0.2, 0.8]]}[args.cov])
        if args.loss == "gm":
            cost.set_loss(mo.LOSS_GEMAN_MCCLURE, 100.0)
        return cost, src, tgt

    if args.total_n:
        lo, hi = args.total_n * rank // world, args.total_n * (rank + 1) // world
        args.n = hi - lo
    cost, src, tgt = make_cost(args.n)
    keep_host = (rank == 0 and not args.no_cpu_baseline)
    if keep_host:
        head = min(args.n, 10_000_000)
        src_host = src[:head].double().cpu().numpy()
        tgt_host = tgt[:head].double().cpu().numpy()
    del src, tgt
    torch.cuda.empty_cache()

    # ---- watchdog: this is the first time more than one GPU runs these paths -------------------
    # Every wait below is bounded (MOPT_PEER_TIMEOUT_MS inside the combines, 60 s in the blocking
    # call), but torch.distributed's own collectives are not on a time scale that helps a driver
    # with a 10-minute limit.  One timer per rank, started before the first pass.
    import threading
    line = {}
    progress = {"stage": "attaching the combine transports", "complete": False}
    emitted = threading.Lock()

    def serialized(extra):
        """The line as JSON, taken while the main thread may still be adding to it (the watchdog's
        case): a dict that changes size under json.dumps raises, so copy and retry."""
        import copy
        for _ in range(200):
            try:
                snap = copy.deepcopy(line)
                snap.update(extra)
                return json.dumps(snap)
            except RuntimeError:
                time.sleep(0.001)
        return json.dumps(dict(extra, error="bench.py: the line kept changing under the watchdog"))

    def emit(final):
        if not emitted.acquire(blocking=False):
            return
        if rank == 0:
            extra = {}
            if not final:
                # (rccl_incomplete is in the line already: True from the start of the RCCL pass to its end)
                extra = {"extras_incomplete": True,
                         "note": "watchdog: still in %r at the deadline; reported without it"
                                 % progress["stage"]}
            print(serialized(extra), flush=True)

    def watchdog():
        # whatever happens in here, this rank ends: a watchdog that dies of an exception would leave
        # the rank with neither a line nor an exit
        code = 3
        try:
            if progress["complete"]:
                log("rank %d: watchdog fired in %r: reporting the measurement without it"
                    % (rank, progress["stage"]))
                emit(False)
                code = 0
            else:
                print(json.dumps({"error": "bench.py watchdog: no complete measurement by the deadline",
                                  "rank": rank, "stage": progress["stage"]}), file=sys.stderr, flush=True)
        finally:
            os._exit(code)

    dog = None
    if world > 1:
        dog = threading.Timer(args.deadline_s, watchdog)
        dog.daemon = True
        dog.start()

    # N > 1: the 43 sums of every sweep are added over the ranks.  Every transport is attached up
    # front — RCCL included: north_star names the all-reduce over xGMI, so its figure is part of
    # the line whatever the headline transport is.  A transport counts only if EVERY rank attached
    # it (attach_combines); it leaves the cost on MOPT_COMBINE_NONE.
    usable, collective, rccl_note = [], "none", None
    if world > 1:
        usable = attach_combines(cost, rank, world, want=("host", "peer", "rccl"), log=log)
        if "rccl" not in usable:
            rccl_note = ("ranks share GPUs (rehearsal backend %r): RCCL refuses duplicate devices"
                         % backend) if backend != "nccl" else "ncclCommInitRank failed on some rank"
        # a transport asked for by name comes first; one that could not be attached (RCCL between ranks
        # that share a GPU, for one) falls back to the automatic order, and the line says so
        auto = ("host", "peer", "rccl")
        prefer = auto if args.collective == "auto" else (args.collective,) + tuple(
            c for c in auto if c != args.collective)
        collective = next((c for c in prefer if c in usable), "torch")
    modes = {"none": mo.COMBINE_NONE, "rccl": mo.COMBINE_RCCL, "host": mo.COMBINE_HOST,
             "peer": mo.COMBINE_PEER}

    x_base = ds.X_GENERIC.astype(np_dtype)
    xs = [x_base + np_dtype(1e-4) * np_dtype(k) for k in range(16)]  # LM moves x every iteration

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def settle_steps(n_per_rank, ms):
        # n_per_rank must be the same number on every rank (shards of an indivisible total differ by
        # one correspondence): every settling step holds a combine, so the counts have to agree
        est_step_s = 20e-6 + n_per_rank * BYTES_PER_CORRESPONDENCE[scalar_bytes] / 6.0e12
        return min(20000, max(50, int(ms * 1e-3 / est_step_s)))

    def counters(costs):
        """(sweeps launched, calls answered from the kept result, sweeps dispatched through the
        library's own AQL packets) summed over `costs`, as the library counts them."""
        sw = hits = direct = 0
        for c in costs:
            a, h = c.stats()
            sw, hits, direct = sw + a, hits + h, direct + c.direct_dispatches()
        return sw, hits, direct

    region = {}  # the library's counters over the most recent timed region (timed_pass fills it)

    def timed_pass(the_cost, combine, steps, warmup, settle, via_torch=None, mode=None, step_fn=None,
                   counted=None):
        """barrier | K blocking steps | barrier, wall clock, max over ranks.  Profiling is off.
        Returns (elapsed, H, sum, stamps): stamps[k] is this rank's clock after step k - 1
        (stamps[0] = t0), read between the steps — one clock read each, ~50 ns.
        A step that fails (a peer that never delivers ends in MOPT_ERR_PEER_TIMEOUT, not in a
        hang) does not take this rank out of the sequence of collectives: it stops stepping, goes
        through both barriers and the max-reduce like everybody else, and raises afterwards."""
        err = None
        call = x_in = H_out = b_out = s_out = None
        mode = jac_mode if mode is None else mode
        try:
            if step_fn is None:
                if combine is not None:
                    the_cost.set_combine(modes[combine])
                call, x_in, H_out, b_out, s_out = the_cost.bound_linearize(mode)
        except Exception as e:  # noqa: BLE001
            err = e

        def step(k):
            if step_fn is not None:
                return step_fn(k)
            if via_torch is not None:
                return via_torch.linearize(xs[k % 16], mode)
            x_in[:] = xs[k % 16]
            call()  # blocking C-ABI call: kernels, the sum over the ranks, the 43 results on the host
            return H_out, b_out, s_out[0]

        H = s = None
        if err is None:
            try:
                for k in range(settle + warmup):
                    step(k)
            except Exception as e:  # noqa: BLE001
                err = e
        stamps = [0.0] * (steps + 1)
        clock = time.perf_counter
        counted = ([the_cost] if the_cost is not None else []) if counted is None else counted
        before = counters(counted)  # nothing is launched between this reading and t0
        barrier()
        # The synchronisation that has to precede t0 is not free for the steps that follow it: it hands
        # the HIP runtime a marker, and when that completes a thread of the runtime releases every command
        # queued since the previous marker — with a thousand commands in the batch the launching thread
        # runs 5-15 us per call slower for the next 0.3-1 ms (scripts/probe_sync_effect.py,
        # profiles/r5_sync_effect.txt).  The library now bounds those batches itself (a stream query every
        # 32 blocking sweeps, c_abi.cpp boundCommandBatch), which takes the transient out of the driver's
        # 20 steps (profiles/r5_marker_bench_ab.txt); the optional host spin here (GPU idle, nothing
        # queued) did the same and costs the first launch after an idle GPU +10 us — off by default.
        until = clock() + args.pause_after_sync_ms * 1e-3
        while clock() < until:
            pass
        t0 = stamps[0] = clock()
        if err is None:
            try:
                for k in range(steps):
                    H, b, s = step(k)
                    stamps[k + 1] = clock()
            except Exception as e:  # noqa: BLE001
                err = e
        barrier()
        elapsed = time.perf_counter() - t0
        after = counters(counted)
        region.clear()
        region.update({"steps": steps, "costs": len(counted), "sweeps": after[0] - before[0],
                       "cache_hits": after[1] - before[1], "direct_dispatches": after[2] - before[2]})
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        if err is not None:
            raise err
        return elapsed, np.array(H, dtype=np.float64), float(s), stamps

    def step_times(stamps, elapsed_s=None):
        """timing block of a pass: the K individual step times of this rank, in microseconds."""
        us = [(b - a) * 1e6 for a, b in zip(stamps[:-1], stamps[1:])]
        srt = sorted(us)
        closing = None if elapsed_s is None else elapsed_s * 1e6 - (stamps[-1] - stamps[0]) * 1e6
        return {"per_step_us": [round(v, 2) for v in us[:400]], "median": srt[len(srt) // 2],
                "min": srt[0], "max": srt[-1], "mean_of_steps": sum(us) / len(us),
                # the barrier + torch.cuda.synchronize() that closes the timed region (inside it, as the
                # contract has it): the runtime queues a marker behind the last kernel and waits for it
                "closing_barrier_us": closing,
                "first": us[0], "pause_after_sync_ms": args.pause_after_sync_ms,
                "what": "rank 0's clock read after every blocking step of the timed "
                "region; ms_per_step is the mean over the region, barriers included; the host spins "
                "pause_after_sync_ms between the synchronisation and t0 (GPU idle, nothing queued)"}

    # ---- the measurement: W warm-up steps, then exactly K timed steps, uninstrumented ----------
    # Untimed settling first: a GPU that has just been handed its data is not yet in its steady
    # state (measured at 10 M, same box: 87.0 us per step timed after 70 steps, 85.8 after 150 ms of
    # sweeps, 84.9 after 1 s), and the metric is the steady-state rate of an LM loop that runs
    # thousands of sweeps.  The count is the same on every rank (each step holds a collective).
    from moptimizer_0_amd.sharded import _all_agree

    def guarded_pass(the_cost, name, steps, warmup, settle):
        """A timed pass that every rank either completes or abandons together: a transport that
        fails at run time on any rank is reported as unusable instead of ending the run."""
        progress["stage"] = "timed pass, combine %r" % name
        ok, err, res = True, None, None
        try:
            if name == "torch":
                the_cost.set_combine(mo.COMBINE_NONE)
                res = timed_pass(the_cost, None, steps, warmup, settle,
                                 via_torch=gpu_point2point_sweep(the_cost))
            else:
                res = timed_pass(the_cost, name, steps, warmup, settle)
        except Exception as e:  # noqa: BLE001 - whatever it was, the ranks must agree on it
            ok, err = False, e
        if world > 1 and not _all_agree(ok):
            if err is not None:
                log("rank %d: combine %r failed in the timed pass: %s" % (rank, name, err))
            return None
        return res

    bpc = BYTES_PER_CORRESPONDENCE[scalar_bytes]
    total = args.total_n if args.total_n else args.n * world

    def as_step_sees_it(ms, n_per_gpu=None, total_n=None):
        """Step time -> whole-job rate and the fraction of ONE GPU's HBM roof its shard's
        algorithmic bytes reach per step (kernel + finalize + combine + hand-over to the host)."""
        n_per_gpu = args.n if n_per_gpu is None else n_per_gpu
        total_n = total if total_n is None else total_n
        return {"ms_per_step": ms, "value": total_n / (ms * 1e-3),
                "step_frac": n_per_gpu * bpc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    cost.set_profiling(False)
    settle = settle_steps(total // world, args.settle_ms)
    if world == 1:
        elapsed, H, s, stamps = timed_pass(cost, None, args.steps, args.warmup, settle)
    else:
        order = [collective] + [c for c in usable + ["torch"] if c != collective]
        res = None
        for name in order:
            res = guarded_pass(cost, name, args.steps, args.warmup, settle)
            if res is not None:
                collective = name
                break
            usable = [c for c in usable if c != name]
        if res is None:
            raise SystemExit("rank %d: no way of adding the ranks' sums worked" % rank)
        elapsed, H, s, stamps = res
    headline_region = dict(region, what="mopt_cost_stats / mopt_cost_direct_dispatches deltas of the "
                           "headline's cost between t0 and t1 of the timed region: K steps are K sweeps "
                           "launched, none answered from the kept result; direct_dispatches = K when the "
                           "library's own AQL queue carried them, 0 on HIP streams")

    # ---- kernel time: a pass of its own, every launch carrying its dispatch timestamps ---------
    def kernel_pass(the_cost, steps, mode=None):
        the_cost.set_combine(mo.COMBINE_NONE)  # the kernel's duration does not involve the ranks
        call, x_in, _, _, _ = the_cost.bound_linearize(jac_mode if mode is None else mode)
        for k in range(5):
            x_in[:] = xs[k % 16]
            call()
        the_cost.set_profiling(1)
        for k in range(steps):
            x_in[:] = xs[k % 16]
            call()
        ms, launches = the_cost.profile()
        the_cost.set_profiling(False)
        return ms / max(launches, 1), launches

    def baseline_configs():
        """BASELINE.json configs 1, 2, 3 and 5, measured as the headline is (same timed_pass /
        kernel_pass): 1 M correspondences fit the Infinity Cache, 100 k reprojection elements are
        pure latency — step times, with the kernel's share beside them."""
        out = {}
        # configs[0]: tst/point2point.cpp's size.  One blocking linearization per step as everywhere
        # else, and — what a 1 k problem is for — the whole registration (forward differences, from
        # x = 0) under mopt_lm_minimize
        n0 = 1000
        c0, s0, t0 = make_cost(n0)
        del s0, t0
        el, _, ssq, st = timed_pass(c0, None, args.steps, args.warmup, 200, mode=mo.JAC_ANALYTIC)
        ms = el / args.steps * 1e3
        x0 = np.zeros(6, dtype=np_dtype)

        def solve_median(jac):
            for _ in range(5):
                mo.capi.lm_minimize([c0], [jac], x0)
            solves = []
            for _ in range(20):
                t_solve = time.perf_counter()
                xs0, rep0 = mo.capi.lm_minimize([c0], [jac], x0)
                solves.append(time.perf_counter() - t_solve)
            return float(np.median(solves)) * 1e3, xs0, rep0

        solve_ms, xs0, rep0 = solve_median(mo.JAC_NUMERIC)
        choice = c0.lm_choice_stats()
        c0.set_kernel_variant(mo.KERNEL_MOMENTS_ALWAYS)
        one_launch_ms, _, rep1 = solve_median(mo.JAC_NUMERIC)
        c0.set_kernel_variant(variant)
        out["cfg1"] = {"ms_per_step": ms, "value": n0 / (ms * 1e-3), "median_step_us": step_times(st)["median"],
                       "check_sum_sq": ssq, "solve_ms": solve_ms,
                       "solve_iterations": rep0["iterations"], "solve_sweeps": rep0["sweeps"],
                       "solve_status": rep0["status"], "solve_x": [float(v) for v in xs0],
                       "solve_points_chosen": choice[0], "solve_points_literal": choice[1],
                       "solve_ms_moments_always": one_launch_ms, "solve_sweeps_moments_always": rep1["sweeps"],
                       "workload": "point2point, 1k synthetic correspondences (tst/point2point.cpp's size): "
                                   "a blocking analytic linearization per step; solve_ms = the whole "
                                   "forward-difference registration from x = 0 under mopt_lm_minimize "
                                   "(median of 20), the sweep of every point chosen as the blocking call "
                                   "chooses it (solve_points_literal of solve_points_chosen took the literal "
                                   "one, over all 25 solves), in one launch of one workgroup that holds both "
                                   "forms; solve_ms_moments_always = the same with MOPT_KERNEL_MOMENTS_ALWAYS: "
                                   "the moments-only variant of that kernel"}
        c0.close()
        n1 = 1_000_000
        c1, s1, t1 = make_cost(n1)
        del s1, t1
        settle1 = settle_steps(n1, 50.0)

        def p2p(mode, kvariant):
            c1.set_kernel_variant(kvariant)
            el, _, ssq, st = timed_pass(c1, None, args.steps, args.warmup, settle1, mode=mode)
            k_ms, _ = kernel_pass(c1, min(ksteps, 30), mode=mode)
            ms = el / args.steps * 1e3
            return {"ms_per_step": ms, "value": n1 / (ms * 1e-3), "kernel_ms": k_ms,
                    "timed_region": dict(region),
                    "frac": n1 * bpc / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "step_frac": n1 * bpc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "median_step_us": step_times(st)["median"], "check_sum_sq": ssq}

        out["cfg2"] = dict(p2p(mo.JAC_ANALYTIC, mo.KERNEL_AUTO),
                           workload="point2point analytical Jacobian, 1M correspondences")
        out["cfg3"] = dict(p2p(mo.JAC_NUMERIC, mo.KERNEL_AUTO),
                           workload="point2point numerical (finite-difference) Jacobian, 1M "
                                    "correspondences, default evaluation (moments where they meet 1e-6)")
        out["cfg3_literal"] = dict(p2p(mo.JAC_NUMERIC, mo.KERNEL_LITERAL),
                                   workload="the same evaluated as the reference does: 7 residuals, "
                                            "18 quotients per point")
        c1.close()
        n5, costs5, step5, kernel5 = camera_problem(mo, ds)
        mo.capi.link_costs(costs5)
        el, _, y5, st = timed_pass(None, None, args.steps, args.warmup, 500, step_fn=step5, counted=costs5)
        region5 = dict(region)
        mo.capi.link_costs([])
        k_ms = kernel5(min(ksteps, 30))
        ms = el / args.steps * 1e3
        out["cfg5"] = {"ms_per_step": ms, "value": n5 / (ms * 1e-3), "unit": "residual-blocks/s",
                       "kernel_ms": k_ms, "median_step_us": step_times(st)["median"],
                       "check_sum_sq": y5, "timed_region": region5,
                       "workload": "camera-calibration reprojection cost: 100000 elements as two "
                                   "linked costs (40k + 60k), Geman-McClure(100), forward differences, "
                                   "both linearized and summed per step; kernel_ms = both sweeps"}
        for c in costs5:
            c.close()
        out["note"] = ("1 GPU, f64, each timed as the headline: settle | %d warm-up | barrier | %d "
                       "blocking steps | barrier; kernel_ms from a pass of its own" % (args.warmup, args.steps))
        return out

    def cache_proof_configs():
        """Two measurements of the headline's sweep that the 256 MiB Infinity Cache (MALL) cannot have
        served (SURVEY.md §7 "Cache vs HBM"; MI355X_MICROARCH.md: FETCH_SIZE counts MALL hits, so the
        PMC traffic figure cannot tell HBM from the cache).  rotating: R distinct costs of the headline's
        size swept round-robin, so that every line a sweep reads was last touched (R - 1) x its input
        ago; hbm_check: one cost whose input is many times the cache.  Neither involves the other
        ranks (combine off, no barrier inside the kernel passes): at N > 1 every rank runs the rotating
        kernel pass on its own GPU and rank 0's figure is reported."""
        out = {}
        if args.rotating_costs >= 2:
            progress["stage"] = "rotating costs"
            ring = [cost]
            for k in range(1, args.rotating_costs):
                extra, s_k, t_k = make_cost(args.n)
                del s_k, t_k
                ring.append(extra)
            torch.cuda.empty_cache()
            for c in ring:
                c.set_combine(mo.COMBINE_NONE)
            bound = [c.bound_linearize(jac_mode) for c in ring]

            def rot_step(k):
                call, x_in, H_out, b_out, s_out = bound[k % len(ring)]
                x_in[:] = xs[k % 16]
                call()
                return H_out, b_out, s_out[0]

            entry = {"costs": len(ring), "n": args.n, "bytes_between_reuse": (len(ring) - 1) * args.n * bpc,
                     "workload": "the headline's sweep over %d distinct costs of %d correspondences each, "
                                 "round-robin: a sweep's input was last read %d MB of other input ago"
                                 % (len(ring), args.n, (len(ring) - 1) * args.n * bpc // 1_000_000)}
            if world == 1:
                rsteps = max(args.steps, 3 * len(ring))
                el, _, _, st = timed_pass(None, None, rsteps, 2 * len(ring), 4 * len(ring), step_fn=rot_step,
                                          counted=ring)
                ms = el / rsteps * 1e3
                entry.update({"ms_per_step": ms, "value": args.n / (ms * 1e-3),
                              "step_frac": args.n * bpc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "median_step_us": step_times(st)["median"], "timed_region": dict(region)})
            for k in range(2 * len(ring)):
                rot_step(k)
            for c in ring:
                c.set_profiling(1)
            per_cost = args.rotating_launches if args.rotating_launches > 0 else max(min(ksteps, 30), 3)
            for k in range(per_cost * len(ring)):
                rot_step(k)
            prof = [c.profile() for c in ring]
            for c in ring:
                c.set_profiling(False)
            k_ms = sum(p[0] for p in prof) / max(sum(p[1] for p in prof), 1)
            entry.update({"kernel_ms": k_ms, "kernel_launches_timed": sum(p[1] for p in prof),
                          "frac": args.n * bpc / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "per_cost_kernel_ms": [p[0] / max(p[1], 1) for p in prof]})
            out["rotating"] = entry
            for c in ring[1:]:
                c.close()
        if args.hbm_check_n > 0 and world == 1:
            progress["stage"] = "hbm_check"
            nb = args.hbm_check_n
            big, s_b, t_b = make_cost(nb)
            del s_b, t_b
            torch.cuda.empty_cache()
            launches_b = min(ksteps, 30)
            el, _, ssq, st = timed_pass(big, None, launches_b, 3, 3)
            reg = dict(region)
            k_ms, got = kernel_pass(big, launches_b)
            ms = el / launches_b * 1e3
            out["hbm_check"] = {
                "n": nb, "input_bytes": nb * bpc, "kernel_ms": k_ms, "kernel_launches_timed": got,
                "frac": nb * bpc / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ms_per_step": ms, "value": nb / (ms * 1e-3),
                "step_frac": nb * bpc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "check_sum_sq": ssq, "timed_region": reg,
                "workload": "the headline's sweep over one cost of %d correspondences: %.1f GB of input, "
                            "%.0f x the 256 MiB Infinity Cache" % (nb, nb * bpc / 1e9, nb * bpc / 2 ** 28)}
            big.close()
        torch.cuda.empty_cache()
        return out

    progress["stage"] = "kernel-time pass"
    ksteps = args.kernel_steps if args.kernel_steps > 0 else min(args.steps, 100)
    kernel_ms, launches = kernel_pass(cost, ksteps)
    # one untimed sweep of this rank's own shard (combine off) at the x the CPU baseline sweeps at:
    # what check.vs_oracle compares with the oracle's sums
    x_check = ds.X_GENERIC.astype(np_dtype)
    own_sums = cost.linearize(x_check, jac_mode)

    ms_per_step = elapsed / args.steps * 1e3
    achieved = args.n * bpc / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            # keyed by what was launched: "--variant literal" runs another kernel than the default
            key = "%s%s_%s_n%d" % (args.mode, "_literal" if args.variant == "literal" else "",
                                   args.dtype, args.n)
            t = json.load(open(tpath)).get(key) if args.cov == "identity" and args.loss == "none" else None
            if t is not None:
                # PMC counters need rocprofv3 --pmc passes of their own; what is reported here
                # is the committed measurement of this same launch, not a reading of this run
                traffic = {"bytes": t, "source": "profiles/hbm_traffic.json (rocprofv3 --pmc "
                           "FETCH_SIZE x2 + WRITE_SIZE, separate passes, same command)",
                           "measured_this_run": False}
        except Exception:
            traffic = None
    how = {"none": "no collective",
           "host": "finalize kernels publish into one shared pinned host block, host adds 43 fp64 x ranks",
           "peer": "finalize kernels push 43 fp64 into each other's HBM slots (xGMI / IPC) and add on the device",
           "rccl": "RCCL all-reduce of 43 fp64 on the cost's stream",
           "torch": "torch.distributed all_reduce of 43 fp64"}[collective]
    rehearsal = world > 1 and backend != "nccl"
    line.update({
        "metric": "point-correspondences/sec per LM linearization sweep; % HBM peak",
        "value": total * args.steps / elapsed,
        "unit": "correspondences/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if args.total_n else "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": "point2point %s Jacobian, %d correspondences per GPU (%s), "
                        "linearize + %s + result to host each step"
                        % (args.mode, args.n, args.dtype, how),
            "correspondences_per_gpu": args.n,
            "total_correspondences": total,
            "parallelism": "shard%d" % world,
            "collective": collective if not (rehearsal and collective == "torch")
            else "torch/" + backend + " (rehearsal)",
            "collective_is": how,
            "collective_requested": args.collective,
            "collective_fell_back": world > 1 and args.collective not in ("auto", collective),
            "collectives_attached": list(usable),
            "rank_backend": backend if world > 1 else None,
            "rehearsal": ("%d ranks on %d GPU(s), torch.distributed over %s" % (world, ndev, backend))
            if rehearsal else None,
            "kernel_variant": args.variant,
            "covariance": args.cov,
            "loss": args.loss,
            # how the blocking sweeps reached the GPU: AQL packets with agent-scope fences written by the
            # library into an HSA queue of its own (csrc/aql.hpp; MOPT_AQL=0 switches it off), or
            # launches on the cost's HIP stream
            "dispatch": ("direct AQL packets, agent-scope fences (library's own HSA queue)"
                         if cost.direct_dispatches() > 0 else "HIP stream"),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_bytes": traffic["bytes"] if traffic else None,
            "kernel_ms": kernel_ms,
            "kernel_launches_timed": launches,
            "kernel_timing": "separate pass after the timed steps, same dispatch path as the steps; every "
                             "launch carries its own dispatch timestamps (the packet processor's start / "
                             "end of the kernel: the profiling signal of the library's queue, or "
                             "hipExtLaunchKernelGGL on the HIP stream); rank 0's GPU",
            "algorithmic_bytes_per_launch": args.n * bpc,
            # rocprofv3 (ROCP_TOOL_LIBRARIES in the environment) intercepts the queues: its interceptor
            # replaces the completion signal the dispatch's own timestamps are read from and serialises the
            # launches, so under it the in-process kernel times are NOT measurements (measured: 1.3 us through
            # the library's queue, 82 us on the HIP stream, for a 72 us kernel) — the tracer's own stats are
            "under_tracer": bool(os.environ.get("ROCP_TOOL_LIBRARIES")),
        },
        "pct_hbm_peak": 100.0 * achieved / HBM_PEAK_GBS,
        "check": {"sum_sq": float(s), "H00": float(H[0, 0]), "timed_region": headline_region},
        "timing": step_times(stamps, elapsed if world == 1 else None),
    })
    if world > 1:
        # Always present for N > 1, and impossible to miss: the headline above is complete on its own;
        # these two say whether what follows it in the line is.  rccl_incomplete: the RCCL timed pass
        # was started and did not finish (it is the one path that has never run with more than one rank
        # before the driver's own run); extras_incomplete: the comparison passes / the 10 M strong-scaling
        # split did not all finish (set by the watchdog when it has to report the line as it stands).
        line["rccl_incomplete"] = False
        line["extras_incomplete"] = False
        # north_star asks for the scaling THROUGH the RCCL all-reduce: whatever transport the headline
        # `value` used (config.collective), the whole-job rate with the sums added by ncclAllReduce is
        # this key — the same number as rccl.value — and null where RCCL could not be attached (ranks
        # sharing a GPU) or its pass did not finish
        line["value_rccl"] = None
    per = {}
    if world > 1:
        per[collective] = as_step_sees_it(ms_per_step)
        line["by_collective"] = per
        line["roofline"]["step_frac_by_collective"] = {collective: per[collective]["step_frac"]}

    if world == 1 and args.variant == "auto" and args.mode != "analytic_tst":
        # the same sweep evaluated literally (every residual and Jacobian entry per point, as the
        # reference does) — a driver-timed number for that kernel too
        cost.set_kernel_variant(mo.KERNEL_LITERAL)
        literal_ms, _ = kernel_pass(cost, min(ksteps, 30))
        cost.set_kernel_variant(variant)
        line["roofline"]["literal_kernel_ms"] = literal_ms
        line["roofline"]["literal_frac"] = args.n * bpc / (literal_ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    # ---- the HBM figure proper: the same kernel where no cache can hold a sweep's input -----------
    # roofline.frac is a statement about HBM.  The headline's own cost is swept back to back (that is what
    # an LM loop does, and what `value` measures), and a part of its 480 MB survives in the 32 MiB of L2
    # and the 256 MiB Infinity Cache from one sweep to the next: measured, the same kernel on the same
    # cost takes ~70 us back to back and ~74 us with three other costs swept in between.  So the roofline
    # figures are taken over the rotating costs; the back-to-back figure stays beside them.
    same_cost = {"kernel_ms": kernel_ms, "kernel_launches_timed": launches, "achieved": achieved,
                 "frac": achieved / HBM_PEAK_GBS,
                 "what": "the headline's own cost swept back to back, as the timed steps run it: includes "
                         "what L2 and the Infinity Cache keep of its input from one sweep to the next"}
    line["roofline"]["measured_over"] = "one cost, back to back"

    def apply_cache_proof():
        try:
            proof = cache_proof_configs()
        except Exception as e:  # noqa: BLE001 - an extra must not cost the headline its line
            proof = {"cache_proof_error": repr(e)}
            log("cache-proof block failed: %r" % (e,))
        if "rotating" in proof:
            rot = proof["rotating"]
            roof = line["roofline"]
            roof["same_cost"] = same_cost
            roof["kernel_ms"], roof["kernel_launches_timed"] = rot["kernel_ms"], rot["kernel_launches_timed"]
            roof["achieved"] = args.n * bpc / (rot["kernel_ms"] * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            roof["frac_rotating"] = roof["frac"]
            roof["frac_same_cost"] = same_cost["frac"]
            roof["measured_over"] = ("%d distinct costs of %d correspondences swept round-robin (%d MB of "
                                     "other input between two sweeps of the same bytes): HBM only"
                                     % (rot["costs"], args.n, rot["bytes_between_reuse"] // 1_000_000))
            line["pct_hbm_peak"] = 100.0 * roof["frac"]
        if "hbm_check" in proof:
            line["roofline"]["hbm_check_frac"] = proof["hbm_check"]["frac"]
        if proof:
            line.setdefault("configs", {}).update(proof)

    # N = 1: here.  N > 1: after the line can stand and the RCCL pass has run (below) — a first run on more
    # than one GPU must not lose its headline or its RCCL figure to an extra, and the watchdog reports the
    # line as it stands (roofline over one cost, `measured_over` saying so) if this pass should stall.
    if world == 1:
        apply_cache_proof()

    # ---- the other BASELINE configs that fit one GPU, driver-timed in the same line ------------
    if (world == 1 and not args.no_configs and args.dtype == "f64" and args.cov == "identity"
            and args.loss == "none"):
        progress["stage"] = "configs 1, 2, 3, 5"
        try:
            line.setdefault("configs", {}).update(baseline_configs())
        except Exception as e:  # noqa: BLE001 - an extra must not cost the headline its line
            line.setdefault("configs", {})["error"] = repr(e)
            log("configs block failed: %r" % (e,))
    # ---- CPU baseline: rank 0's host cores, every world size -----------------------------------
    progress["stage"] = "cpu baseline"
    if rank == 0:
        line["cpu_baseline"] = None
        if keep_host:
            line["cpu_baseline"], oracle_sums = cpu_baseline(src_host, tgt_host, ds.X_GENERIC, jac_mode,
                                                             args.cpu_seconds)
            # the oracle swept the first `head` correspondences of rank 0's shard at X_GENERIC; the GPU
            # sums to compare are the same shard's, no combine (own_sums) — or, for a shard longer than
            # the CPU sample, a cost over exactly that prefix
            if head == args.n:
                got = own_sums
            else:
                prefix = mo.Point2PointCost(src_host.astype(np_dtype), tgt_host.astype(np_dtype),
                                            device=local_rank, dtype=np_dtype)
                prefix.set_kernel_variant(variant)
                got = prefix.linearize(x_check, jac_mode)
                prefix.close()
            if args.cov == "identity" and args.loss == "none":
                bar = 1e-6 if args.dtype == "f64" else 5e-3
                line["check"]["vs_oracle"] = dict(
                    vs_oracle(got, oracle_sums, bar),
                    what="H, b, sum of squares of one GPU linearize over rank 0's first %d "
                         "correspondences at x = X_GENERIC (kernel variant %r, no combine) against the "
                         "CPU restatement's (oracle/, linearization.h:126-158 / :65-124) on the same "
                         "correspondences; norm-wise max |d| / max |want|" % (head, args.variant))
    if world > 1:
        dist.barrier()  # the others wait here, not inside a combine with a 5 s limit
    # From here on the watchdog reports the line as it stands instead of failing: the headline
    # measurement, its kernel time and the CPU baseline are in it.  The RCCL pass comes next — it is the
    # one path that has never run with more than one rank before this very run, and a collective that
    # never completes must not take the measurement already made down with it.
    progress["complete"] = True

    # ---- RCCL: part of the measurement proper, whatever the headline transport ------------------
    if world > 1:
        progress["stage"] = "RCCL pass"
        info = {"attached": "rccl" in usable}
        if "rccl" in usable:
            ranks_here, user_rank = cost.comm_info()  # ncclCommCount / ncclCommUserRank
            t = torch.tensor([ranks_here, -ranks_here], dtype=torch.int64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            info.update({"ranks": ranks_here, "user_rank": user_rank,
                         "ranks_min_over_ranks": int(t[0].item()),
                         "ranks_max_over_ranks": -int(t[1].item()),
                         "spans_all_ranks": int(t[0].item()) == world == -int(t[1].item())})
            # in the line before the pass starts: should the watchdog have to report from inside it, the
            # line still says that the communicator spanned the ranks and that its pass did not finish
            info["timed_pass"] = "not finished"
            line["rccl"] = info
            line["config"]["rccl_ranks"] = ranks_here
            if collective == "rccl":
                info.update(per["rccl"])
                info["timed_pass"] = "the headline pass"
                line["value_rccl"] = per["rccl"]["value"]
            else:
                line["rccl_incomplete"] = True
                r = guarded_pass(cost, "rccl", args.steps, args.warmup, min(settle, 200))
                line["rccl_incomplete"] = False
                info["timed_pass"] = "done" if r is not None else "failed"
                if r is not None:
                    per["rccl"] = as_step_sees_it(r[0] / args.steps * 1e3)
                    info.update(per["rccl"])
                    line["value_rccl"] = per["rccl"]["value"]
                    info["check"] = {"sum_sq": r[2], "H00": float(r[1][0, 0])}
                else:
                    info["failed_in_timed_pass"] = True
        else:
            info["reason"] = rccl_note
        line["rccl"] = info
        line["config"]["rccl_ranks"] = info.get("ranks")

    if world > 1:
        progress["stage"] = "rotating costs (HBM-only kernel time)"
        apply_cache_proof()
        barrier()  # (the passes below hold collectives again: every rank starts them together)

    if world > 1 and not args.no_compare:
        # every other way of adding the ranks' sums, and no combine at all, K steps each
        for name in ["none"] + [u for u in usable if u not in per] + (["torch"] if "torch" not in per else []):
            if name in per:
                continue
            r = guarded_pass(cost, name, args.steps, min(args.warmup, 10), 20)
            if r is not None:
                per[name] = as_step_sees_it(r[0] / args.steps * 1e3)
        line["ms_per_step_by_collective"] = {k: v["ms_per_step"] for k, v in per.items()}
        line["roofline"]["step_frac_by_collective"] = {k: v["step_frac"] for k, v in per.items()}
        if "none" in per:
            line["ms_per_step_without_collective"] = per["none"]["ms_per_step"]
    if world > 1 and not args.no_compare and not args.total_n:
        # BASELINE config 4: 10 M correspondences IN TOTAL split over the ranks (strong scaling;
        # 60 MB per GPU at 8 ranks, Infinity-Cache resident, latency-bound)
        progress["stage"] = "config 4 (10 M in total)"
        total4 = 10_000_000
        lo, hi = total4 * rank // world, total4 * (rank + 1) // world
        cost4, s4, t4 = make_cost(hi - lo)
        del s4, t4
        attached4 = attach_combines(cost4, rank, world, want=tuple(usable), log=log) if usable else []
        per4 = {}
        for name in ["none"] + attached4:
            r = guarded_pass(cost4, name, args.steps, min(args.warmup, 10),
                             settle_steps(total4 // world, 50.0))
            if r is not None:
                per4[name] = as_step_sees_it(r[0] / args.steps * 1e3, hi - lo, total4)
        combined = {k: v for k, v in per4.items() if k != "none"}
        best = min((v["ms_per_step"], k) for k, v in combined.items()) if combined else (None, None)
        progress["stage"] = "config 4 kernel-time pass"
        k4_ms, _ = kernel_pass(cost4, min(ksteps, 30))
        line["config4_strong"] = {
            "total_correspondences": total4, "correspondences_per_gpu": hi - lo,
            "ms_per_step_by_collective": {k: v["ms_per_step"] for k, v in per4.items()},
            "by_collective": per4, "rccl": per4.get("rccl"),
            "value_rccl": per4["rccl"]["value"] if "rccl" in per4 else None,
            "collective": best[1], "ms_per_step": best[0],
            "value": (total4 / (best[0] * 1e-3)) if best[0] else None,
            "unit": "correspondences/s", "scaling": "strong",
            "kernel_ms": k4_ms,
            "kernel_frac": (hi - lo) * bpc / (k4_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        barrier()
        cost4.close()

    progress["stage"] = "closing"
    if dog is not None:
        dog.cancel()
    emit(True)

    barrier()  # nobody releases its slot blocks while a peer may still push into them
    cost.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
