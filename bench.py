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

Prints ONE JSON line on rank 0 (contract in the task statement) including
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    single-threaded: linearization.h:97,142) on a bounded prefix of the same workload."""
    from tests import oracle_binding as ob
    oracle = ob.load()
    cost_class = ob.NUMERIC_DYN if jac_mode == 2 else ob.ANALYTIC_DYN
    layout = ob.LAYOUT_TST if jac_mode == 1 else ob.LAYOUT_ROW_MAJOR
    sample = src_host.shape[0]
    sweeps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < target_seconds and sweeps < 1000:
        oracle.p2p_linearize(src_host, tgt_host, x, cost_class=cost_class, layout=layout)
        sweeps += 1
        dt = time.perf_counter() - t0
    out = {
        "value": sample * sweeps / dt,
        "unit": "correspondences/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d sweeps over the first %d correspondences of rank 0's shard, %.1f s of CPU "
                  "work (single thread, as the reference's linearize loop)" % (sweeps, sample, dt),
    }
    # for orientation only: the same sweep split over every host core (the reference parallelises
    # only its cost-only sweep, linearization.h:52)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    oracle.p2p_linearize(src_host, tgt_host, x, cost_class=cost_class, layout=layout, threads=cores)
    dt = time.perf_counter() - t0
    out["all_cores"] = {"value": sample / dt, "cores": cores}
    return out


def camera_main(args):
    """BASELINE config 5 (latency-dominated: 4 MB of input): one step = the multi-objective
    linearization of levenberg_marquadt_dyn.cpp:48-60 — linearize both costs, add H, b, cost on the
    host."""
    import moptimizer_0_amd as mo
    from tests import datasets as ds
    quiesce_python_gc()
    n, split = 100_000, 40_000
    pts, pix = ds.synthetic_camera(n, seed=17)
    costs = [mo.ReprojectionCost(pts[:split], pix[:split]), mo.ReprojectionCost(pts[split:], pix[split:])]
    for c in costs:
        c.set_loss(mo.LOSS_GEMAN_MCCLURE, 100.0)
    x = np.zeros(6)

    # the optimizer's loop over its costs (levenberg_marquadt_dyn.cpp:48-60) through pre-bound calls and
    # reused buffers, as the point2point workload does: what is timed is the library, not allocations
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
    for c in costs:  # kernel times from a separate pass: a profiled launch carries two events
        c.set_profiling(True)
    for k in range(args.steps):
        step(k)
    prof = [c.profile() for c in costs]
    kernel_ms = sum(p[0] for p in prof) / max(prof[0][1], 1)  # both costs' sweeps per step
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


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
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

    def timed_pass(the_cost, combine, steps, warmup, settle, via_torch=None):
        """barrier | K blocking steps | barrier, wall clock, max over ranks.  Profiling is off.
        A step that fails (a peer that never delivers ends in MOPT_ERR_PEER_TIMEOUT, not in a
        hang) does not take this rank out of the sequence of collectives: it stops stepping, goes
        through both barriers and the max-reduce like everybody else, and raises afterwards."""
        err = None
        call = x_in = H_out = b_out = s_out = None
        try:
            if combine is not None:
                the_cost.set_combine(modes[combine])
            call, x_in, H_out, b_out, s_out = the_cost.bound_linearize(jac_mode)
        except Exception as e:  # noqa: BLE001
            err = e

        def step(k):
            if via_torch is not None:
                return via_torch.linearize(xs[k % 16], jac_mode)
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
        barrier()
        t0 = time.perf_counter()
        if err is None:
            try:
                for k in range(steps):
                    H, b, s = step(k)
            except Exception as e:  # noqa: BLE001
                err = e
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        if err is not None:
            raise err
        return elapsed, np.array(H, dtype=np.float64), float(s)

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
        elapsed, H, s = timed_pass(cost, None, args.steps, args.warmup, settle)
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
        elapsed, H, s = res

    # ---- kernel time: a pass of its own, every launch carrying its dispatch timestamps ---------
    def kernel_pass(the_cost, steps):
        the_cost.set_combine(mo.COMBINE_NONE)  # the kernel's duration does not involve the ranks
        call, x_in, _, _, _ = the_cost.bound_linearize(jac_mode)
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

    progress["stage"] = "kernel-time pass"
    ksteps = args.kernel_steps if args.kernel_steps > 0 else min(args.steps, 100)
    kernel_ms, launches = kernel_pass(cost, ksteps)

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
            "kernel_timing": "separate pass after the timed steps; every launch carries its "
                             "dispatch timestamps (hipExtLaunchKernelGGL); rank 0's GPU",
            "algorithmic_bytes_per_launch": args.n * bpc,
        },
        "pct_hbm_peak": 100.0 * achieved / HBM_PEAK_GBS,
        "check": {"sum_sq": float(s), "H00": float(H[0, 0])},
    })
    if world > 1:
        # Always present for N > 1, and impossible to miss: the headline above is complete on its own;
        # these two say whether what follows it in the line is.  rccl_incomplete: the RCCL timed pass
        # was started and did not finish (it is the one path that has never run with more than one rank
        # before the driver's own run); extras_incomplete: the comparison passes / the 10 M strong-scaling
        # split did not all finish (set by the watchdog when it has to report the line as it stands).
        line["rccl_incomplete"] = False
        line["extras_incomplete"] = False
    per = {}
    if world > 1:
        per[collective] = as_step_sees_it(ms_per_step)
        line["by_collective"] = per
        line["roofline"]["step_frac_by_collective"] = {collective: per[collective]["step_frac"]}

    # ---- CPU baseline: rank 0's host cores, every world size -----------------------------------
    progress["stage"] = "cpu baseline"
    if rank == 0:
        line["cpu_baseline"] = (cpu_baseline(src_host, tgt_host, ds.X_GENERIC, jac_mode, args.cpu_seconds)
                                if keep_host else None)
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
            else:
                line["rccl_incomplete"] = True
                r = guarded_pass(cost, "rccl", args.steps, args.warmup, min(settle, 200))
                line["rccl_incomplete"] = False
                info["timed_pass"] = "done" if r is not None else "failed"
                if r is not None:
                    per["rccl"] = as_step_sees_it(r[0] / args.steps * 1e3)
                    info.update(per["rccl"])
                    info["check"] = {"sum_sq": r[2], "H00": float(r[1][0, 0])}
                else:
                    info["failed_in_timed_pass"] = True
        else:
            info["reason"] = rccl_note
        line["rccl"] = info
        line["config"]["rccl_ranks"] = info.get("ranks")

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
    if world == 1 and args.variant == "auto" and args.mode != "analytic_tst":
        # the same sweep evaluated literally (every residual and Jacobian entry per point, as the
        # reference does) — a driver-timed number for that kernel too
        cost.set_kernel_variant(mo.KERNEL_LITERAL)
        literal_ms, _ = kernel_pass(cost, min(ksteps, 30))
        cost.set_kernel_variant(variant)
        line["roofline"]["literal_kernel_ms"] = literal_ms
        line["roofline"]["literal_frac"] = args.n * bpc / (literal_ms * 1e-3) / 1e9 / HBM_PEAK_GBS

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
