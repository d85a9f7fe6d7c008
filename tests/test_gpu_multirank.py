"""The library's own multi-rank paths with MORE THAN ONE RANK: fresh processes, one shard each,
the 43 sums added over the ranks by the fused combines of include/moptimizer_hip.h
(MOPT_COMBINE_HOST: shared pinned host slots; MOPT_COMBINE_PEER: HBM slots opened over IPC) and,
when every rank has a GPU of its own, by the RCCL all-reduce.  Replaces the host accumulation of
levenberg_marquadt_dyn.cpp:57-59 across shards.  On a one-GPU box the ranks share the device (the
host and peer combines work there; RCCL refuses, and is skipped)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import datasets as ds

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def runner_gives_its_queues_back(hip_lib):
    """The ranks of these tests share this box's GPU with the test runner, and a GPU has a fixed
    number of hardware queues: the runner hands back those its earlier tests' costs left behind
    (mopt_device_trim) before the ranks start — what a parent process that starts workers does."""
    import gc
    gc.collect()
    try:
        hip_lib.capi.device_trim(0)
    except hip_lib.MoptError:
        pass  # a cost of another test module is still alive: its queues stay
    yield


def run_ranks(tmp_path, world, n_total, extra_env=None, timeout=420):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ds.ROOT, "tests", "multirank_worker.py"), str(tmp_path),
             str(n_total)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]


def run_thread_ranks(tmp_path, world, processes, n_total, extra_env=None, timeout=420):
    """`world` ranks as `processes` processes of world / processes rank threads each
    (tests/multirank_threads_worker.py): how this pool — at most 6 processes on a GPU at once, the
    test runner being one — reaches world size 8."""
    assert world % processes == 0
    here = world // processes
    shm = "/mopt-test-%d-%x" % (os.getpid(), int.from_bytes(os.urandom(4), "little"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MOPT_TEST_SHM=shm)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ds.ROOT, "tests", "multirank_threads_worker.py"),
                               str(tmp_path), str(n_total), str(world), str(k * here), str(here)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for k in range(processes)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for k, p in enumerate(procs):
        assert p.returncode == 0, "process %d (ranks %d..%d) failed:\n%s" % (k, k * here, k * here + here - 1,
                                                                             outs[k][-4000:])
    return [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,processes,n_total", [(8, 4, 400_009), (8, 2, 400_009), (4, 4, 400_009),
                                                     (8, 4, 10_000_000)])
def test_eight_ranks_combine_through_the_library(hip_lib, tmp_path, world, processes, n_total):
    """BASELINE config 4's world size, G = kMaxPeers = 8 (csrc/sweep.hpp) — every one of the eight
    slots of both fused transports in use — on the one GPU there is: the 43 sums of every sweep on every
    rank are, word for word, the per-shard sums added in shard order (levenberg_marquadt_dyn.cpp:57-59
    across shards); 300 linearizations back to back per transport; the device-resident loop over the
    peer slots takes identical iterates on all eight ranks and lands on the unsharded solution.
    (8, 4): 4 processes x 2 rank threads - IPC handles between processes and same-process pointers
    mixed; (8, 2): 4 rank threads per process; (4, 4): the threaded worker with one rank per process,
    IPC only; n_total = 10 M: BASELINE config 4's own sizes — 10 M correspondences in eight shards of
    1.25 M — on the one GPU (what it lacks of config 4 is the seven other GPUs and RCCL)."""
    res = run_thread_ranks(tmp_path, world, processes, n_total, timeout=900)
    for name in ("host", "peer"):
        for jm in (0, 2):
            for xi in range(3):
                want = res[0]["expected_%d_%d" % (jm, xi)]
                for r in range(world):
                    got = res[r]["%s_%d_%d" % (name, jm, xi)]
                    # bit-equal: same shards, same kernels, the G additions in shard order
                    assert got[:43].tobytes() == want[:43].tobytes(), (name, jm, xi, r)
                    assert got[43] == want[43], (name, jm, xi, r)
        want = res[0]["expected_linchain"]
        for r in range(world):
            bad = np.argwhere(res[r][name + "_linchain"] != want)
            assert bad.size == 0, (name, r, bad[:5])
    for r in range(1, world):
        assert list(res[r]["lm_rep"]) == list(res[0]["lm_rep"])
        assert res[r]["lm_x"].tobytes() == res[0]["lm_x"].tobytes()
        assert res[r]["after_lm"].tobytes() == res[0]["after_lm"].tobytes()
    assert res[0]["lm_rep"][0] == res[0]["lm_whole_rep"][0]
    assert abs(int(res[0]["lm_rep"][1]) - int(res[0]["lm_whole_rep"][1])) <= 1
    assert np.abs(res[0]["lm_x"] - res[0]["lm_whole_x"]).max() < 1e-9 * 11
    assert np.abs(res[0]["lm_x"] - ds.FIXTURE_X).max() < 1e-3
    # forward differences with the sweep chosen per point, over the peer slots: the same on every rank
    for r in range(1, world):
        assert list(res[r]["lm_fd_rep"]) == list(res[0]["lm_fd_rep"])
        assert res[r]["lm_fd_x"].tobytes() == res[0]["lm_fd_x"].tobytes()
        assert list(res[r]["lm_fd_choice"]) == list(res[0]["lm_fd_choice"])
        assert res[r]["after_lm_fd"].tobytes() == res[0]["after_lm_fd"].tobytes()
    assert res[0]["lm_fd_choice"][0] == res[0]["lm_fd_rep"][2] > 0
    assert np.abs(res[0]["lm_fd_x"] - res[0]["lm_fd_whole_x"]).max() < 1e-7 * 11


def test_a_ninth_rank_is_refused_by_name(hip_lib):
    """One past kMaxPeers: the device-side combine says so when asked (MOPT_ERR_INVALID_ARGUMENT, the
    limit in the message); the host slots, which have no such limit, attach."""
    src, tgt = ds.synthetic_pair(5000, seed=3)
    cost = hip_lib.Point2PointCost(src, tgt)
    with pytest.raises(hip_lib.MoptError, match="at most 8 ranks"):
        cost.peer_export(9)
    handle = cost.peer_export(8)
    with pytest.raises(hip_lib.MoptError, match="bad rank / num_ranks"):
        cost.peer_attach([handle] * 9, 0, 9)
    name = "/mopt-test-nine-%d" % os.getpid()
    cost.hostcomm_attach(name, 8, 9)
    assert cost.get_combine()[1:] == (8, 9)
    cost.close()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_combine_through_the_library(hip_lib, tmp_path, world):
    res = run_ranks(tmp_path, world, 200_003)
    usable = list(res[0]["usable"])
    assert "host" in usable and "peer" in usable, (usable, list(res[0]["notes"]))
    for r in range(1, world):
        assert list(res[r]["usable"]) == usable
    for name in usable:
        for jm in (0, 2):
            for xi in range(3):
                for part in "Hbs":
                    key = "%s_%s_%d_%d" % (name, part, jm, xi)
                    want = res[0]["group_%s_%d_%d" % (part, jm, xi)]
                    for r in range(world):
                        # bit-equal: same shards, same kernels, same order of the G additions
                        assert res[r][key].tobytes() == want.tobytes(), (key, r)
        for r in range(world):
            sweeps, hits, c, s = res[r][name + "_spec"]
            assert (sweeps, hits) == (1, 1), (name, r, sweeps, hits)  # one sweep answered both calls
            assert c == s
            assert res[r][name + "_spec_H"].tobytes() == res[0][name + "_spec_H"].tobytes()
            assert res[r][name + "_chain"].tobytes() == res[0]["group_chain"].tobytes()


def test_handoffs_hold_under_uneven_load(hip_lib, tmp_path):
    """3 ranks, one with 97 % of the points: 600 linearizations per transport, every one of the 43
    words of every result on every rank equal to the sum of the per-shard results formed in shard
    order (a stale or torn slot would show as a wrong word somewhere)."""
    res = run_ranks(tmp_path, 3, 400_001, extra_env={"MOPT_TEST_UNEVEN": "1"})
    want = res[0]["expected_linchain"]
    for name in ("host", "peer"):
        for r in range(3):
            got = res[r][name + "_linchain"]
            assert got.shape == want.shape
            bad = np.argwhere(got != want)
            assert bad.size == 0, (name, r, bad[:5], got[bad[0][0]][:3], want[bad[0][0]][:3])


def test_a_rank_that_never_delivers_is_an_error_not_a_hang(hip_lib, tmp_path):
    """Rank 1 attaches and then stays away: rank 0's blocking linearize returns
    MOPT_ERR_PEER_TIMEOUT after MOPT_PEER_TIMEOUT_MS — from the host-side wait (host slots) and
    from the bounded wait inside the finalize kernel (peer slots) — and the cost keeps working."""
    res = run_ranks(tmp_path, 2, 50_001, extra_env={"MOPT_TEST_ABSENT": "1", "MOPT_PEER_TIMEOUT_MS": "400"})
    for name in ("host", "peer"):
        flag, seconds = res[0][name + "_absent"]
        assert flag == 1.0, (name, flag)            # MoptError with code 6 (MOPT_ERR_PEER_TIMEOUT)
        assert 0.3 < seconds < 5.0, (name, seconds)  # it waited for the limit, and not much longer
    assert res[0]["alone_after_timeout"][0] > 0


def test_device_resident_lm_over_sharded_cost(hip_lib, tmp_path):
    """mopt_lm_minimize on every rank with MOPT_COMBINE_PEER: all ranks return the same x, bit for
    bit, and it is the solution of the unsharded problem."""
    res = run_ranks(tmp_path, 2, 200_003)
    assert list(res[0]["lm_rep"]) == list(res[1]["lm_rep"])  # status, iterations, sweeps
    assert res[0]["lm_x"].tobytes() == res[1]["lm_x"].tobytes()
    # against the unsharded cost the sums differ by fp64 reassociation only; the last iteration,
    # taken at the noise floor, may differ
    assert res[0]["lm_rep"][0] == res[0]["lm_whole_rep"][0]
    assert abs(int(res[0]["lm_rep"][1]) - int(res[0]["lm_whole_rep"][1])) <= 1
    assert np.abs(res[0]["lm_x"] - res[0]["lm_whole_x"]).max() < 1e-9 * 11
    assert np.abs(res[0]["lm_x"] - ds.FIXTURE_X).max() < 1e-3  # noisy data: near the fixture pose
    assert res[0]["after_lm_H"].tobytes() == res[1]["after_lm_H"].tobytes()


def test_sharded_sums_match_the_unsharded_cost(hip_lib, tmp_path, oracle):
    """Shard-invariance against the oracle on the whole data set (1e-6 bar; the sums differ from a
    single-cost sweep only by fp64 reassociation)."""
    from tests import oracle_binding as ob
    n = 200_003
    res = run_ranks(tmp_path, 2, n)
    src, tgt = ds.synthetic_pair(n, seed=11, noise=0.02)
    xs = [ds.X_ZERO, ds.X_GENERIC, ds.X_GENERIC * 0.3]
    for xi, x in enumerate(xs):
        Hr, br, sr = oracle.p2p_linearize(src, tgt, x, cost_class=ob.ANALYTIC_DYN,
                                          layout=ob.LAYOUT_ROW_MAJOR, loss_kind=1, loss_param=50.0)
        for name in ("host", "peer"):
            H = res[1]["%s_H_0_%d" % (name, xi)]
            b = res[1]["%s_b_0_%d" % (name, xi)]
            s = res[1]["%s_s_0_%d" % (name, xi)]
            assert np.abs(H - Hr).max() <= 1e-6 * np.abs(Hr).max()
            assert np.abs(b - br).max() <= 1e-6 * np.abs(br).max()
            assert abs(s[0] - sr) <= 1e-6 * sr and abs(s[1] - sr) <= 1e-6 * sr


@pytest.mark.parametrize("world,seed", [(2, 7), (3, 8)])
def test_random_call_sequences_over_sharded_costs(hip_lib, tmp_path, world, seed):
    """Every rank walks the same seeded sequence of calls — blocking sweeps with and without
    speculation, the transport switched between the host and the peer slots along the way, loss
    changes, device-resident solves over the peer slots: the numbers are the same words on every rank
    and equal the unsharded cost's to rounding (the shards' sums are added in another order)."""
    res = run_ranks(tmp_path, world, 150_001, {"MOPT_TEST_FUZZ": str(seed)})
    got0, want, ops = res[0]["fuzz_got"], res[0]["fuzz_want"], res[0]["fuzz_ops"]
    assert len(got0) == len(want) == len(ops) and len(got0) > 900
    for r in range(1, world):
        assert res[r]["fuzz_got"].tobytes() == got0.tobytes(), r
    scale = np.abs(want).max(axis=1, keepdims=True)
    err = np.abs(got0 - want) / scale
    assert err[ops != 2].max() < 1e-11, err[ops != 2].max()   # sums: rounding only
    assert err[ops == 2].max() < 1e-9, err[ops == 2].max()    # iterates of up to three LM iterations
    assert (ops == 2).sum() > 30


def _check_multirank_line(line, world, rehearsal):
    """What a > 1-rank bench line must carry: the headline transport named, the
    RCCL figure with the rank count the communicator itself reports (or, when ranks share a GPU,
    why there is none), every transport's step time / rate / step-level roofline fraction, and the
    CPU baseline on rank 0."""
    assert line["n_gpus"] == world
    cfg = line["config"]
    assert cfg["collective"] in ("host", "peer", "rccl"), cfg["collective"]
    assert cfg["collective"] in line["by_collective"]
    for name, entry in line["by_collective"].items():
        assert entry["ms_per_step"] > 0 and entry["value"] > 0 and 0 < entry["step_frac"] < 1.2, (name, entry)
    assert set(line["roofline"]["step_frac_by_collective"]) == set(line["by_collective"])
    assert {"none", "host", "peer"} <= set(line["ms_per_step_by_collective"])
    rccl = line["rccl"]
    c4 = line["config4_strong"]
    # the rate THROUGH the RCCL all-reduce (what north_star's scaling is judged on) is a top-level key
    # whatever the headline transport was: a number wherever RCCL is attached, null on a shared GPU
    assert "value_rccl" in line and "value_rccl" in c4
    if rehearsal:
        assert rccl["attached"] is False and "share" in rccl["reason"]
        assert cfg["rehearsal"] and cfg["rccl_ranks"] is None
        assert line["value_rccl"] is None and c4["value_rccl"] is None and c4["rccl"] is None
    else:
        assert rccl["attached"] and rccl["ranks"] == world and rccl["spans_all_ranks"]
        assert rccl["ms_per_step"] > 0 and cfg["rccl_ranks"] == world
        assert "rccl" in line["by_collective"]
        assert line["value_rccl"] == rccl["value"] > 0
        assert c4["value_rccl"] == c4["rccl"]["value"] > 0
    vs = line["check"]["vs_oracle"]   # rank 0's shard, combine off, against the CPU restatement
    assert vs["ok"] and max(vs["H_rel"], vs["b_rel"], vs["cost_rel"]) <= vs["bar"] == 1e-6, vs
    assert len(line["timing"]["per_step_us"]) == line["steps"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 1e6
    # the two flags a reader of a one-shot multi-GPU run looks at first: always there, both clear
    assert line["extras_incomplete"] is False and line["rccl_incomplete"] is False, line.get("note")
    assert cfg["collective_requested"] in ("auto", "host", "peer", "rccl", "torch")
    if rccl.get("timed_pass") in ("done", "the headline pass"):
        assert 0 < rccl["step_frac"] < 1.2 and rccl["ms_per_step"] > 0
    assert c4["total_correspondences"] == 10_000_000 and c4["ms_per_step"] > 0
    assert c4["kernel_ms"] > 0 and set(c4["by_collective"]) >= {"none", "host", "peer"}


def test_bench_launches_its_own_ranks(hip_lib):
    """`python bench.py --gpus 2` without a launcher (as the driver calls it), rehearsed on however
    many GPUs this box has: one JSON line, n_gpus = 2, a fused combine selected, and every field a
    > 1-rank line has to carry."""
    import json
    env = dict(os.environ)
    import torch
    rehearsal = torch.cuda.device_count() < 2
    if rehearsal:
        env["MOPT_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ds.ROOT, "bench.py"), "--gpus", "2",
                          "--steps", "20", "--warmup", "3", "--n", "300000", "--settle-ms", "5",
                          "--cpu-seconds", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["check"]["H00"] == 600000.0  # both shards were summed
    _check_multirank_line(line, 2, rehearsal)


def test_eight_rank_processes_on_one_gpu_are_refused_by_name(hip_lib):
    """BASELINE config 4's world size as rank PROCESSES cannot be rehearsed on this pool: it ends a run that
    puts more than 6 processes on a GPU (its process guard), the test runner being one — so `bench.py
    --gpus 8` on one GPU would be killed, not measured.  Told the limit (MOPT_MAX_PROCESSES_PER_GPU), the
    bench refuses before it starts a single rank: exit code 4 and the resource named on stderr, within a
    second or two — not a time-out.  (World size 8 itself runs above, as rank threads.)"""
    import json
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("several GPUs: eight ranks need not share one")
    env = dict(os.environ, MOPT_BENCH_BACKEND="gloo", MOPT_MAX_PROCESSES_PER_GPU="6")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ds.ROOT, "bench.py"), "--gpus", "8", "--steps", "20",
                          "--warmup", "5", "--collective", "rccl"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 4, (out.returncode, out.stderr.decode()[-2000:])
    assert out.stdout.decode().strip() == ""
    said = json.loads([ln for ln in out.stderr.decode().splitlines() if ln.startswith("{")][-1])
    assert said["resource"] == "processes per GPU" and said["ranks"] == 8 and said["limit"] == 6
    assert "not started" in said["error"]
    assert time.time() - t0 < 60
    # ... and a world size the limit admits is not refused by that check (5 ranks + the caller = 6)
    env["MOPT_MAX_PROCESSES_PER_GPU"] = "6"
    import bench
    os.environ.update({"MOPT_BENCH_BACKEND": "gloo", "MOPT_MAX_PROCESSES_PER_GPU": "6"})
    try:
        bench.refuse_more_processes_than_the_gpu_admits(5)
        with pytest.raises(SystemExit):
            bench.refuse_more_processes_than_the_gpu_admits(6)
    finally:
        os.environ.pop("MOPT_BENCH_BACKEND", None)
        os.environ.pop("MOPT_MAX_PROCESSES_PER_GPU", None)


def test_driver_command_rehearsed_with_four_ranks(hip_lib):
    """The driver's multi-GPU command — default workload (10 M correspondences per rank), default
    settling, every extra pass, the 10 M strong-scaling split, the CPU baseline; plus `--collective
    rccl`, so that the transport north_star names is the one asked for — with 4 ranks (this pool lets 6 processes share one GPU, and the test runner is one of them;
    scripts/rehearse_driver_command.sh runs the same with 6 outside pytest), started by
    torch.distributed.run like the driver does.  It must report every field and finish well inside
    the driver's 600 s."""
    import json
    import time
    import torch
    ndev = torch.cuda.device_count()
    world = 4
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    rehearsal = ndev < world
    if rehearsal:
        env["MOPT_BENCH_BACKEND"] = "gloo"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    t0 = time.time()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ds.ROOT, "bench.py"),
                          "--gpus", str(world), "--steps", "20", "--warmup", "5",
                          "--collective", "rccl"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=500)
    elapsed = time.time() - t0
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["check"]["H00"] == 10_000_000.0 * world
    assert line["config"]["correspondences_per_gpu"] == 10_000_000
    try:
        _check_multirank_line(line, world, rehearsal)
    except (AssertionError, TypeError, KeyError):
        # what the ranks said about a transport that dropped out (bench.py logs it and carries on)
        said = [ln for ln in out.stderr.decode().splitlines() if "rank" in ln and ("failed" in ln or "unavailable" in ln)]
        print("\n".join(said[-12:]))
        raise
    # RCCL asked for by name: the headline where it can run; where the ranks share a GPU it refuses, the
    # bench falls back to the fused combines and says so (the branch a driver run would take if
    # ncclCommInitRank failed on its node)
    assert line["config"]["collective_requested"] == "rccl"
    if rehearsal:
        assert line["config"]["collective_fell_back"] is True and line["config"]["collective"] in ("host", "peer")
        assert line["rccl"]["attached"] is False
    else:
        assert line["config"]["collective"] == "rccl" and line["config"]["collective_fell_back"] is False
        assert line["rccl"]["timed_pass"] == "the headline pass"
    assert elapsed < 300, elapsed
    keep = os.path.join(ds.ROOT, "gpurun_out")
    if os.path.isdir(keep):
        line["_rehearsal_wall_s"] = elapsed
        with open(os.path.join(keep, "r6_4rank_driver_command_rehearsal.json"), "w") as f:
            json.dump(line, f, indent=1)
