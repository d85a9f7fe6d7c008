"""Several ranks of a sharded cost inside ONE process, a thread each (tests/test_gpu_multirank.py starts
a few of these processes to reach world size 8 = kMaxPeers, csrc/sweep.hpp — BASELINE config 4's world
size — on a pool that lets at most 6 processes use a GPU at once).  Ranks need not be processes: the
combines of include/moptimizer_hip.h are per cost (mopt_group_* runs a thread per device the same
way), host slots are one POSIX shared-memory object whoever maps it, and a peer slot block exported by
this very process is attached through its pointer (csrc/combine.cpp, ownExport).

    multirank_threads_worker.py <dir> <n_total> <world> <first_rank> <ranks_here>

The side channel (a name, the IPC handles, barriers) is the file system under <dir>; torch.distributed
is per process and not involved.  Every rank writes <dir>/rank<k>.npz; rank 0 also writes what the
sums must be: the per-shard sums of the same shards, one cost each, added in shard order."""
import os
import sys
import threading
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FileChannel:
    """put / gather / barrier between `world` ranks through files of one directory."""

    def __init__(self, root, world, limit_s=180.0):
        self.root, self.world, self.limit_s = root, world, limit_s

    def put(self, key, rank, data):
        tmp = os.path.join(self.root, ".%s.%d.tmp" % (key, rank))
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, os.path.join(self.root, "%s.%d" % (key, rank)))

    def gather(self, key):
        until = time.monotonic() + self.limit_s
        paths = [os.path.join(self.root, "%s.%d" % (key, r)) for r in range(self.world)]
        while not all(os.path.exists(p) for p in paths):
            if time.monotonic() > until:
                missing = [r for r, p in enumerate(paths) if not os.path.exists(p)]
                raise RuntimeError("side channel: ranks %r never wrote %r" % (missing, key))
            time.sleep(0.002)
        return [open(p, "rb").read() for p in paths]

    def barrier(self, tag, rank):
        self.put("barrier-" + tag, rank, b"1")
        self.gather("barrier-" + tag)


def rank_main(rank, world, out, n_total, shm_name, clouds):
    import moptimizer_0_amd as mo
    from moptimizer_0_amd.sharded import shard_range
    from tests import datasets as ds

    src, tgt = clouds
    chan = FileChannel(out, world)
    lo, hi = shard_range(n_total, rank, world)
    cost = mo.Point2PointCost(src[lo:hi], tgt[lo:hi], device=0)
    cost.set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
    cost.set_speculation(False)
    res = {}
    # ---- attach both fused transports at this world size -----------------------------------------
    cost.hostcomm_attach(shm_name, rank, world)
    chan.put("handle", rank, bytes(cost.peer_export(world)))
    cost.peer_attach(chan.gather("handle"), rank, world)
    cost.set_combine(mo.COMBINE_NONE)
    chan.barrier("attached", rank)  # nobody sweeps before everybody has attached
    modes = {"host": mo.COMBINE_HOST, "peer": mo.COMBINE_PEER}
    xs = [ds.X_ZERO, ds.X_GENERIC, ds.X_GENERIC * 0.3]
    for name in ("host", "peer"):
        cost.set_combine(modes[name])
        got = cost.get_combine()
        assert (got[1], got[2]) == (rank, world), got
        for jm in (mo.JAC_ANALYTIC, mo.JAC_NUMERIC):
            for xi, x in enumerate(xs):
                H, b, s = cost.linearize(x, jm)
                c = cost.compute_cost(x)
                res["%s_%d_%d" % (name, jm, xi)] = np.concatenate([H.ravel(order="F"), b, [s, c]])
        # many sweeps back to back: slot parity and sequence numbers with every one of the G slots in use
        chain = np.zeros((300, 43))
        for k in range(300):
            H, b, s = cost.linearize(ds.X_GENERIC * (0.003 * k), mo.JAC_ANALYTIC)
            chain[k] = np.concatenate([H.ravel(order="F"), b, [s]])
        res[name + "_linchain"] = chain
        chan.barrier("done-" + name, rank)
    if os.environ.get("MOPT_TEST_THREADS_LM", "1") != "0":
        # the device-resident loop over the peer slots at this world size: identical iterates on every rank
        cost.set_combine(mo.COMBINE_PEER)
        cost.set_loss(mo.LOSS_NONE)
        x, rep = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], np.zeros(6))
        res["lm_x"] = x
        res["lm_rep"] = np.array([rep["status"], rep["iterations"], rep["sweeps"]])
        H, b, s = cost.linearize(ds.X_GENERIC, mo.JAC_ANALYTIC)  # the sequence numbers stayed in step
        res["after_lm"] = np.concatenate([H.ravel(order="F"), b, [s]])
        # ... and with forward differences, whose sweep the step kernel chooses per evaluated point (the
        # kernels that hold both forms, sums exchanged over the peer slots): every rank sees the same x, so
        # every rank makes the same choice
        x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6))
        res["lm_fd_x"] = x
        res["lm_fd_rep"] = np.array([rep["status"], rep["iterations"], rep["sweeps"]])
        res["lm_fd_choice"] = np.array(cost.lm_choice_stats())
        H, b, s = cost.linearize(ds.X_GENERIC, mo.JAC_NUMERIC)
        res["after_lm_fd"] = np.concatenate([H.ravel(order="F"), b, [s]])
        cost.set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
        chan.barrier("done-lm", rank)
    if rank == 0:
        # what the sums must be: the same shards, one cost each, added in shard order starting from zero
        parts = []
        for k in range(world):
            a, b_ = shard_range(n_total, k, world)
            parts.append(mo.Point2PointCost(src[a:b_], tgt[a:b_], device=0))
            parts[-1].set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
            parts[-1].set_speculation(False)

        def summed(x, jm, with_cost):
            tot = np.zeros(44 if with_cost else 43)
            for c in parts:
                H, b, s = c.linearize(x, jm)
                tot[:36] += H.ravel(order="F")
                tot[36:42] += b
                tot[42] += s
                if with_cost:
                    tot[43] += c.compute_cost(x)
            return tot

        for jm in (mo.JAC_ANALYTIC, mo.JAC_NUMERIC):
            for xi, x in enumerate(xs):
                res["expected_%d_%d" % (jm, xi)] = summed(x, jm, True)
        res["expected_linchain"] = np.array([summed(ds.X_GENERIC * (0.003 * k), mo.JAC_ANALYTIC, False)
                                             for k in range(300)])
        for c in parts:
            c.close()
        whole = mo.Point2PointCost(src, tgt, device=0)
        x, rep = mo.capi.lm_minimize([whole], [mo.JAC_ANALYTIC], np.zeros(6))
        res["lm_whole_x"] = x
        res["lm_whole_rep"] = np.array([rep["status"], rep["iterations"], rep["sweeps"]])
        x, rep = mo.capi.lm_minimize([whole], [mo.JAC_NUMERIC], np.zeros(6))
        res["lm_fd_whole_x"] = x
        whole.close()
    np.savez(os.path.join(out, "rank%d.npz" % rank), **res)
    chan.barrier("closing", rank)  # nobody releases its slot blocks while a peer may still push into them
    cost.close()


def main():
    out, n_total, world, first, here = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    # ranks sharing one GPU: no hardware queues of their own for the sharded costs (csrc/aql.hpp)
    os.environ.setdefault("MOPT_AQL_SHARDED", "0")
    import moptimizer_0_amd  # noqa: F401 - loads the library before the threads race for it
    from tests import datasets as ds
    clouds = ds.synthetic_pair(n_total, seed=11, noise=0.02)
    failures = []

    def guarded(rank):
        try:
            rank_main(rank, world, out, n_total, os.environ["MOPT_TEST_SHM"], clouds)
        except BaseException:  # noqa: BLE001 - reported, and the process exits non-zero
            failures.append((rank, traceback.format_exc()))

    threads = [threading.Thread(target=guarded, args=(first + k,)) for k in range(here)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for rank, text in failures:
        print("rank %d failed:\n%s" % (rank, text), flush=True)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
