"""One rank of the multi-process shard-combine tests (tests/test_gpu_multirank.py starts two or
more of these as fresh processes).  Ranks may share a GPU: the host and peer combines work between
processes on one device (same-device IPC); RCCL is only attempted when every rank has a GPU of its
own.  Writes what it computed to <out>/rank<k>.npz for the parent to compare."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    n_total = int(sys.argv[2])
    import torch
    import torch.distributed as dist

    import moptimizer_0_amd as mo
    from moptimizer_0_amd.sharded import attach_combines, shard_range
    from tests import datasets as ds

    ndev = torch.cuda.device_count()
    own_gpu = ndev >= world
    if not own_gpu and world > 2:
        # more than two ranks on one GPU: no extra hardware queues per rank (the library's direct-dispatch
        # queues, csrc/aql.hpp); two ranks keep them, so that sharded costs are tested on that path too
        os.environ.setdefault("MOPT_AQL_SHARDED", "0")
    device = rank if own_gpu else 0
    torch.cuda.set_device(device)
    dist.init_process_group("gloo")
    src, tgt = ds.synthetic_pair(n_total, seed=11, noise=0.02)
    lo, hi = shard_range(n_total, rank, world)
    if os.environ.get("MOPT_TEST_UNEVEN"):
        # hand-offs have to hold under UNEVEN load: rank 0 sweeps 97 % of the points, the others a
        # sliver each, so the fast ranks' pushes always wait in the slow rank's slots (and the slow
        # rank's arrive last everywhere)
        cut = [0] + [int(n_total * 0.97) + k * ((n_total - int(n_total * 0.97)) // (world - 1))
                     for k in range(world - 1)] + [n_total]
        lo, hi = cut[rank], cut[rank + 1]
    cost = mo.Point2PointCost(src[lo:hi], tgt[lo:hi], device=device)
    cost.set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
    want = ("host", "peer", "rccl") if own_gpu else ("host", "peer")
    notes = []
    usable = attach_combines(cost, rank, world, want=want, log=notes.append)
    res = {"usable": np.array(usable), "notes": np.array(notes)}
    modes = {"host": mo.COMBINE_HOST, "peer": mo.COMBINE_PEER, "rccl": mo.COMBINE_RCCL}
    if os.environ.get("MOPT_TEST_ABSENT"):
        # a rank that never delivers: the others' waits — on the host for the host slots, inside
        # the finalize kernel for the peer slots — must END (MOPT_PEER_TIMEOUT_MS) with an error,
        # not hang the caller or the GPU
        import time
        for name in ("host", "peer"):
            cost.set_combine(modes[name])
            if rank == 0:
                t0 = time.perf_counter()
                try:
                    cost.linearize(ds.X_GENERIC, mo.JAC_ANALYTIC)
                    res[name + "_absent"] = np.array([0.0, 0.0])
                except mo.MoptError as e:
                    res[name + "_absent"] = np.array([1.0 if "error 6" in str(e) else -1.0,
                                                      time.perf_counter() - t0])
            dist.barrier()
            # the sequence numbers of this transport are out of step from here on: not used again
        # the cost itself still works on its own
        cost.set_combine(mo.COMBINE_NONE)
        H, b, s = cost.linearize(ds.X_GENERIC, mo.JAC_ANALYTIC)
        res["alone_after_timeout"] = np.array([s])
        np.savez(os.path.join(out, "rank%d.npz" % rank), **res)
        dist.barrier()
        cost.close()
        dist.destroy_process_group()
        return
    if os.environ.get("MOPT_TEST_FUZZ"):
        # every rank walks the same seeded sequence: blocking calls with and without speculation,
        # the transport switched in between (all ranks at the same point, as the contract asks),
        # loss changes, device-resident solves over the peer slots.  Each call's numbers must be the
        # same words on every rank and equal the unsharded cost's (rank 0 holds one) to rounding.
        rng = np.random.default_rng(int(os.environ["MOPT_TEST_FUZZ"]))
        whole = mo.Point2PointCost(src, tgt, device=device) if rank == 0 else None
        if whole:
            whole.set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
            whole.set_speculation(False)
        got, want, ops = [], [], []
        recent = [rng.uniform(-0.3, 0.3, 6) for _ in range(3)]
        current = usable[0]
        cost.set_combine(modes[current])
        for step in range(int(os.environ.get("MOPT_TEST_FUZZ_STEPS", "1500"))):
            op = rng.random()
            x = recent[rng.integers(3)].copy() if rng.random() < 0.5 else rng.uniform(-0.4, 0.4, 6)
            if rng.random() < 0.3:
                recent[rng.integers(3)] = x.copy()
            if op < 0.40:
                jm = [mo.JAC_ANALYTIC, mo.JAC_NUMERIC, mo.JAC_ANALYTIC_LEFT][rng.integers(3)]
                H, b, s = cost.linearize(x, jm)
                got.append(np.concatenate([H.ravel(order="F"), b, [s]]))
                if whole:
                    Hw, bw, sw = whole.linearize(x, jm)
                    want.append(np.concatenate([Hw.ravel(order="F"), bw, [sw]]))
                ops.append(0)
            elif op < 0.70:
                c = cost.compute_cost(x)
                got.append(np.full(43, c))
                if whole:
                    want.append(np.full(43, whole.compute_cost(x)))
                ops.append(1)
            elif op < 0.80:
                current = usable[rng.integers(len(usable))]
                cost.set_combine(modes[current])
            elif op < 0.86:
                kind = int(rng.integers(2))
                param = float(rng.uniform(1.0, 80.0)) if kind else 0.0
                cost.set_loss(kind, param)
                if whole:
                    whole.set_loss(kind, param)
            elif op < 0.90:
                cost.set_speculation(bool(rng.integers(2)))
            elif "peer" in usable:
                cost.set_combine(mo.COMBINE_PEER)
                k = int(rng.integers(1, 4))
                xs0 = x * 0.2
                xd, rep = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], xs0, max_iterations=k)
                got.append(np.concatenate([xd, np.zeros(37)]))
                if whole:
                    xw, _ = mo.capi.lm_minimize([whole], [mo.JAC_ANALYTIC], xs0, max_iterations=k)
                    want.append(np.concatenate([xw, np.zeros(37)]))
                ops.append(2)
                cost.set_combine(modes[current])
        res["fuzz_got"] = np.array(got)
        res["fuzz_ops"] = np.array(ops)
        if whole:
            res["fuzz_want"] = np.array(want)
            whole.close()
        np.savez(os.path.join(out, "rank%d.npz" % rank), **res)
        dist.barrier()
        cost.close()
        dist.destroy_process_group()
        return
    xs = [ds.X_ZERO, ds.X_GENERIC, ds.X_GENERIC * 0.3]
    for name in usable:
        cost.set_combine(modes[name])
        cost.set_speculation(False)
        for jm in (mo.JAC_ANALYTIC, mo.JAC_NUMERIC):
            for xi, x in enumerate(xs):
                H, b, s = cost.linearize(x, jm)
                c = cost.compute_cost(x)
                res["%s_H_%d_%d" % (name, jm, xi)] = H
                res["%s_b_%d_%d" % (name, jm, xi)] = b
                res["%s_s_%d_%d" % (name, jm, xi)] = np.array([s, c])
        # the LM pattern with speculation: computeCost(x) then linearize(x) answered from the kept
        # result, on every rank alike (the kept result holds the sums of ALL ranks)
        cost.set_speculation(True)
        x = ds.X_GENERIC * 0.7
        cost.linearize(x * 0.5, mo.JAC_NUMERIC)
        sweeps0, hits0 = cost.stats()
        c = cost.compute_cost(x)
        H, b, s = cost.linearize(x, mo.JAC_NUMERIC)
        sweeps1, hits1 = cost.stats()
        res[name + "_spec"] = np.array([sweeps1 - sweeps0, hits1 - hits0, c, s])
        res[name + "_spec_H"] = H
        # many sweeps back to back: slot parity / sequence bookkeeping
        acc = 0.0
        cost.set_speculation(False)
        for k in range(200):
            acc += cost.compute_cost(ds.X_GENERIC * (0.01 * k))
        res[name + "_chain"] = np.array([acc])
        # ... and a long run of full linearizations, every word of every result kept
        chain = np.zeros((600, 43))
        for k in range(600):
            H, b, s = cost.linearize(ds.X_GENERIC * (0.003 * k), mo.JAC_ANALYTIC)
            chain[k, :36] = H.ravel(order="F")
            chain[k, 36:42] = b
            chain[k, 42] = s
        res[name + "_linchain"] = chain
    if "peer" in usable:
        # the whole LM loop resident on every rank's device, each on its shard, the sums of every
        # point added over the ranks inside the finalize kernels: identical iterates everywhere
        cost.set_combine(mo.COMBINE_PEER)
        cost.set_loss(mo.LOSS_NONE)
        x, rep = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], np.zeros(6))
        res["lm_x"] = x
        res["lm_rep"] = np.array([rep["status"], rep["iterations"], rep["sweeps"]])
        # ... and the blocking path keeps working afterwards (sequence numbers stayed in step)
        H, b, s = cost.linearize(ds.X_GENERIC, mo.JAC_ANALYTIC)
        res["after_lm_H"] = H
        cost.set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
    dist.barrier()
    if rank == 0:
        whole = mo.Point2PointCost(src, tgt, device=device)
        x, rep = mo.capi.lm_minimize([whole], [mo.JAC_ANALYTIC], np.zeros(6))
        res["lm_whole_x"] = x
        res["lm_whole_rep"] = np.array([rep["status"], rep["iterations"], rep["sweeps"]])
        whole.close()
        # what the sums must be: the same shards, in one process, added on the host in shard order
        if os.environ.get("MOPT_TEST_UNEVEN"):
            # the group splits evenly; the reference for uneven shards is the sum of per-shard costs
            parts = [mo.Point2PointCost(src[cut[k]:cut[k + 1]], tgt[cut[k]:cut[k + 1]], device=device)
                     for k in range(world)]
            for c in parts:
                c.set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
            chain = np.zeros((600, 43))
            for k in range(600):
                tot = np.zeros(43)
                for c in parts:  # shard order, starting from zero: the order every rank adds in
                    H, b, s = c.linearize(ds.X_GENERIC * (0.003 * k), mo.JAC_ANALYTIC)
                    tot[:36] += H.ravel(order="F")
                    tot[36:42] += b
                    tot[42] += s
                chain[k] = tot
            res["expected_linchain"] = chain
            for c in parts:
                c.close()
        group = mo.Point2PointGroup(src, tgt, [0] * world)
        group.set_loss(mo.LOSS_GEMAN_MCCLURE, 50.0)
        for jm in (mo.JAC_ANALYTIC, mo.JAC_NUMERIC):
            for xi, x in enumerate(xs):
                H, b, s = group.linearize(x, jm)
                c = group.compute_cost(x)
                res["group_H_%d_%d" % (jm, xi)] = H
                res["group_b_%d_%d" % (jm, xi)] = b
                res["group_s_%d_%d" % (jm, xi)] = np.array([s, c])
        acc = 0.0
        for k in range(200):
            acc += group.compute_cost(ds.X_GENERIC * (0.01 * k))
        res["group_chain"] = np.array([acc])
        group.close()
    np.savez(os.path.join(out, "rank%d.npz" % rank), **res)
    dist.barrier()  # nobody releases its slot blocks while a peer may still push into them
    cost.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
