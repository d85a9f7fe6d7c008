"""Sharding of one cost over ranks (one process per GPU).

Every correspondence adds an independent term to H, b and the cost
(/root/reference/include/moptimizer/linearization.h:113-115,150-152), so a rank sweeps its own
contiguous index range and the n*n + n + 1 = 43 partial sums are combined with ONE all-reduce
per sweep — RCCL over xGMI when the tensor lives on a GPU (torch backend "nccl"), gloo in the
CPU tests.  This is the only collective on the path; the per-cost accumulation of the LM loop
(/root/reference/src/levenberg_marquadt_dyn.cpp:57-59) stays on the host as in the reference.
"""
import torch
import torch.distributed as dist

RESULT_DOUBLES = 43


def shard_range(count, rank, world_size):
    """Contiguous index range [lo, hi) of `rank`: [rank*N/G, (rank+1)*N/G)."""
    return (count * rank) // world_size, (count * (rank + 1)) // world_size


class ShardedSweep:
    """Combines per-rank sweep partials.

    local_linearize(x, jac_mode, out): enqueue this rank's linearize into `out`
        (float64[43] tensor = H column-major | b | sum_sq) on the current stream.
    local_cost(x, out): same for the cost-only sweep (out[42] receives the sum).
    """

    def __init__(self, local_linearize, local_cost=None, device="cpu", group=None):
        self.local_linearize = local_linearize
        self.local_cost = local_cost
        self.group = group
        self.device = torch.device(device)
        self.result = torch.zeros(RESULT_DOUBLES, dtype=torch.float64, device=self.device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def _all_reduce(self, t):
        # gloo cannot reduce device tensors: rehearsal runs (several ranks sharing one GPU, gloo
        # backend) take the sums through the host; with RCCL ("nccl") the tensor stays in HBM.
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def linearize_device(self, x, jac_mode):
        """Asynchronous: returns the device tensor holding the all-reduced sums."""
        self.local_linearize(x, jac_mode, self.result)
        if self.world > 1:
            self._all_reduce(self.result)
        return self.result

    def linearize(self, x, jac_mode):
        r = self.linearize_device(x, jac_mode).cpu().numpy()
        return r[:36].reshape(6, 6, order="F").copy(), r[36:42].copy(), float(r[42])

    def compute_cost(self, x):
        self.local_cost(x, self.result)
        if self.world > 1:
            self._all_reduce(self.result[42:43])
        return float(self.result[42].item())


def gpu_point2point_sweep(cost, jac_mode_default=None):
    """ShardedSweep over a Point2PointCost (this rank's shard) using torch's current stream.

    The all-reduce is this wrapper's: the cost itself must not be combining (with the peer
    transport selected the asynchronous calls already return the sums over all ranks, and the
    all-reduce would multiply them by the world size)."""

    def _local_only():
        mode = cost.get_combine()[0]
        if mode != 0:  # MOPT_COMBINE_NONE
            raise RuntimeError("ShardedSweep all-reduces the ranks' sums itself: select "
                               "set_combine(COMBINE_NONE) on the cost first (combine mode %d is "
                               "selected)" % mode)

    def lin(x, jac_mode, out):
        _local_only()
        cost.linearize_async(x, jac_mode, out.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)

    def cst(x, out):
        _local_only()
        cost.compute_cost_async(x, out.data_ptr() + 42 * 8,
                                torch.cuda.current_stream().cuda_stream)

    dev = torch.device("cuda", torch.cuda.current_device())
    return ShardedSweep(lin, cst, device=dev)


def _all_agree(ok, group=None):
    """True only if every rank says ok (MIN all-reduce; on the device with RCCL, on the host with
    gloo)."""
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item()) == 1


def attach_combines(cost, rank, world, want=("host", "peer", "rccl"), group=None, log=None):
    """Attach the shard-combine transports of include/moptimizer_hip.h to this rank's cost, using
    torch.distributed only as the side channel (a name, IPC handles, an RCCL id).  A transport
    counts only if EVERY rank attached it.  Returns the list of usable transports, in `want` order;
    collective: every rank calls it with the same arguments.  RCCL cannot span ranks that share a
    GPU, so it is only tried with the "nccl" backend (one rank per GPU).

    The C attach calls select what they attach; a transport that attached here but not on every
    rank must not stay selected (this rank's next blocking sweep would wait for peers that never
    push), and neither should whichever happened to be attached last.  On return the cost is
    back on the transport it had selected on entry (MOPT_COMBINE_NONE for a fresh cost): the
    caller picks one of the returned names with `cost.set_combine(...)`, on every rank alike."""
    from . import _capi as capi
    import os
    selected_on_entry = cost.get_combine()[0]
    usable = []
    for name in want:
        ok, err = True, None
        try:
            if name == "host":
                tag = [("/mopt-%d-%x" % (os.getpid(), int.from_bytes(os.urandom(4), "little")))
                       if rank == 0 else None]
                dist.broadcast_object_list(tag, src=0, group=group)
                cost.hostcomm_attach(tag[0], rank, world)
            elif name == "peer":
                mine, err = None, None
                try:
                    mine = cost.peer_export(world)
                except Exception as e:  # still take part in the gather below
                    err = e
                handles = [None] * world
                dist.all_gather_object(handles, mine, group=group)
                if err is not None or any(h is None for h in handles):
                    raise err or RuntimeError("a rank could not export its slot block")
                cost.peer_attach(handles, rank, world)
            elif name == "rccl":
                if dist.get_backend(group) != "nccl":
                    raise RuntimeError("ranks share GPUs (rehearsal backend): RCCL not attempted")
                ids = [capi.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0, group=group)
                cost.comm_init_rank(ids[0], rank, world)
            else:
                raise ValueError(name)
        except Exception as e:
            ok, err = False, e
        if _all_agree(ok, group):
            usable.append(name)
        elif log is not None:
            log("rank %d: combine transport %r unavailable (%s)" % (rank, name, err or "on a peer"))
    cost.set_combine(selected_on_entry)
    dist.barrier(group=group)  # nobody sweeps before everybody has attached
    return usable
