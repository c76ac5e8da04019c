"""
Multi-GPU execution of the pair grid: one process per GPU under torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Pairs are independent; the grid is
cut into B x B track tiles that libacx deals to the ranks by cost (acx_grid_plan), every rank
writes the scores of its tiles into ONE dense device buffer (acx_grid_run), and the only exchange
on the path is ONE gather of those buffers to rank 0 -- device to device under
RCCL, no host staging.  Rank 0 then scatters the tiles into the N x N matrices.
"""
import numpy as np

__all__ = ["world", "bind_device", "barrier", "broadcast_object", "on_root", "any_rank", "shard_bounds", "gather_scores",
           "gather_tiles", "gather_tiles_device", "exchange_kind", "exchange_in_use"]

# The GPU this process works on (set by bind_device).  Under "nccl" EVERY collective needs a device:
# torch picks torch.cuda.current_device() for dist.barrier() and for the tensors behind
# broadcast_object_list(), which is cuda:0 on every rank unless somebody called set_device -- RCCL then
# reports "Duplicate GPU detected" or hangs.  libacx picks its GPU from LOCAL_RANK on its own and never
# touches torch's current device, so the package binds it here, once, when a context joins a process group.
_BOUND = None


def _forced():
    """ACX_GRID_VIA_COLLECTIVE=1: take the multi-rank code path -- tile buffers, the all-gather, the broadcasts --
    even when the process group has ONE rank.  A 1-GPU box cannot hold two RCCL ranks (one communicator rank per
    device), so this is how the nccl branch of this module gets executed there at all
    (tests/test_gpu_grid.py::test_nccl_world_of_one)."""
    import os
    return os.environ.get("ACX_GRID_VIA_COLLECTIVE", "") not in ("", "0")


def single():
    """True when no collective is needed: no process group, or one rank and not forced through the collectives.
    (Imports torch.distributed when torch is installed -- importing it creates no process group and touches no GPU.)"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size() == 1 and not _forced()
    except ImportError:
        pass
    return True


def world():
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def _is_nccl():
    import torch.distributed as dist
    return dist.get_backend() == "nccl"


def bind_device(index):
    """Make GPU `index` this process's device for torch's collectives (torch.cuda.set_device) -- called by
    the device-backed classes with the device of their libacx context before the first collective.  One
    process, one GPU: binding a second, different device is an error."""
    global _BOUND
    index = int(index)
    if _BOUND is not None and _BOUND != index:
        raise RuntimeError("acoss_amd.dist: this process is bound to GPU %d, cannot also use GPU %d "
                           "(one process per GPU)" % (_BOUND, index))
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(index)
    _BOUND = index


def _collective_device():
    """torch.device the nccl collectives of this rank run on: the bound GPU, else LOCAL_RANK's."""
    import os
    import torch
    if _BOUND is None:
        bind_device(int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cuda", _BOUND)


def barrier():
    if not single():
        import torch.distributed as dist
        if _is_nccl():
            dist.barrier(device_ids=[_collective_device().index])
        else:
            dist.barrier()


def broadcast_object(obj, src=0):
    """Small picklable object from rank `src` to every rank (a collective: every rank must call it)."""
    rank, ws = world()
    if single():
        return obj
    import torch.distributed as dist
    box = [obj if rank == src else None]
    if _is_nccl():
        dist.broadcast_object_list(box, src=src, device=_collective_device())
    else:
        dist.broadcast_object_list(box, src=src)
    return box[0]


def on_root(fn, src=0):
    """Run `fn()` on rank `src` only and hand its result to every rank (a collective: every rank must call it).
    An exception on the root is broadcast too and re-raised on EVERY rank -- a rank-0-only section that fails (a
    missing feature file while the clique table is built, a bad matrix in the statistics) must not leave the other
    ranks waiting in the broadcast for ever."""
    rank, ws = world()
    if single():
        return fn()
    box, err = None, None
    if rank == src:
        try:
            box = ("ok", fn())
        except Exception as e:                        # noqa: BLE001 -- whatever it is, the other ranks must hear of it
            err = e
            box = ("err", "%s: %s" % (type(e).__name__, e))
    kind, payload = broadcast_object(box, src=src)
    if kind == "err":
        if err is not None:
            raise err
        raise RuntimeError("rank %d failed in a rank-%d-only section: %s" % (src, src, payload))
    return payload


def any_rank(flag):
    """True on every rank iff `flag` is true on at least one (a collective: every rank must call it).
    Decisions that lead into another collective are taken with this, never from rank-local state."""
    rank, ws = world()
    if single():
        return bool(flag)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_collective_device() if _is_nccl() else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def exchange_kind():
    """"gather" (default: ONE gather of the per-rank buffers to rank 0, the owner of the result -- only rank 0 holds
    world x stride floats) or "allgather" (ACX_GRID_EXCHANGE=allgather: every rank receives every buffer, rounds 1-4)."""
    import os
    return "allgather" if os.environ.get("ACX_GRID_EXCHANGE", "") == "allgather" else "gather"


# The default exchange -- torch.distributed.gather into unbind() views of ONE tensor on rank 0 -- is grouped send / recv inside
# ProcessGroupNCCL; whether a given torch / RCCL pair accepts it is found out ONCE per process group, on a few floats, before the
# score buffers go through it: every rank sends (rank + 1) x 4, rank 0 checks what arrived, the verdict is agreed through
# any_rank() (an all-reduce), and a gather that raised or returned the wrong values turns the exchange into the all-gather of
# rounds 1-4 on EVERY rank.  (An exception out of argument checking is raised on all ranks alike: nobody is left inside it.)
_GATHER_OK = {}


def _gather_probe():
    import torch
    import torch.distributed as dist
    rank, ws = world()
    key = (dist.get_backend(), ws)
    if key in _GATHER_OK:
        return _GATHER_OK[key]
    dev = _collective_device() if _is_nccl() else torch.device("cpu")
    bad = False
    try:
        mine = torch.full((4,), float(rank + 1), dtype=torch.float32, device=dev)
        out = torch.zeros(ws * 4, dtype=torch.float32, device=dev) if rank == 0 else None
        dist.gather(mine, gather_list=list(out.view(ws, 4).unbind(0)) if rank == 0 else None, dst=0)
        if rank == 0:
            want = torch.arange(1, ws + 1, dtype=torch.float32).repeat_interleave(4)
            bad = not torch.equal(out.cpu(), want)
    except Exception:                                 # noqa: BLE001 -- any refusal means: use the other collective
        bad = True
    _GATHER_OK[key] = not any_rank(bad)
    return _GATHER_OK[key]


def exchange_in_use():
    """The exchange gather_tiles_device performs in this process group: exchange_kind(), unless the probe of the gather failed."""
    if single():
        return exchange_kind()
    return "gather" if exchange_kind() == "gather" and _gather_probe() else "allgather"


def gather_tiles_device(local, stride):
    """The one exchange of the path on the DEVICE buffers (nccl = RCCL over xGMI): rank 0 gets a (world * stride,)
    float32 tensor on its GPU, rank r's buffer at r * stride; every other rank None.  `local`: this rank's `stride`
    float32 (what acx_grid_run filled).  Under gloo (CPU tests, ranks sharing one GPU) the buffers travel through
    host memory and the result is a CPU tensor."""
    import torch
    import torch.distributed as dist
    rank, ws = world()
    assert local.dtype == torch.float32 and local.numel() == stride
    if single():
        return local
    nccl = dist.get_backend() == "nccl"
    loc = local if nccl else local.cpu()
    if exchange_in_use() == "allgather":
        out = torch.empty(ws * stride, dtype=torch.float32, device=loc.device)
        if nccl:
            dist.all_gather_into_tensor(out, loc)
        else:
            dist.all_gather(list(out.view(ws, stride).unbind(0)), loc)
        return out if rank == 0 else None
    # reference: the joblib workers' results come back to the one parent process (algorithm_template.py:172-191)
    out = torch.empty(ws * stride, dtype=torch.float32, device=loc.device) if rank == 0 else None
    dist.gather(loc, gather_list=list(out.view(ws, stride).unbind(0)) if rank == 0 else None, dst=0)
    return out


def gather_tiles(local, stride):
    """Gather of the per-rank tile-score buffers: `local` is a torch tensor of `stride` float32
    (on the rank's GPU under nccl).  Rank 0 -- the owner of the result -- gets the gathered
    (world * stride,) float32 numpy array, every other rank None (no device-to-host copy and, since
    round 5, no world x stride allocation there: at N = 15 000 EarlyFusion that was 1.8 GB per rank for
    nothing)."""
    out = gather_tiles_device(local, stride)
    return out.cpu().numpy() if out is not None else None


# ---- pair-LIST sharding: the CPU loop of user subclasses that implement similarity() themselves
def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced [lo, hi) slice of range(n_items) for this rank."""
    base, rem = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_scores(local, n_items, device=None):
    """All-gather the per-rank score slices (float32, shape (n_local, C)) into the full
    (n_items, C) array on every rank.  One collective: ranks pad to the longest slice."""
    import torch
    import torch.distributed as dist
    rank, ws = world()
    local = np.ascontiguousarray(local, dtype=np.float32)
    if local.ndim == 1:
        local = local[:, None]
    C = local.shape[1]
    if single():
        return local
    longest = max(shard_bounds(n_items, r, ws)[1] - shard_bounds(n_items, r, ws)[0] for r in range(ws))
    buf = np.zeros((longest, C), np.float32)
    buf[:len(local)] = local
    t = torch.from_numpy(buf)
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else _collective_device())
    outs = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(outs, t)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(n_items, r, ws)
        parts.append(outs[r].cpu().numpy()[:hi - lo])
    return np.concatenate(parts, axis=0)
