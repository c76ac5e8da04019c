"""
Multi-GPU execution of the pair grid: one process per GPU under torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Pairs are independent; the grid is
cut into B x B track tiles that libacx deals to the ranks by cost (acx_grid_plan), every rank
writes the scores of its tiles into ONE dense device buffer (acx_grid_run), and the only exchange
on the path is ONE all-gather of those buffers -- device to device under
RCCL, no host staging.  Rank 0 then scatters the tiles into the N x N matrices.
"""
import numpy as np

__all__ = ["world", "barrier", "broadcast_object", "shard_bounds", "gather_scores", "gather_tiles"]


def world():
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def barrier():
    rank, ws = world()
    if ws > 1:
        import torch.distributed as dist
        dist.barrier()


def broadcast_object(obj, src=0):
    """Small picklable object from rank `src` to every rank."""
    rank, ws = world()
    if ws == 1:
        return obj
    import torch.distributed as dist
    box = [obj if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def gather_tiles(local, stride):
    """All-gather of the per-rank tile-score buffers: `local` is a torch tensor of `stride` float32
    (on the rank's GPU under nccl).  Returns the gathered (world * stride,) float32 numpy array.
    Under nccl the collective runs on the device buffers themselves (one D2H copy of the result);
    under gloo (CPU tests, ranks sharing one GPU) the buffer is moved to the host first."""
    import torch
    import torch.distributed as dist
    rank, ws = world()
    assert local.dtype == torch.float32 and local.numel() == stride
    if ws == 1:
        return local.cpu().numpy()
    if dist.get_backend() == "nccl":
        out = torch.empty(ws * stride, dtype=torch.float32, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out.cpu().numpy()
    loc = local.cpu()
    outs = [torch.empty_like(loc) for _ in range(ws)]
    dist.all_gather(outs, loc)
    return torch.cat(outs).numpy()


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
    if ws == 1:
        return local
    longest = max(shard_bounds(n_items, r, ws)[1] - shard_bounds(n_items, r, ws)[0] for r in range(ws))
    buf = np.zeros((longest, C), np.float32)
    buf[:len(local)] = local
    t = torch.from_numpy(buf)
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    outs = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(outs, t)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(n_items, r, ws)
        parts.append(outs[r].cpu().numpy()[:hi - lo])
    return np.concatenate(parts, axis=0)
