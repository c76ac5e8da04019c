"""
Multi-GPU sharding of the pair list: one process per GPU under torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Pairs are
independent, so the only exchange on the path is ONE gather of the score vectors at the
end; there is no data-path collective inside the kernels.
"""
import numpy as np

__all__ = ["world", "shard_bounds", "gather_scores"]


def world():
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


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
