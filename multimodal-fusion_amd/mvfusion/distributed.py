"""Multi-GPU layout of the fusion path: one process per GPU, samples
(cylinders with their images) are independent, so every rank maps and pools
its own block of samples and the only exchange is the final gather of the
pooled point features and labels (RCCL all-gather; ``gloo`` on CPU tensors in
the tests).  The reference processes samples one by one and concatenates at
collate time (core/multimodal/data.py:180-204, csr.py:347-416).
"""
import torch


def partition_samples(sizes, world_size):
    """Greedy longest-first assignment of samples to ranks, balanced by point
    count.  Returns a list (one entry per rank) of ascending sample indices;
    every sample appears exactly once."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(sizes[i])
    return [sorted(p) for p in parts]


def my_samples(sizes, rank=None, world_size=None):
    import torch.distributed as dist
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return partition_samples(sizes, world_size)[rank]


def gather_rows(t, group=None):
    """All-gather of per-rank row blocks of different length: [n_r, ...] on
    rank r -> [sum n_r, ...] on every rank, rank-major.  One small all-gather
    of the counts, then one padded all-gather of the payload."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t, [t.shape[0]]
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.tolist()
    cap = max(counts)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = torch.empty((world * cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.view((world, cap) + tuple(t.shape[1:]))
    return torch.cat([out[r, :counts[r]] for r in range(world)], dim=0), counts


def gather_pooled(features, labels=None, group=None):
    """The path's exchange step: pooled point features [N_r, C] (and labels
    [N_r]) of every rank, concatenated rank-major on every rank."""
    f, counts = gather_rows(features, group)
    if labels is None:
        return f, None, counts
    l, _ = gather_rows(labels, group)
    return f, l, counts
