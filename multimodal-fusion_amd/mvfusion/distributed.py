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


def gather_rows(t, group=None, max_padding=1.25):
    """All-gather of per-rank row blocks of different length: [n_r, ...] on
    rank r -> [sum n_r, ...] on every rank, rank-major.  One small all-gather
    of the counts, then the payload: one all-gather of blocks padded to the
    longest when that moves at most ``max_padding`` times the rows there are
    (balanced shards); otherwise every rank sends its block straight to every
    peer and receives the peers' blocks into their slices of the result, all in
    one batch of point-to-point operations -- on a fully connected xGMI node each
    pair has its own link, so an uneven partition neither travels as padding nor
    waits for one rank after the other."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t, [t.shape[0]]
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.tolist()
    cap, total = max(counts), sum(counts)
    tail = tuple(t.shape[1:])
    if world * cap <= max_padding * total:
        pad = torch.zeros((cap,) + tail, dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        out = torch.empty((world * cap,) + tail, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
        out = out.view((world, cap) + tail)
        return torch.cat([out[r, :counts[r]] for r in range(world)], dim=0), counts
    out = torch.empty((total,) + tail, dtype=t.dtype, device=t.device)
    me = dist.get_rank(group)
    t = t.contiguous()
    ops, start = [], 0
    for r in range(world):
        block = out[start:start + counts[r]]
        start += counts[r]
        if r == me:
            block.copy_(t)
            continue
        peer = dist.get_global_rank(group, r) if group is not None else r
        if counts[me]:
            ops.append(dist.P2POp(dist.isend, t, peer, group=group))
        if counts[r]:
            ops.append(dist.P2POp(dist.irecv, block, peer, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts


def rank_env():
    """(rank, world size, local rank) of this process as ``torch.distributed.run`` exports
    them (a single process without a launcher: 0, 1, 0)."""
    import os
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


class RankTimer:
    """Brackets of a timed region over all ranks: barrier + device synchronisation on
    both sides, elapsed time = MAX over the ranks."""

    def __init__(self, device=None, group=None):
        import torch.distributed as dist
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.group = group
        self.dist = dist if dist.is_initialized() and dist.get_world_size(group) > 1 else None

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(group=self.group)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def elapsed(self, t0):
        import time
        dt = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([dt], device=self.device, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
            dt = float(t.item())
        return dt


def timed_steps(run_steps, steps, warmup, finish=None, device=None, group=None):
    """The timing contract of a sharded run.  ``run_steps(k)`` performs k steps of this
    rank's share and returns what the last one produced; ``finish(out)`` is the path's
    exchange at the end of a run (the final gather of the pooled features).  Runs
    ``warmup`` untimed steps and one untimed ``finish`` (the first collective of a kind
    sets up its channels), then times exactly ``steps`` steps + ``finish`` between
    barrier + synchronise brackets.  Returns (seconds, MAX over ranks; what the timed
    ``finish`` -- or the last step -- returned)."""
    import time
    out = run_steps(warmup) if warmup > 0 else None
    if finish is not None and out is not None:
        finish(out)
    timer = RankTimer(device, group)
    timer.barrier()
    t0 = time.perf_counter()
    out = run_steps(steps)
    if finish is not None:
        out = finish(out)
    timer.barrier()
    return timer.elapsed(t0), out


def gather_pooled(features, labels=None, group=None):
    """The path's exchange step: pooled point features [N_r, C] (and labels
    [N_r]) of every rank, concatenated rank-major on every rank."""
    f, counts = gather_rows(features, group)
    if labels is None:
        return f, None, counts
    l, _ = gather_rows(labels, group)
    return f, l, counts


# --------------------------------------------------------------------------- #
# One huge scene: views sharded over the ranks                                  #
# --------------------------------------------------------------------------- #
# Views are independent for the mapping and for the pixel -> view pool; only
# the view -> point pool couples the views of a point.  Every rank therefore
# maps and pools its own block of views against the whole (replicated) cloud
# and pre-reduces its views per point; one exchange step combines the partial
# results: all-reduce MAX for the max pool, all-reduce SUM of (sum, count) for
# the mean pool.

def partition_views(num_views, rank=None, world_size=None):
    """Contiguous, balanced block of view indices of this rank (a ``range``)."""
    import torch.distributed as dist
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, extra = divmod(int(num_views), world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def local_view_pools(x_view, view_ptr, segment_csr=None):
    """Per-point partial pools over this rank's views: (max, sum, count) with
    max = -inf where the rank holds no view of the point, so that partial
    maxima of negative features survive the reduction.  ``x_view`` [Vw_r, C]
    view features in point order, ``view_ptr`` [N + 1] point -> local views."""
    if segment_csr is None:
        from .pooling import segment_csr
    count = (view_ptr[1:] - view_ptr[:-1]).to(x_view.dtype)
    mx = segment_csr(x_view, view_ptr, reduce='max')
    sm = segment_csr(x_view, view_ptr, reduce='sum')
    mx = torch.where(count[:, None] > 0, mx, torch.full_like(mx, float('-inf')))
    return mx, sm, count


def finish_view_pools(mx, sm, count):
    """(max pool, mean pool) from combined partials: points without any view
    get 0 in both (the empty-segment value of the pools); mean = sum / count."""
    seen = count[:, None] > 0
    mx = torch.where(seen, mx, torch.zeros_like(mx))
    mean = torch.where(seen, sm / count.clamp(min=1)[:, None], torch.zeros_like(sm))
    return mx, mean


def combine_view_pools(mx, sm, count, group=None):
    """The exchange step of the view-sharded layout when every rank needs the
    whole result: element-wise MAX of the partial maxima and SUM of the partial
    sums and counts over the ranks (two all-reduces: the sum and the count travel
    in one buffer).  Returns (max pool [N, C], mean pool [N, C]) on every rank.
    The max pool equals the single-device pool bit for bit; the mean differs from
    it by float32 summation order only.  Per rank a ring all-reduce sends
    2 (w - 1) / w times the two buffers (4 N (2 C + 1) bytes): see
    ``exchange_bytes_per_rank``."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        mx = mx.contiguous()
        packed = torch.cat([sm, count[:, None]], dim=1).contiguous()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        sm, count = packed[:, :-1], packed[:, -1]
    return finish_view_pools(mx, sm, count)


def _reduce_scatter_rows(t, op, n_own, group):
    """Rows [world * n_own, ...] reduced over the ranks, block r to rank r.  RCCL: one
    reduce-scatter; backends without one (gloo): all-reduce, then the own block."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    out = torch.empty((n_own,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if dist.get_backend(group) == "nccl":
        dist.reduce_scatter_tensor(out, t, op=op, group=group)
    else:
        dist.all_reduce(t, op=op, group=group)
        out.copy_(t[rank * n_own:(rank + 1) * n_own])
    return out


def combine_view_pools_scattered(mx, sm, count, group=None):
    """The exchange step when the consumer is point-sharded (every rank goes on with
    its own block of points, e.g. the 3D branch of a scene split by points): two
    reduce-scatters instead of two all-reduces, half the bytes per rank.  Points are
    dealt in equal blocks of ceil(N / world) (the last block is short).  Returns
    (range of owned points, max pool [n_own, C], mean pool [n_own, C])."""
    import torch.distributed as dist
    n = mx.shape[0]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return (range(n),) + finish_view_pools(mx, sm, count)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    block = (n + world - 1) // world
    pad = world * block - n

    def padded(t, fill):
        if not pad:
            return t.contiguous()
        return torch.cat([t, t.new_full((pad,) + tuple(t.shape[1:]), fill)])

    mx = _reduce_scatter_rows(padded(mx, float('-inf')), dist.ReduceOp.MAX, block, group)
    packed = _reduce_scatter_rows(padded(torch.cat([sm, count[:, None]], dim=1), 0.0),
                                  dist.ReduceOp.SUM, block, group)
    own = range(rank * block, min((rank + 1) * block, n))
    mx, mean = finish_view_pools(mx[:len(own)], packed[:len(own), :-1], packed[:len(own), -1])
    return own, mx, mean


def exchange_bytes_per_rank(num_points, channels, world_size, scattered=False):
    """Bytes one rank sends in the exchange step of the view-sharded max + mean pools
    (float32): the two buffers hold 4 N C and 4 N (C + 1) bytes; a ring all-reduce
    sends 2 (w - 1) / w of a buffer per rank, a reduce-scatter (w - 1) / w."""
    if world_size <= 1:
        return 0
    total = 4 * num_points * (2 * channels + 1)
    factor = (world_size - 1) / world_size
    return int(total * factor * (1 if scattered else 2))


def view_sharded_pool(x_view, view_ptr, group=None, segment_csr=None, scattered=False):
    """View -> point max and mean pools of a scene whose views are sharded over
    the ranks: local pre-reduction + ``combine_view_pools`` (every rank gets all
    points) or, ``scattered``, ``combine_view_pools_scattered`` (every rank gets its
    block of points: returns (owned range, max, mean))."""
    parts = local_view_pools(x_view, view_ptr, segment_csr)
    if scattered:
        return combine_view_pools_scattered(*parts, group=group)
    return combine_view_pools(*parts, group=group)


# --------------------------------------------------------------------------- #
# Attentive pools on a view-sharded scene: view rows travel to point owners     #
# --------------------------------------------------------------------------- #
# The attentive pools (QKVBimodalCSRPool, GroupBimodalCSRPool) normalise over
# all the views of a point, so partial results cannot be combined with a
# reduction.  Instead the points are block-partitioned over the ranks and every
# rank sends the view rows it holds (features and mapping features of its own
# views, in point order) to the rank that owns the point: one all-to-all of row
# counts and one all-to-all per row tensor.  A rank's rows of one owner are
# contiguous in its CSR, so nothing is permuted before sending.  The owner
# interleaves what it receives into one CSR, per point in source-rank order,
# which is ascending view order because the ranks hold ascending view blocks:
# the merged rows are exactly the rows of a single-device run, restricted to the
# owned points, and the pool then runs locally.

def partition_points(num_points, rank=None, world_size=None):
    """Contiguous, balanced block of point indices owned by this rank."""
    return partition_views(num_points, rank, world_size)


def merge_view_rows(counts, rows):
    """Interleave per-source CSR blocks over the same points.  ``counts[s]``
    [n] rows per point from source s, ``rows[s]`` a list of [sum counts[s], *]
    tensors.  Returns (ptr [n + 1], list of merged tensors): per point the rows
    of source 0, then source 1, ..."""
    n = counts[0].shape[0]
    dev = counts[0].device
    total = torch.stack(counts, 0).sum(0)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(total, 0)
    merged = [torch.empty((int(ptr[-1]),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
              for t in rows[0]]
    before = torch.zeros(n, dtype=torch.int64, device=dev)      # rows of earlier sources per point
    for c, block in zip(counts, rows):
        m = int(c.sum())
        src_ptr = torch.cumsum(c, 0) - c
        dest = (ptr[:-1] + before - src_ptr).repeat_interleave(c) + torch.arange(m, device=dev)
        for out, t in zip(merged, block):
            assert t.shape[0] == m, "row tensor does not match its counts"
            out[dest.to(out.device)] = t
        before = before + c
    return ptr, merged


def exchange_view_rows(view_ptr, rows, group=None):
    """The exchange step of the attentive pools on a view-sharded scene.
    ``view_ptr`` [N + 1]: point -> this rank's view rows; ``rows``: list of
    [Vw_r, *] tensors aligned with it (view features, mapping features, ...).
    Returns (owned point range, ptr [n_own + 1], merged row tensors): the
    complete view rows of the points this rank owns, views ascending."""
    import torch.distributed as dist
    n = view_ptr.shape[0] - 1
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return range(n), view_ptr, list(rows)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    blocks = [partition_points(n, r, world) for r in range(world)]
    own = blocks[rank]
    dev = view_ptr.device
    counts = (view_ptr[1:] - view_ptr[:-1]).contiguous()
    # counts of every owner's points: [n] split by owner -> [world * n_own] from every source
    recv_counts = torch.empty(world * len(own), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_counts, counts, output_split_sizes=[len(own)] * world,
                           input_split_sizes=[len(b) for b in blocks], group=group)
    recv_counts = recv_counts.view(world, len(own))
    bounds = view_ptr[torch.tensor([b.start for b in blocks] + [n], device=dev)]
    send_rows = (bounds[1:] - bounds[:-1]).tolist()
    recv_rows = recv_counts.sum(1).tolist()
    received = []
    for t in rows:
        t = t.contiguous()
        buf = torch.empty((sum(recv_rows),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_to_all_single(buf, t, output_split_sizes=recv_rows, input_split_sizes=send_rows,
                               group=group)
        received.append(buf.split(recv_rows, 0))
    ptr, merged = merge_view_rows(
        [recv_counts[s] for s in range(world)],
        [[received[k][s] for k in range(len(rows))] for s in range(world)])
    return own, ptr, merged


def view_sharded_attentive_pool(pool, x_main, x_mod, x_map, view_ptr, group=None, gather=True):
    """An attentive pool (``pool(x_main, x_mod, x_map, csr_idx)``) on a scene
    whose views are sharded over the ranks: ``exchange_view_rows``, the pool on
    the owned points, and (``gather``) an all-gather of the pooled blocks so that
    every rank holds [N, C].  ``x_main`` [N, F] is replicated like the cloud."""
    own, ptr, (x_mod, x_map) = exchange_view_rows(view_ptr, [x_mod, x_map], group)
    x_main_own = None if x_main is None else x_main[own.start:own.stop]
    out = pool(x_main_own, x_mod, x_map, ptr)
    if gather:
        out, _ = gather_rows(out, group)
    return out
