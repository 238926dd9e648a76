"""The N > 1 layout on CPU: sample partition and the variable-length gather
of pooled features, world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvfusion import distributed as D


def test_partition_covers_every_sample_once_and_balances():
    rng = np.random.default_rng(0)
    sizes = rng.integers(50_000, 150_000, 64).tolist()
    for world in (1, 2, 4, 8):
        parts = D.partition_samples(sizes, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(64))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes)
    assert D.partition_samples([], 2) == [[], []]
    assert D.partition_samples([5], 2) == [[0], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [7, 3, 5, 1, 9]
        mine = D.my_samples(sizes)
        # every rank "pools" its samples: row value encodes (sample, row)
        feats = torch.cat([torch.full((sizes[i], 4), float(i)) +
                           torch.arange(sizes[i]).view(-1, 1) / 100 for i in mine]) \
            if mine else torch.zeros((0, 4))
        labels = torch.cat([torch.full((sizes[i],), i, dtype=torch.int64) for i in mine]) \
            if mine else torch.zeros(0, dtype=torch.int64)
        f, l, counts = D.gather_pooled(feats, labels)
        q.put((rank, mine, f.numpy(), l.numpy(), counts))
    finally:
        dist.destroy_process_group()


def test_gather_pooled_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sizes = [7, 3, 5, 1, 9]
    parts = [r[1] for r in res]
    assert sorted(parts[0] + parts[1]) == [0, 1, 2, 3, 4]
    # both ranks hold the same rank-major concatenation
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    want_labels = np.concatenate([np.full(sizes[i], i) for p in parts for i in p])
    assert np.array_equal(res[0][3], want_labels)
    assert res[0][4] == [sum(sizes[i] for i in p) for p in parts]
    assert np.allclose(res[0][2][:, 0] // 1, want_labels)


def test_gather_rows_single_process():
    t = torch.arange(6.).view(3, 2)
    out, counts = D.gather_rows(t)
    assert out is t and counts == [3]
