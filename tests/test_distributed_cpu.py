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
        # uneven blocks (20 rows against 1, then against none): travel as one
        # broadcast per rank instead of blocks padded to the longest
        for n_other in (1, 0):
            mine_rows = 20 if rank == 0 else n_other
            t = torch.arange(mine_rows * 3, dtype=torch.float32).view(mine_rows, 3) + 1000 * rank
            u, c = D.gather_rows(t)
            assert c == [20, n_other] and u.shape == (20 + n_other, 3)
            assert torch.equal(u[:20], torch.arange(60, dtype=torch.float32).view(20, 3))
            if n_other:
                assert torch.equal(u[20:], torch.arange(3, dtype=torch.float32).view(1, 3) + 1000)
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


# ------------------------------------------------------- view-sharded scene
def _np_segment_csr(src, ptr, reduce="sum"):
    import mvf_oracle as orc
    return torch.from_numpy(orc.segment_csr(src.numpy(), ptr.numpy(), reduce))


def _view_scene():
    """Views of a scene in (point, view) order: per point the ids of the views
    that see it, with signed features so that empty partial maxima matter."""
    rng = np.random.default_rng(7)
    n_pts, n_views, C = 500, 12, 5
    sees = rng.random((n_pts, n_views)) < 0.3
    sees[:25] = False                                     # points no view sees
    feats = rng.normal(size=(n_pts, n_views, C)).astype(np.float32) - 0.5
    return sees, feats


def _local_csr(sees, feats, views):
    sub = sees[:, list(views)]
    ptr = np.concatenate([[0], np.cumsum(sub.sum(1))]).astype(np.int64)
    rows = feats[:, list(views)][sub]
    return torch.from_numpy(np.ascontiguousarray(rows)), torch.from_numpy(ptr)


def _view_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sees, feats = _view_scene()
        views = D.partition_views(sees.shape[1])
        x, ptr = _local_csr(sees, feats, views)
        mx, mean = D.view_sharded_pool(x, ptr, segment_csr=_np_segment_csr)
        # the point-sharded consumer: every rank keeps its block of points only
        own, mx_own, mean_own = D.view_sharded_pool(x, ptr, segment_csr=_np_segment_csr, scattered=True)
        assert torch.equal(mx_own, mx[own.start:own.stop]) and torch.equal(mean_own, mean[own.start:own.stop])
        q.put((rank, list(views), mx.numpy(), mean.numpy(), [own.start, own.stop]))
    finally:
        dist.destroy_process_group()


def test_partition_views():
    for world in (1, 2, 3, 8):
        blocks = [D.partition_views(13, r, world) for r in range(world)]
        assert [v for b in blocks for v in b] == list(range(13))
        assert max(len(b) for b in blocks) - min(len(b) for b in blocks) <= 1
    assert len(D.partition_views(2, 3, 4)) == 0


def test_view_sharded_pool_world2_gloo():
    import mvf_oracle as orc
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_view_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] + res[1][1] == list(range(12))
    sees, feats = _view_scene()
    x, ptr = _local_csr(sees, feats, range(12))
    want_max = orc.segment_csr(x.numpy(), ptr.numpy(), "max")
    want_mean = orc.segment_csr(x.numpy(), ptr.numpy(), "mean")
    for r in res:
        assert np.array_equal(r[2], want_max)             # bit-exact, empties -> 0
        np.testing.assert_allclose(r[3], want_mean, rtol=1e-5, atol=1e-6)
    assert np.array_equal(res[0][3], res[1][3])
    assert (want_max[25:] < 0).any() and not want_max[:25].any()
    assert res[0][4] == [0, 250] and res[1][4] == [250, 500]
    # bytes a rank sends in the exchange: reduce-scatter moves half of what all-reduce does
    assert D.exchange_bytes_per_rank(2_000_000, 64, 8) == 2 * D.exchange_bytes_per_rank(2_000_000, 64, 8, True)
    assert D.exchange_bytes_per_rank(2_000_000, 64, 8, True) == int(4 * 2_000_000 * 129 * 7 / 8)
    assert D.exchange_bytes_per_rank(10, 4, 1) == 0


def test_combine_view_pools_scattered_single_process_and_uneven_blocks():
    sees, feats = _view_scene()
    x, ptr = _local_csr(sees, feats, range(12))
    parts = D.local_view_pools(x, ptr, _np_segment_csr)
    own, mx, mean = D.combine_view_pools_scattered(*parts)
    want = D.combine_view_pools(*parts)
    assert own == range(500) and torch.equal(mx, want[0]) and torch.equal(mean, want[1])


# ------------------------------------------------------------ timing harness
def _timed_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        assert D.rank_env() == (rank, world, rank)
        calls = []
        n_rows = 5 + 3 * rank

        def run_steps(k):
            calls.append(k)
            time.sleep(0.02 * k * (1 + 2 * rank))             # rank 1 is three times slower
            return torch.full((n_rows, 2), float(rank)), torch.full((n_rows,), rank, dtype=torch.int64)

        seconds, (f, l, counts) = D.timed_steps(
            run_steps, steps=4, warmup=2, finish=lambda out: D.gather_pooled(out[0], out[1]))
        q.put((rank, calls, seconds, f.numpy(), l.numpy(), counts))
    finally:
        dist.destroy_process_group()


def test_timed_steps_world2_gloo():
    """The function bench.py times its steps with: warm-up steps and one untimed exchange,
    then exactly K steps + the final gather between barriers, the slowest rank's time on
    every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_timed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] == [2, 4]                                  # warm-up, then exactly K steps
        assert r[5] == [5, 8] and r[3].shape == (13, 2)
        assert np.array_equal(r[4], np.array([0] * 5 + [1] * 8))
    assert res[0][2] == res[1][2] >= 0.02 * 4 * 3              # MAX over ranks, the same on both


def test_timed_steps_single_process():
    seen = []
    seconds, out = D.timed_steps(lambda k: seen.append(k) or k, steps=3, warmup=0)
    assert seen == [3] and out == 3 and seconds >= 0.0


# ------------------------------------------------- view rows to point owners
def _rows_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sees, feats = _view_scene()
        views = D.partition_views(sees.shape[1])
        x, ptr = _local_csr(sees, feats, views)
        view_id = torch.from_numpy(np.broadcast_to(np.arange(sees.shape[1]), sees.shape)[:, list(views)][
            sees[:, list(views)]].copy())
        own, optr, (ox, oid) = D.exchange_view_rows(ptr, [x, view_id])
        # a pool stand-in that needs all the views of a point at once: softmax-weighted sum
        def pool(x_main, x_mod, x_map, csr):
            out = torch.zeros(csr.shape[0] - 1, x_mod.shape[1], dtype=torch.float64)
            for s in range(csr.shape[0] - 1):
                a, b = int(csr[s]), int(csr[s + 1])
                if b > a:
                    w = torch.softmax(x_mod[a:b, 0].double(), 0)
                    out[s] = (x_mod[a:b].double() * w[:, None]).sum(0) + x_main[s, 0].double()
            return out
        x_main = torch.arange(sees.shape[0], dtype=torch.float32)[:, None]
        pooled = D.view_sharded_attentive_pool(pool, x_main, x, view_id[:, None].float(), ptr)
        q.put((rank, [own.start, own.stop], optr.numpy(), ox.numpy(), oid.numpy(), pooled.numpy()))
    finally:
        dist.destroy_process_group()


def test_merge_view_rows_orders_sources_per_point():
    c0, c1 = torch.tensor([2, 0, 1, 0]), torch.tensor([1, 0, 0, 3])
    r0, r1 = torch.tensor([10., 11., 12.]), torch.tensor([20., 21., 22., 23.])
    ptr, (rows,) = D.merge_view_rows([c0, c1], [[r0], [r1]])
    assert ptr.tolist() == [0, 3, 3, 4, 7]
    assert rows.tolist() == [10., 11., 20., 12., 21., 22., 23.]
    own, p, rws = D.exchange_view_rows(ptr, [rows])          # no process group: identity
    assert list(own) == [0, 1, 2, 3] and p is ptr and rws[0] is rows
    assert [list(D.partition_points(10, r, 3)) for r in range(3)] == [[0, 1, 2, 3], [4, 5, 6], [7, 8, 9]]


def test_exchange_view_rows_world2_gloo():
    """Views sharded over two ranks, points block-partitioned: after the exchange
    every owner holds exactly the rows of a single-process CSR of its points, in
    ascending view order, and a pool that normalises over all views of a point
    returns the single-process result on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sees, feats = _view_scene()
    V = sees.shape[1]
    x, ptr = _local_csr(sees, feats, range(V))
    x, ptr = x.numpy(), ptr.numpy()
    ids = np.broadcast_to(np.arange(V), sees.shape)[sees]
    n = sees.shape[0]
    assert [r[1] for r in res] == [[0, (n + 1) // 2], [(n + 1) // 2, n]]
    for _, (a, b), optr, ox, oid, pooled in res:
        assert np.array_equal(optr, ptr[a:b + 1] - ptr[a])
        assert np.array_equal(ox, x[ptr[a]:ptr[b]])
        assert np.array_equal(oid, ids[ptr[a]:ptr[b]])
    want = np.zeros((n, x.shape[1]))
    for s in range(n):
        a, b = ptr[s], ptr[s + 1]
        if b > a:
            w = torch.softmax(torch.from_numpy(x[a:b, 0]).double(), 0).numpy()
            want[s] = (x[a:b].astype(np.float64) * w[:, None]).sum(0) + s
    for r in res:
        np.testing.assert_allclose(r[5], want, rtol=1e-12, atol=1e-12)
