"""Randomised parity sweep of the CSR pooling kernels and the CSR re-indexing against the CPU
oracle: segment_csr (all modes; short, long, empty segments; any channel count), the one-pass
two-level pool, segment_softmax_csr, gather_csr, ImageMapping.from_dense and select_points.
usage (GPU box, repo root): python profiles/scripts/fuzz_pools.py [cases] [first seed]"""
import sys
import time

sys.path.insert(0, "multimodal-fusion_amd")
sys.path.insert(0, "oracle")
import numpy as np
import torch

import mvf_oracle as orc
from mvfusion import mapping as M
from mvfusion.pooling import (gather_atomic_pool, gather_csr, gather_features, segment_csr,
                              segment_csr_two_level, segment_softmax_csr, sparse_interpolation)

DEV = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
N = lambda t: t.cpu().numpy()


def random_ptr(rng, nseg, kind):
    if kind == "short":
        sizes = rng.integers(0, 6, nseg)
    elif kind == "long":
        sizes = rng.integers(10, 300, nseg)
    else:
        sizes = np.where(rng.random(nseg) < 0.1, rng.integers(50, 2000, nseg), rng.integers(0, 4, nseg))
    sizes[rng.random(nseg) < 0.1] = 0
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def close(got, want, exact):
    if exact:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        scale = max(float(np.abs(want).max()), 1e-30)
        assert float(np.abs(got - want).max()) <= 2e-6 * scale + 1e-7, float(np.abs(got - want).max()) / scale


def one_case(seed):
    rng = np.random.default_rng(seed)
    what = rng.choice(["segment", "segment", "two_level", "softmax", "gather", "dense", "select",
                       "feature_maps", "interpolation"])
    if what in ("feature_maps", "interpolation"):
        B, C_ = int(rng.integers(1, 9)), int(rng.choice([1, 3, 13, 32, 70]))
        H, W = int(rng.integers(2, 90)), int(rng.integers(2, 120))
        x = rng.normal(size=(B, C_, H, W)).astype(np.float32)
        desc = f"seed {seed} {what} B={B} C={C_} {W}x{H}"
        if what == "interpolation":
            n = int(rng.integers(1, 20000))
            coords = rng.uniform(0.0, 1.0, (n, 2)).astype(np.float32)      # the reference asserts [0, 1]
            coords[rng.random(n) < 0.1] = rng.choice([0.0, 1.0, 0.5])
            batch = rng.integers(0, B, n)
            pad = str(rng.choice(["border", "zeros", "reflection"]))
            got = N(sparse_interpolation(T(x), T(coords), T(batch), padding_mode=pad))
            close(got, orc.sparse_interpolation(x, coords, batch, pad), True)
            return desc + f" n={n} {pad}"
        n_views = int(rng.integers(1, 5000))
        vptr = random_ptr(rng, n_views, "short")
        images = rng.integers(0, B, n_views)
        pix = np.stack([rng.integers(0, W, int(vptr[-1])), rng.integers(0, H, int(vptr[-1]))], 1).astype(np.int16)
        m = M.ImageMapping.from_csr_arrays(T(np.arange(n_views + 1, dtype=np.int64)), T(images), T(pix), None,
                                           view_pointers=T(vptr))
        dense = orc.gather_features(x, images, vptr, pix)
        assert np.array_equal(N(gather_features(T(x), m)), dense)
        mode = str(rng.choice(["max", "min", "mean", "sum"]))
        close(N(gather_atomic_pool(T(x), m, mode)), orc.segment_csr(dense, vptr, mode), mode in ("max", "min"))
        return desc + f" views={n_views} {mode}"
    if what in ("segment", "two_level", "softmax", "gather"):
        nseg = int(rng.integers(1, 4000))
        kind = str(rng.choice(["short", "long", "mixed"]))
        ptr = random_ptr(rng, nseg, kind)
        C_ = int(rng.choice([1, 2, 3, 4, 8, 13, 16, 31, 32, 64, 70, 96, 130]))
        rows = int(ptr[-1])
        x = rng.normal(size=(rows, C_)).astype(np.float32)
        if rng.random() < 0.3:
            x = np.round(x * 2) / 2
            x[rng.random(x.shape) < 0.1] = -0.0
        desc = f"seed {seed} {what} nseg={nseg} {kind} rows={rows} C={C_}"
        if what == "segment":
            mode = str(rng.choice(["max", "min", "sum", "mean"]))
            got = N(segment_csr(T(x), T(ptr), reduce=mode))
            close(got, orc.segment_csr(x, ptr, mode), mode in ("max", "min"))
            return desc + " " + mode
        if what == "two_level":
            n_outer = int(rng.integers(1, max(2, nseg // 2)))
            cuts = np.sort(rng.integers(0, nseg + 1, n_outer - 1)) if n_outer > 1 else np.zeros(0, np.int64)
            optr = np.concatenate([[0], cuts, [nseg]]).astype(np.int64)
            inner = str(rng.choice(["max", "min", "mean", "sum"]))
            outer = [("max", "mean"), ("min", "sum"), ("mean",), ("max",)][int(rng.integers(0, 4))]
            got = segment_csr_two_level(T(x), T(ptr), T(optr), inner=inner, outer=outer)
            mid = orc.segment_csr(x, ptr, inner)
            for g_, o_ in zip(got, outer):
                close(N(g_), orc.segment_csr(mid, optr, o_), inner in ("max", "min") and o_ in ("max", "min"))
            return desc + f" {inner}->{outer}"
        if what == "softmax":
            G = int(rng.choice([1, 2, 4, 8]))
            xs = rng.normal(size=(rows, G)).astype(np.float32) * 3
            scaling = bool(rng.random() < 0.5)
            got = N(segment_softmax_csr(T(xs), T(ptr), scaling=scaling))
            want = orc.segment_softmax_csr(xs, ptr, scaling=scaling)
            assert float(np.abs(got - want).max()) <= 2e-6 if rows else True
            return desc + f" G={G} scaling={scaling}"
        xs = rng.normal(size=(nseg, C_)).astype(np.float32)
        got = N(gather_csr(T(xs), T(ptr)))
        assert np.array_equal(got, orc.gather_csr(xs, ptr))
        return desc
    small = what == "select"               # the oracle's merge walks the groups one by one
    n_pts = int(rng.integers(1, 1200 if small else 5000))
    n_img = int(rng.integers(1, 40))
    rows = int(rng.integers(1, 4000 if small else 40000))
    F = int(rng.integers(1, 5))
    point_ids = rng.integers(0, n_pts, rows)
    image_ids = rng.integers(0, n_img, rows)
    pixels = rng.integers(0, int(rng.choice([8, 100, 3000])), (rows, 2)).astype(np.int16)
    desc = f"seed {seed} {what} pts={n_pts} img={n_img} rows={rows} F={F}"
    # the reference's contract: features are given per row and identical inside a (point, image) pair
    pair = point_ids * n_img + image_ids
    feats = rng.random((n_pts * n_img, F)).astype(np.float32)[pair]
    # distinct rows only (from_dense expects every (point, image, pixel) once)
    key = (pair * 65536 + pixels[:, 0].astype(np.int64)) * 65536 + pixels[:, 1]
    _, first = np.unique(key, return_index=True)
    first = np.sort(first)
    point_ids, image_ids, pixels, feats = point_ids[first], image_ids[first], pixels[first], feats[first]
    want = orc.mapping_from_dense(point_ids, image_ids, pixels, feats, num_points=n_pts)
    m = M.ImageMapping.from_dense(T(point_ids), T(image_ids), T(pixels), T(feats), num_points=n_pts)

    def same(m, want):
        from_ptr = want["view_pointers"]
        assert np.array_equal(N(m.pointers), want["pointers"])
        assert np.array_equal(N(m.images), want["images"])
        assert np.array_equal(N(m.values[1].pointers), from_ptr)
        px, wp = N(m.pixels).astype(np.int64), want["pixels"].astype(np.int64)
        seg = np.repeat(np.arange(from_ptr.shape[0] - 1), np.diff(from_ptr))
        o1 = np.lexsort((px[:, 1], px[:, 0], seg))
        o2 = np.lexsort((wp[:, 1], wp[:, 0], seg))
        assert np.array_equal(px[o1], wp[o2])
        np.testing.assert_allclose(N(m.features), want["features"], rtol=1e-6, atol=1e-7)

    same(m, want)
    if what == "select":
        mode = str(rng.choice(["pick", "merge"]))
        if mode == "pick":
            idx = rng.permutation(n_pts)[:int(rng.integers(1, n_pts + 1))]
        else:
            idx = rng.integers(0, max(1, n_pts // 3), n_pts)
            idx[rng.permutation(n_pts)[:int(idx.max()) + 1]] = np.arange(int(idx.max()) + 1)   # every cluster used
        got = m.select_points(T(idx.astype(np.int64)), mode=mode)
        same(got, orc.select_points(want, idx.astype(np.int64), mode))
        return desc + " " + mode
    return desc


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad, t0 = 0, time.time()
    for seed in range(first, first + cases):
        try:
            print("ok |", one_case(seed), flush=True)
        except Exception as e:          # report and go on: the sweep lists every failing seed
            import traceback
            bad += 1
            where = traceback.extract_tb(e.__traceback__)[-1]
            print("FAIL", seed, type(e).__name__, f"line {where.lineno}: {where.line}",
                  str(e)[:300].replace("\n", " "), flush=True)
    print(f"{cases} cases, {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
