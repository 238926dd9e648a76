"""The library's host routine for the local point order (mvf_kdtree_order) against
sklearn's KDTree(leaf_size=50).idx_array, which is what the reference's radius
samplers traverse (core/data_transform/transforms.py:326,380), and against the
orders stored in the MapImages fixtures (taken from the reference's run).  Host
code only: runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from mvfusion import _lib
from mvfusion.transforms import kdtree_order

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.SO_PATH), reason="libmvfusion.so not built")


def sk_order(p, cylinder, leaf_size=50):
    from sklearn.neighbors import KDTree
    p = np.asarray(p)
    return np.asarray(KDTree(p[:, :2] if cylinder else p, leaf_size=leaf_size).idx_array)


@pytest.mark.parametrize("name", ["mapimages_exact", "mapimages_exact_sphere", "mapimages_splat"])
def test_order_of_the_reference_fixtures(golden, name):
    g = golden(name)
    got = kdtree_order(torch.from_numpy(g["xyz"]), bool(g["cylinder"])).numpy()
    assert np.array_equal(got, g["order"])


@pytest.mark.parametrize("scene,cylinder", [("c1", True), ("c1", False), ("c2", True)])
def test_order_of_the_bench_clouds(scene, cylinder):
    pytest.importorskip("sklearn")
    from mvfusion import synthetic as S
    for order in ("voxel", "random"):
        xyz = S.make_scene(scene, seed=2, order=order)["xyz"]
        want = sk_order(xyz, cylinder)
        for threads in (1, 3, 8):
            got = kdtree_order(torch.from_numpy(xyz), cylinder, threads=threads).numpy()
            assert np.array_equal(got, want), (scene, order, threads)


def test_order_degenerate_clouds():
    pytest.importorskip("sklearn")
    rng = np.random.default_rng(5)
    sizes = list(range(1, 60)) + [99, 100, 101, 102, 103, 201, 202, 401, 402, 403, 801, 802, 1601, 1602,
                                  3202, 6401, 6402, 25000, 51201, 51202]
    for n in sizes:
        clouds = {
            "normal": rng.normal(size=(n, 3)),
            "duplicates": rng.integers(0, 4, size=(n, 3)),
            "voxel grid": rng.integers(0, 30, size=(n, 3)) * 0.05,
            "collinear": np.outer(rng.normal(size=n), [1.0, -2.0, 0.0]),
            "constant": np.ones((n, 3)),
            "signed zeros": np.where(rng.random((n, 3)) < 0.5, 0.0, -0.0) * rng.integers(0, 2, size=(n, 3)),
            "sorted": np.sort(rng.normal(size=(n, 3)), axis=0),
            "reversed": -np.sort(rng.normal(size=(n, 3)), axis=0),
        }
        for kind, p in clouds.items():
            p = np.ascontiguousarray(p, dtype=np.float32)
            for cylinder in (True, False):
                got = kdtree_order(torch.from_numpy(p), cylinder, threads=4).numpy()
                assert np.array_equal(got, sk_order(p, cylinder)), (n, kind, cylinder)


def test_order_is_a_permutation_without_sklearn():
    rng = np.random.default_rng(1)
    p = rng.normal(size=(70001, 3)).astype(np.float32)
    a = kdtree_order(torch.from_numpy(p), True, threads=1).numpy()
    b = kdtree_order(torch.from_numpy(p), True, threads=16).numpy()
    assert np.array_equal(np.sort(a), np.arange(p.shape[0]))
    assert np.array_equal(a, b)
    # every leaf-level half is a set split along the dimension of largest spread:
    # the first split puts the n // 2 smallest coordinates in front
    dim = int(np.argmax(p[:, :2].max(0) - p[:, :2].min(0)))
    half = p.shape[0] // 2
    assert p[a[:half], dim].max() <= p[a[half:], dim].min()


def test_bad_arguments():
    lib = _lib.load()
    out = np.empty(4, dtype=np.int64)
    p = np.zeros((4, 3), dtype=np.float32)
    assert lib.mvf_kdtree_order(p.ctypes.data, 4, 4, 3, 50, out.ctypes.data, 1) == -1
    assert lib.mvf_kdtree_order(None, 4, 2, 3, 50, out.ctypes.data, 1) == -1
    assert b"kdtree_order" in lib.mvf_last_error()
