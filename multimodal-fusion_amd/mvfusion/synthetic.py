"""Seeded synthetic scenes shaped like the benchmark configurations.

Everything here is numpy-only so that the golden-vector generator, the CPU
oracle tests, the GPU parity tests and ``bench.py`` all build the very same
inputs from a seed.  Clouds are surface-like (ground disc + vertical facades on
a ring + a few free-standing walls) so that occlusion is meaningful for the
z-buffer; cameras sit on a small circle looking outward, with
``kitti360_perspective`` extrinsics (camera-to-world, rotation columns = the
camera's right / down / forward axes expressed in world coordinates).
"""
import numpy as np

__all__ = [
    "CONFIGS", "make_cloud", "make_cameras", "make_scene", "make_csr",
    "surface_normals"]

# name -> (n_points, n_views, (W, H), fx=fy, cx, cy, scene radius)
CONFIGS = {
    # KITTI-360-shaped cylinder: radius 6 m, 1408x376 perspective images
    "c1": dict(n_points=100_000, n_views=8, img_size=(1408, 376),
               f=552.55, cx=682.05, cy=238.77, radius=6.0),
    # WildScenes-shaped scan: 2016x1512 images
    "c2": dict(n_points=300_000, n_views=16, img_size=(2016, 1512),
               f=1400.0, cx=1008.0, cy=756.0, radius=15.0),
    # dense stress: 4 MP images
    "c5": dict(n_points=2_000_000, n_views=128, img_size=(2304, 1728),
               f=1600.0, cx=1152.0, cy=864.0, radius=20.0),
}

# visibility parameters of the shipped KITTI-360 / WildScenes pipelines
VISIBILITY_DEFAULTS = dict(
    voxel=0.05, r_max=20, r_min=0, k_swell=1.5, d_swell=1e6, exact=True,
    camera="kitti360_perspective")


def make_cloud(n_points, radius, rng, order="random", voxel=0.05):
    """Surface-like cloud, float32 [n, 3]: 55 % ground disc, 30 % facade ring,
    15 % free-standing walls (occluders) at mid radius.  ``order``: 'random'
    (shuffled; the default, which the committed golden vectors were made with)
    or 'voxel' (points sorted by the cell of a ``voxel``-sized grid, as after
    grid sampling)."""
    n_ground = int(0.55 * n_points)
    n_facade = int(0.30 * n_points)
    n_wall = n_points - n_ground - n_facade

    # ground: uniform in area over the disc, slightly rough
    r = radius * np.sqrt(rng.random(n_ground))
    a = 2 * np.pi * rng.random(n_ground)
    ground = np.stack(
        [r * np.cos(a), r * np.sin(a), rng.normal(0.0, 0.02, n_ground)], 1)

    # facades: vertical cylinder shell at 0.9 radius, 0..4 m high
    a = 2 * np.pi * rng.random(n_facade)
    rf = 0.9 * radius + rng.normal(0.0, 0.02, n_facade)
    facade = np.stack(
        [rf * np.cos(a), rf * np.sin(a), 4.0 * rng.random(n_facade)], 1)

    # walls: 6 planar patches at 0.45 radius, tangent to the ring, 2 m high
    k = rng.integers(0, 6, n_wall)
    ang = 2 * np.pi * k / 6 + 0.3
    u = (rng.random(n_wall) - 0.5) * 0.5 * radius      # along the wall
    c = 0.45 * radius
    wall = np.stack(
        [c * np.cos(ang) - u * np.sin(ang),
         c * np.sin(ang) + u * np.cos(ang),
         2.0 * rng.random(n_wall)], 1)

    xyz = np.concatenate([ground, facade, wall], 0)
    xyz = xyz[rng.permutation(n_points)]
    if order == "voxel":
        # The order a voxel-grid sampler leaves the points in: by ascending
        # linear voxel index, x fastest (the reference's GridSampling3D, the
        # first pre-transform of every shipped pipeline, groups the points by
        # consecutive cluster id: core/data_transform/grid_transform.py:125-140).
        cell = np.floor(xyz / voxel).astype(np.int64)
        xyz = xyz[np.lexsort((cell[:, 0], cell[:, 1], cell[:, 2]))]
    return xyz.astype(np.float32)


def surface_normals(xyz):
    """Cheap analytic per-point unit 'normals' (float32 [n, 3]) and three
    geometric attributes in [0, 1] (linearity, planarity, scattering) derived
    from a hash of the coordinates: they only need to be deterministic
    per-point payloads for the mapping-feature columns."""
    h = np.abs(np.sin(xyz.astype(np.float64) @ np.array([12.9898, 78.233, 37.719])) * 43758.5453)
    h = h - np.floor(h)
    lin = h.astype(np.float32)
    pla = ((h * 7.0) % 1.0).astype(np.float32)
    sca = ((h * 13.0) % 1.0).astype(np.float32)
    nrm = np.stack([np.cos(6.28 * h), np.sin(6.28 * h), 0.5 + 0 * h], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return nrm.astype(np.float32), lin, pla, sca


def make_cameras(n_views, f, cx, cy, ring=0.5, height=1.5, yaw0=0.0):
    """Returns dict of per-view camera arrays:
    pos float64 [V,3]; extrinsic float32 [V,4,4] (camera-to-world);
    fx, fy, mx, my float32 [V]."""
    yaw = yaw0 + 2 * np.pi * np.arange(n_views) / n_views
    c, s = np.cos(yaw), np.sin(yaw)
    pos = np.stack([ring * c, ring * s, np.full(n_views, height)], 1)
    ext = np.zeros((n_views, 4, 4), dtype=np.float64)
    # rotation columns: right, down, forward (right x down = forward)
    ext[:, 0, 0], ext[:, 1, 0], ext[:, 2, 0] = s, -c, 0.0
    ext[:, 0, 1], ext[:, 1, 1], ext[:, 2, 1] = 0.0, 0.0, -1.0
    ext[:, 0, 2], ext[:, 1, 2], ext[:, 2, 2] = c, s, 0.0
    ext[:, :3, 3] = pos
    ext[:, 3, 3] = 1.0
    ext = ext.astype(np.float32)
    one = np.ones(n_views, dtype=np.float32)
    return dict(
        pos=ext[:, :3, 3].astype(np.float64), extrinsic=ext,
        fx=one * np.float32(f), fy=one * np.float32(f),
        mx=one * np.float32(cx), my=one * np.float32(cy))


def make_scene(name=None, seed=0, **overrides):
    """Cloud + cameras for a named configuration (or explicit overrides)."""
    cfg = dict(CONFIGS[name]) if name is not None else {}
    cfg.update(overrides)
    rng = np.random.default_rng(seed)
    xyz = make_cloud(cfg["n_points"], cfg["radius"], rng, order=cfg.get("order", "random"))
    cams = make_cameras(cfg["n_views"], cfg["f"], cfg["cx"], cfg["cy"])
    # keep every point at least 1 cm away from every camera centre
    d = np.linalg.norm(
        xyz[:, None, :].astype(np.float64) - cams["pos"][None], axis=2).min(1)
    xyz[d < 0.01] += np.float32(0.05)
    return dict(xyz=xyz, img_size=tuple(cfg["img_size"]), **cams)


def make_csr(n_groups, mean_size, max_size, rng, empty_frac=0.1):
    """Random CSR pointers int64 [n_groups + 1]: sizes ~ min(Poisson(mean),
    max_size), a fraction of groups forced empty."""
    sizes = np.minimum(rng.poisson(mean_size, n_groups), max_size)
    sizes[rng.random(n_groups) < empty_frac] = 0
    ptr = np.zeros(n_groups + 1, dtype=np.int64)
    np.cumsum(sizes, out=ptr[1:])
    return ptr
