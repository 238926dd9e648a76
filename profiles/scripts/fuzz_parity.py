"""Randomised parity sweep of MapImages (fused passes and the batched pass) against the CPU
oracle: random scene sizes, view counts, image sizes that are no multiple of the tile size,
crops, masks, scale factors, sampling radii, swell parameters, exact / plain splat visibility,
depth-map visibility, mapping ids with gaps, point attributes on / off.
A third argument "wide" draws fisheye / equirectangular cameras instead (per-view path).
usage (GPU box, repo root): python profiles/scripts/fuzz_parity.py [cases] [first seed] [wide]"""
import sys
import time

sys.path.insert(0, "multimodal-fusion_amd")
sys.path.insert(0, "oracle")
sys.path.insert(0, "tests")
import numpy as np
import torch

import mvf_oracle as orc
from mvfusion import synthetic as S
from mvfusion import visibility as V
from mvfusion.images import SameSettingImageBatch, SameSettingImageData
from mvfusion.transforms import MapImages, kdtree_order
from test_gpu_parity import Data, check_mapping

T = torch.from_numpy


WIDE_ONLY = False
FISHEYE = np.array([2.2134, 0.016, 0.4325, 1336.3, 1335.8, 716.9, 705.8], np.float32)


def wide_case(seed, rng):
    """MapImages with fisheye / equirectangular images (view by view on the device)."""
    cam = str(rng.choice(["kitti360_fisheye", "s3dis_equirectangular"]))
    n_points, n_views = int(rng.integers(500, 40000)), int(rng.integers(1, 7))
    sc = S.make_scene("c1", seed=seed, n_points=n_points, n_views=n_views,
                      radius=float(rng.uniform(3, 20)))
    cams = dict(pos=sc["pos"])
    if cam == "kitti360_fisheye":
        side = int(rng.integers(300, 1500)) // 2 * 2
        size = (side, side)
        fe = FISHEYE.copy()
        fe[3:] *= side / 1400.0
        fe[:3] *= np.float32(rng.uniform(0.9, 1.1))
        cams["extrinsic"] = sc["extrinsic"]
        cams["fisheye"] = np.tile(fe, (n_views, 1)).astype(np.float32)
    else:
        h = int(rng.integers(100, 600)) // 2 * 2
        size = (2 * h, h)
        yaw = rng.uniform(0, 2 * np.pi, n_views)
        cams["opk"] = np.stack([rng.uniform(-0.1, 0.1, n_views), rng.uniform(-0.1, 0.1, n_views), yaw],
                               1).astype(np.float32)
    up = int(rng.choice([1, 2]))
    ref_size = (size[0] // up, size[1] // up)
    exact = bool(rng.random() < 0.6)
    cylinder = bool(rng.random() < 0.7)
    vis = dict(r_max=float(rng.uniform(4, 25)), r_min=float(rng.choice([0, 0.5])), camera=cam, exact=exact,
               voxel=float(rng.choice([0.05, 0.1])), k_swell=float(rng.uniform(0.5, 2)), d_swell=1e6)
    desc = f"seed {seed} wide {cam} N={n_points} V={n_views} {size} up={up} exact={exact} cyl={cylinder}"
    N = sc["xyz"].shape[0]
    data = Data()
    data.pos = T(sc["xyz"])
    data.mapping_index = torch.arange(N)
    if "fisheye" in cams:
        f = T(cams["fisheye"])
        extra = dict(extrinsic=T(cams["extrinsic"]), xi=f[:, 0], k1=f[:, 1], k2=f[:, 2],
                     gamma1=f[:, 3], gamma2=f[:, 4], u0=f[:, 5], v0=f[:, 6])
    else:
        extra = dict(opk=T(cams["opk"]))
    images = SameSettingImageData(pos=T(cams["pos"]), ref_size=ref_size, proj_upscale=up, **extra)
    if rng.random() < 0.25 and up == 1:
        # depth-map visibility on maps rendered from the cloud itself
        dv = {k: vis[k] for k in ("r_max", "r_min", "camera")}
        thr = float(rng.choice([0.05, 0.2]))
        kw = {k: vis[k] for k in ("voxel", "k_swell", "d_swell", "r_max", "r_min")}
        maps = []
        for v in range(n_views):
            mk = {}
            if "extrinsic" in cams:
                mk["img_extrinsic"] = cams["extrinsic"][v]
            if "fisheye" in cams:
                mk["img_intrinsic_fisheye"] = cams["fisheye"][v]
            if "opk" in cams:
                mk["img_opk"] = cams["opk"][v]
            img_xyz = cams["pos"][v].astype(np.float32)
            i1, dist, x, y = orc.camera_projection(sc["xyz"], img_xyz, img_size=size, camera=cam,
                                                   r_max=vis["r_max"], r_min=vis["r_min"], **mk)
            dm = np.full(size, -1.0, np.float32)
            if i1.shape[0]:
                i2, xp, yp = orc.visibility_from_splatting(
                    x, y, dist, sc["xyz"][i1], exact=False, img_size=size, camera=cam,
                    voxel=vis["voxel"], k_swell=vis["k_swell"], d_swell=vis["d_swell"], **mk)
                dm[xp, yp] = np.round(dist[i2] * np.float32(512)) / np.float32(512)
            maps.append(dm)
        images.depth_maps = [T(m).cuda() for m in maps]
        t = MapImages(method="DepthBasedVisibility", cylinder=cylinder, point_order="index",
                      depth_threshold=thr, **dv)
        assert t._fusable(images, t._settle(data, images))
        try:
            _, out = t(data, images)
        except ValueError as e:
            return desc, "no mapping (%s)" % str(e)[:30]
        ref = orc.map_images(sc["xyz"], np.arange(N), cams, order=None, ref_size=ref_size, cylinder=cylinder,
                             method="depth_map", depth_maps=maps, depth_threshold=thr, **dv)
        check_mapping(out.mappings, ref)
        return desc + " depth", "ok %d mappings" % ref["images"].shape[0]
    t = MapImages(method="SplattingVisibility", cylinder=cylinder, point_order="index", **vis)
    fused = t._fusable(images, t._settle(data, images))
    try:
        _, out = t(data, images)
    except ValueError as e:
        return desc, "no mapping (%s)" % str(e)[:30]
    ref = orc.map_images(sc["xyz"], np.arange(N), cams, order=None, ref_size=ref_size, proj_upscale=up,
                         cylinder=cylinder, **vis)
    check_mapping(out.mappings, ref)
    return desc + (" fused" if fused else " per-view"), \
        "ok %d rows %d pixels" % (ref["images"].shape[0], ref["pixels"].shape[0])


def one_case(seed):
    rng = np.random.default_rng(seed)
    mode = rng.choice(["exact", "exact", "exact", "splat", "depth", "batch"] if not WIDE_ONLY else ["wide"])
    if mode == "wide":
        return wide_case(seed, rng)
    W = int(rng.integers(90, 700))
    H = int(rng.integers(70, 500))
    n_views = int(rng.integers(1, 40 if mode != "splat" else 8))
    n_points = int(rng.integers(500, 60000 if mode != "splat" else 15000))
    f = float(rng.uniform(0.4, 1.2) * W)
    cfg = dict(n_points=n_points, n_views=n_views, img_size=(W, H), f=f, cx=W / 2 + rng.uniform(-9, 9),
               cy=H / 2 + rng.uniform(-9, 9), radius=float(rng.uniform(3, 25)),
               order=str(rng.choice(["random", "voxel"])))
    cylinder = bool(rng.random() < 0.7)
    vis = dict(voxel=float(rng.choice([0.02, 0.05, 0.1, 0.3])), r_max=float(rng.uniform(4, 30)),
               r_min=float(rng.choice([0, 0, 0.5, 2.0])), k_swell=float(rng.uniform(0.5, 2.5)),
               d_swell=float(rng.choice([1e6, 1000.0, 50.0])), camera="kitti360_perspective")
    if rng.random() < 0.3 and H > 120:
        vis.update(crop_top=int(rng.integers(0, 30)), crop_bottom=int(rng.integers(0, 30)))
    attrs = bool(rng.random() < 0.5)
    desc = f"seed {seed} {mode} N={n_points} V={n_views} {W}x{H} cyl={cylinder} attrs={attrs} {vis}"

    def sample(k):
        sc = S.make_scene(None, seed=seed * 10 + k, **cfg)
        n = sc["xyz"].shape[0]
        ids = np.arange(n) if rng.random() < 0.6 else np.sort(rng.permutation(n + 200)[:n])
        if rng.random() < 0.5:
            ids = rng.permutation(ids)
        return sc, ids

    def inputs(sc, ids, mask, kw_img):
        data = Data()
        data.pos = T(sc["xyz"])
        data.mapping_index = T(ids.astype(np.int64))
        extra = {}
        if attrs:
            nrm, lin, pla, sca = S.surface_normals(sc["xyz"])
            data.linearity, data.planarity, data.scattering, data.norm = T(lin), T(pla), T(sca), T(nrm)
            extra = dict(linearity=lin, planarity=pla, scattering=sca, normals=nrm)
        images = SameSettingImageData(
            pos=T(sc["pos"]), extrinsic=T(sc["extrinsic"]), fx=T(sc["fx"]), fy=T(sc["fy"]),
            mx=T(sc["mx"]), my=T(sc["my"]), ref_size=kw_img.pop("ref_size"), mask=mask, **kw_img)
        return data, images, extra

    mask_np = None
    if mode in ("exact", "batch") and rng.random() < 0.3:
        mask_np = rng.random((W, H)) > 0.2
    mask = None if mask_np is None else T(mask_np)

    if mode == "depth":
        sc, ids = sample(0)
        K = np.tile(np.eye(4, dtype=np.float32), (n_views, 1, 1))
        K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = sc["fx"], sc["fy"], sc["mx"], sc["my"]
        dv = {k: vis[k] for k in ("r_max", "r_min", "camera")}
        thr = float(rng.choice([0.02, 0.05, 0.2]))
        data, images, extra = inputs(sc, ids, None, dict(ref_size=(W, H), proj_upscale=1))
        kw = {k: vis[k] for k in ("voxel", "k_swell", "d_swell", "camera", "r_max", "r_min")}
        images.depth_maps = V.render_depth_maps(
            data.pos.cuda(), images.pos, T(K), images.extrinsic, (W, H), **kw)
        maps = [m.cpu().numpy() for m in images.depth_maps]
        t = MapImages(method="DepthBasedVisibility", cylinder=cylinder, depth_threshold=thr, **dv)
        try:
            _, out = t(data, images)
        except ValueError as e:
            return desc, "no mapping (%s)" % str(e)[:30]
        ref = orc.map_images(sc["xyz"], ids, sc, order=None, ref_size=(W, H), cylinder=cylinder,
                             method="depth_map", depth_maps=maps, depth_threshold=thr, **extra, **dv)
        check_mapping(out.mappings, ref)
        return desc, "ok %d mappings" % ref["images"].shape[0]

    exact = mode != "splat"
    kw_img, kw_orc = dict(ref_size=(W, H), proj_upscale=1), {}
    if mode == "exact" and rng.random() < 0.3 and mask is None and "crop_top" not in vis:
        up, down = int(rng.choice([1, 2])), int(rng.choice([1, 2]))
        ref_size = (W // up, H // up)
        cs = (ref_size[0] - 2 * int(rng.integers(0, 9)) * down, ref_size[1] - 2 * int(rng.integers(0, 5)) * down)
        cs = (cs[0] // down * down, cs[1] // down * down)
        offs = np.stack([rng.integers(0, ref_size[0] - cs[0] + 1, n_views),
                         rng.integers(0, ref_size[1] - cs[1] + 1, n_views)], 1).astype(np.int64)
        W_, H_ = ref_size[0] * up, ref_size[1] * up
        if (W_, H_) == (W, H):
            kw_img = dict(ref_size=ref_size, proj_upscale=up, downscale=down, crop_size=cs, crop_offsets=T(offs))
            kw_orc = dict(proj_upscale=up, downscale=down, crop_size=cs, crop_offsets=offs)
    t = MapImages(method="SplattingVisibility", cylinder=cylinder, exact=exact, **vis)
    n_samples = int(rng.integers(2, 5)) if mode == "batch" else 1
    samples = [sample(k) for k in range(n_samples)]
    refs, pairs = [], []
    for sc, ids in samples:
        order = kdtree_order(T(sc["xyz"]), cylinder).numpy()
        data, images, extra = inputs(sc, ids, mask, dict(kw_img))
        data.mvf_point_order = T(order)
        pairs.append((data, images))
        refs.append(lambda sc=sc, ids=ids, order=order, extra=extra: orc.map_images(
            sc["xyz"], ids, sc, order=order, ref_size=kw_img["ref_size"], cylinder=cylinder,
            img_mask=mask_np, exact=exact, **kw_orc, **extra, **vis))
    if mode != "batch":
        try:
            _, out = t(*pairs[0])
        except ValueError as e:
            return desc, "no mapping (%s)" % str(e)[:30]
        ref = refs[0]()
        check_mapping(out.mappings, ref)
        return desc, "ok %d rows %d pixels" % (ref["images"].shape[0], ref["pixels"].shape[0])
    try:
        _, got = t.process_batch([p[0] for p in pairs], [p[1] for p in pairs])
    except ValueError as e:
        return desc, "no mapping (%s)" % str(e)[:30]
    assert isinstance(got, SameSettingImageBatch)
    parts = got.mappings.to_csr_list()
    total = 0
    for m, r in zip(parts, refs):
        ref = r()
        check_mapping(m, ref)
        total += ref["images"].shape[0]
    return desc, "ok batch of %d, %d rows" % (n_samples, total)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    WIDE_ONLY = len(sys.argv) > 3 and sys.argv[3] == "wide"
    bad = 0
    t0 = time.time()
    for seed in range(first, first + cases):
        try:
            desc, res = one_case(seed)
            print(res, "|", desc, flush=True)
        except Exception as e:          # report and go on: the sweep lists every failing seed
            bad += 1
            print("FAIL", seed, type(e).__name__, str(e)[:300].replace("\n", " "), flush=True)
    print(f"{cases} cases, {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
