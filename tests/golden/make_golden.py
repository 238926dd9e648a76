"""Generate the golden vectors of tests/golden/*.npz from the reference itself.

Run in the build container only (``python tests/golden/make_golden.py``): it
executes the reference's own source files from /root/reference through the
loader in ``_refshim.py`` on seeded synthetic inputs and stores inputs and
outputs as compressed .npz.  The fixtures are data only; tests never touch the
reference.
"""
import os
import sys
import tempfile

import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "multimodal-fusion_amd"))

import _refshim  # noqa: E402
from mvfusion import synthetic as S  # noqa: E402

ref = _refshim.load_reference()
vis = ref["visibility"]
T = torch.from_numpy


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def intrinsic(sc, v=0):
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = sc["fx"][v], sc["fy"][v], sc["mx"][v], sc["my"][v]
    return K


def attrs(xyz):
    nrm, lin, pla, sca = S.surface_normals(xyz)
    return nrm, lin, pla, sca


# ---------------------------------------------------------------- projection
def gen_projection():
    sc = S.make_scene("c1", seed=0, n_points=20000)
    xyz, W, H = sc["xyz"], *sc["img_size"]
    rng = np.random.default_rng(100)
    mask = rng.random((W, H)) > 0.2
    cases = {
        "plain": dict(v=0, crop_top=0, crop_bottom=0, r_max=20, r_min=0, mask=None,
                      camera="kitti360_perspective"),
        "crop_mask": dict(v=3, crop_top=40, crop_bottom=25, r_max=8, r_min=2, mask=mask,
                          camera="kitti360_perspective"),
        "scannet": dict(v=5, crop_top=0, crop_bottom=10, r_max=20, r_min=0.5, mask=None,
                        camera="scannet"),
        "tiny": dict(v=1, crop_top=0, crop_bottom=0, r_max=20, r_min=0, mask=None,
                     camera="kitti360_perspective", n=7),
    }
    for name, c in cases.items():
        v = c["v"]
        x = xyz[: c.get("n", len(xyz))]
        E = sc["extrinsic"][v]
        if c["camera"] == "scannet":
            # scannet extrinsics are world-to-camera: the function inverts them
            E = np.linalg.inv(E.astype(np.float64)).astype(np.float32)
        K = intrinsic(sc, v)
        img_xyz = sc["pos"][v].astype(np.float32)
        out = vis.camera_projection(
            T(x), T(img_xyz), img_intrinsic_pinhole=T(K), img_extrinsic=T(E),
            img_mask=None if c["mask"] is None else T(c["mask"]),
            img_size=(W, H), crop_top=c["crop_top"], crop_bottom=c["crop_bottom"],
            r_max=c["r_max"], r_min=c["r_min"], camera=c["camera"])
        idx, dist, xp, yp = [o.numpy() for o in out]
        assert dist.dtype == np.float32 and xp.dtype == np.float64
        save(f"proj_{name}", xyz=x, img_xyz=img_xyz, extrinsic=E, intrinsic=K,
             mask=np.zeros(0) if c["mask"] is None else c["mask"],
             img_size=np.array([W, H]), crop_top=c["crop_top"],
             crop_bottom=c["crop_bottom"], r_max=float(c["r_max"]),
             r_min=float(c["r_min"]), camera=c["camera"],
             indices=idx, dist=dist, x_proj=xp, y_proj=yp)


# ------------------------------------------------------------- splat + zbuf
def gen_visibility():
    sc = S.make_scene("c1", seed=1, n_points=20000)
    xyz, W, H = sc["xyz"].copy(), *sc["img_size"]
    # duplicated points: equal depth (first index must win the z-buffer) and
    # shared centre pixels (last index must win the exact re-stamp)
    xyz[1000:1200] = xyz[5000:5200]
    xyz[7000:7050] = xyz[5000:5050]
    K = intrinsic(sc, 0)
    for name, ct, cb, v in [("nocrop", 0, 0, 0), ("crop", 37, 21, 2)]:
        E = sc["extrinsic"][v]
        img_xyz = sc["pos"][v].astype(np.float32)
        idx1, dist, xp, yp = vis.camera_projection(
            T(xyz), T(img_xyz), img_intrinsic_pinhole=T(K), img_extrinsic=T(E),
            img_size=(W, H), crop_top=ct, crop_bottom=cb, r_max=20, r_min=0,
            camera="kitti360_perspective")
        kw = dict(img_size=(W, H), crop_top=ct, crop_bottom=cb, voxel=0.05,
                  k_swell=1.5, d_swell=1e6)
        splat = vis.pinhole_splat_cpu(
            xp.numpy(), yp.numpy(), dist.numpy(), K, **kw)
        out = {}
        for exact in (False, True):
            i2, xpix, ypix = vis.visibility_from_splatting(
                xp, yp, dist, T(xyz)[idx1], img_extrinsic=T(E),
                img_intrinsic_pinhole=T(K), exact=exact,
                camera="kitti360_perspective", **kw)
            tag = "exact" if exact else "splat"
            out[f"{tag}_idx"] = i2.numpy()
            out[f"{tag}_x"] = xpix.numpy()
            out[f"{tag}_y"] = ypix.numpy()
        save(f"vis_{name}", x_proj=xp.numpy(), y_proj=yp.numpy(), dist=dist.numpy(),
             intrinsic=K, img_size=np.array([W, H]), crop_top=ct, crop_bottom=cb,
             voxel=0.05, k_swell=1.5, d_swell=1e6, splat=splat, **out)


# ----------------------------------------- fisheye / equirectangular cameras
def gen_wide_cameras():
    """camera_projection + the camera's *_splat_cpu + visibility_from_splatting
    for 'kitti360_fisheye' and 's3dis_equirectangular'.  NB: the reference's
    file runs here without numba, i.e. with numpy >= 2 typing (float32 arrays
    stay float32 next to Python float literals) where numba computes in
    float64: these fixtures pin the formulas to about 1e-3 pixel, not to the
    bit."""
    sc = S.make_scene("c1", seed=7, n_points=15000)
    xyz = sc["xyz"]
    kw = dict(voxel=0.05, k_swell=1.5, d_swell=1e6)
    cams = {
        "fisheye": dict(
            camera="kitti360_fisheye", img_size=(700, 700), v=2, crop_top=0, crop_bottom=0,
            fisheye=np.array([2.2134, 0.016, 0.4325, 668.2, 667.9, 358.4, 352.9], np.float32)),
        "equirect": dict(
            camera="s3dis_equirectangular", img_size=(1024, 512), v=5, crop_top=30,
            crop_bottom=20, opk=np.array([0.05, -0.1, 1.2], np.float32)),
    }
    for name, c in cams.items():
        v, (W, H) = c["v"], c["img_size"]
        E = sc["extrinsic"][v]
        img_xyz = sc["pos"][v].astype(np.float32)
        f = c.get("fisheye")
        opk = c.get("opk")
        common = dict(img_size=(W, H), crop_top=c["crop_top"], crop_bottom=c["crop_bottom"],
                      camera=c["camera"])
        idx1, dist, xp, yp = vis.camera_projection(
            T(xyz), T(img_xyz), img_opk=None if opk is None else T(opk),
            img_intrinsic_fisheye=None if f is None else T(f), img_extrinsic=T(E),
            r_max=20, r_min=0.5, **common)
        xyz1 = T(xyz)[idx1]
        if name == "fisheye":
            splat = vis.fisheye_splat_cpu(
                xp.numpy(), yp.numpy(), xyz1.numpy(), E, f, img_size=(W, H),
                crop_top=c["crop_top"], crop_bottom=c["crop_bottom"],
                camera=c["camera"], **kw)
        else:
            splat = vis.equirectangular_splat_cpu(
                xp.numpy(), yp.numpy(), dist.numpy(), img_size=(W, H),
                crop_top=c["crop_top"], crop_bottom=c["crop_bottom"], **kw)
        out = {}
        for exact in (False, True):
            i2, xpix, ypix = vis.visibility_from_splatting(
                xp, yp, dist, xyz1, img_extrinsic=T(E),
                img_intrinsic_fisheye=None if f is None else T(f), exact=exact,
                **common, **kw)
            tag = "exact" if exact else "splat"
            out[f"{tag}_idx"] = i2.numpy()
            out[f"{tag}_x"] = xpix.numpy()
            out[f"{tag}_y"] = ypix.numpy()
        save(f"wide_{name}", xyz=xyz, img_xyz=img_xyz, extrinsic=E,
             fisheye=np.zeros(0) if f is None else f, opk=np.zeros(0) if opk is None else opk,
             img_size=np.array([W, H]), crop_top=c["crop_top"], crop_bottom=c["crop_bottom"],
             r_max=20.0, r_min=0.5, camera=c["camera"], voxel=0.05, k_swell=1.5, d_swell=1e6,
             indices=idx1.numpy(), dist=dist.numpy(), x_proj=xp.numpy(), y_proj=yp.numpy(),
             splat=splat, **out)


# --------------------------------------------------- visibility model + feats
def gen_model():
    sc = S.make_scene("c1", seed=2, n_points=15000)
    xyz, W, H = sc["xyz"], *sc["img_size"]
    nrm, lin, pla, sca = attrs(xyz)
    v = 4
    K, E = intrinsic(sc, v), sc["extrinsic"][v]
    img_xyz = sc["pos"][v].astype(np.float32)
    import contextlib
    import io
    for name, use in [("full", True), ("bare", False)]:
        model = vis.SplattingVisibility(
            voxel=0.05, k_swell=1.5, d_swell=1e6, exact=True, img_size=(W, H),
            crop_top=0, crop_bottom=0, r_max=20, r_min=0,
            camera="kitti360_perspective")
        with contextlib.redirect_stdout(io.StringIO()):
            out = model(
                T(xyz), T(img_xyz), linearity=T(lin) if use else None,
                planarity=T(pla) if use else None, scattering=T(sca) if use else None,
                normals=T(nrm) if use else None, img_intrinsic_pinhole=T(K),
                img_extrinsic=T(E))
        save(f"model_{name}", xyz=xyz, img_xyz=img_xyz, extrinsic=E, intrinsic=K,
             img_size=np.array([W, H]), use_attrs=use,
             idx=out["idx"].numpy(), x=out["x"].numpy(), y=out["y"].numpy(),
             depth=out["depth"].numpy(), features=out["features"].numpy())


# ------------------------------------------------------------- depth-map vis
def gen_depth_map():
    from PIL import Image
    sc = S.make_scene("c1", seed=3, n_points=15000)
    xyz, (W, H) = sc["xyz"], (704, 188)
    v = 1
    K = intrinsic(sc, v)
    K[0, 0] /= 2; K[1, 1] /= 2; K[0, 2] /= 2; K[1, 2] /= 2
    E = sc["extrinsic"][v]
    img_xyz = sc["pos"][v].astype(np.float32)
    idx1, dist, xp, yp = vis.camera_projection(
        T(xyz), T(img_xyz), img_intrinsic_pinhole=T(K), img_extrinsic=T(E),
        img_size=(W, H), r_max=20, r_min=0, camera="kitti360_perspective")
    # S3DIS-format 16-bit depth image [H, W]: per-pixel min distance / holes
    raw = np.full((H, W), 65535, np.uint16)
    dm = np.full((W, H), np.inf)
    xi, yi = xp.numpy().astype(int), yp.numpy().astype(int)
    np.minimum.at(dm, (xi, yi), dist.numpy())
    ok = np.isfinite(dm)
    raw.T[ok] = np.round(dm[ok] * 512).astype(np.uint16)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "view_depth.png")
        Image.fromarray(raw).save(path)
        decoded = vis.read_s3dis_depth_map(path, img_size=(W, H), empty=-1)
        i2, xo, yo = vis.visibility_from_depth_map(
            xp, yp, dist, depth_map_path=path, img_size=(W, H), depth_threshold=0.05)
    save("depth_map", x_proj=xp.numpy(), y_proj=yp.numpy(), dist=dist.numpy(),
         raw_png=raw, depth_map=decoded.numpy().astype(np.float32),
         img_size=np.array([W, H]), depth_threshold=0.05,
         indices=i2.numpy(), x=xo.numpy(), y=yo.numpy())


# ----------------------------------------------------------------- MapImages
def run_map_images(name, n_points, n_views, seed, exact, cylinder, ref_size,
                   proj_upscale=1, downscale=1, crop=None, use_attrs=True,
                   shuffle_ids=False, img_scale=1.0):
    import contextlib
    import io
    from sklearn.neighbors import KDTree
    tr, image = ref["transforms"], ref["image"]
    sc = S.make_scene("c1", seed=seed, n_points=n_points, n_views=n_views)
    xyz = sc["xyz"]
    nrm, lin, pla, sca = attrs(xyz)
    rng = np.random.default_rng(seed + 50)
    mapping_index = rng.permutation(n_points) if shuffle_ids else np.arange(n_points)
    data = ref["Data"](pos=T(xyz), mapping_index=T(mapping_index.astype(np.int64)))
    if use_attrs:
        data.linearity, data.planarity, data.scattering, data.norm = \
            T(lin), T(pla), T(sca), T(nrm)
    V = n_views
    s = np.float32(img_scale)
    images = image.SameSettingImageData(
        path=np.array([f"/nowhere/rgb/v{i}_rgb.png" for i in range(V)], dtype="O"),
        pos=T(sc["pos"]), extrinsic=T(sc["extrinsic"]),
        fx=T(sc["fx"] * s), fy=T(sc["fy"] * s), mx=T(sc["mx"] * s), my=T(sc["my"] * s),
        ref_size=ref_size, proj_upscale=proj_upscale, downscale=1)
    fx0 = images.fx.numpy().copy()
    if crop is not None:
        images.crop_size = crop[0]
        images.crop_offsets = T(np.asarray(crop[1], np.int64))
    images.downscale = downscale
    cams = dict(pos=images.pos.numpy().copy(), extrinsic=images.extrinsic.numpy().copy(),
                fx=images.fx.numpy().copy(), fy=images.fy.numpy().copy(),
                mx=images.mx.numpy().copy(), my=images.my.numpy().copy())
    t = tr.MapImages(
        method="SplattingVisibility", proj_upscale=None, ref_size=None,
        use_cuda=False, cylinder=cylinder, voxel=0.05, r_max=20, r_min=0,
        k_swell=1.5, d_swell=1e6, exact=exact, camera="kitti360_perspective")
    with contextlib.redirect_stdout(io.StringIO()):
        _, out = t._process(data, images)
    m = out.mappings
    # traversal order of the KD-tree the samplers built (cached then deleted by
    # the transform): rebuild it the same way, it is deterministic
    tree = KDTree(np.asarray(data.pos[:, :2] if cylinder else data.pos), leaf_size=50)
    order = np.asarray(tree.idx_array).astype(np.int64)
    save(name, xyz=xyz, mapping_index=mapping_index, order=order,
         cam_pos=cams["pos"], extrinsic=cams["extrinsic"], fx=cams["fx"],
         fy=cams["fy"], mx=cams["mx"], my=cams["my"], fx_before=fx0,
         ref_size=np.array(ref_size), proj_upscale=proj_upscale, downscale=downscale,
         crop_size=np.array(images.crop_size),
         crop_offsets=images.crop_offsets.numpy(), exact=exact, cylinder=cylinder,
         use_attrs=use_attrs, seed=seed,
         out_num_views=out.num_views, out_pos=out.pos.numpy(),
         pointers=m.pointers.numpy(), images=m.images.numpy(),
         view_pointers=m.values[1].pointers.numpy(), pixels=m.pixels.numpy(),
         features=m.features.numpy())


def gen_map_images():
    run_map_images("mapimages_exact", 30000, 4, 10, True, True, (1408, 376))
    run_map_images("mapimages_exact_sphere", 12000, 3, 11, True, False, (1408, 376),
                   use_attrs=False, shuffle_ids=True)
    # non-exact splats on small images, projection at 2x, mappings at 1/2, cropped
    run_map_images("mapimages_splat", 6000, 2, 12, False, True, (352, 94),
                   proj_upscale=2, downscale=2,
                   crop=((320, 80), [[16, 8], [24, 4]]), img_scale=0.25)


# ------------------------------------------------------------------- lex ops
def gen_lexops():
    u = ref["utils"]
    rng = np.random.default_rng(7)
    a = rng.integers(0, 40, 5000)
    b = rng.integers(0, 7, 5000)
    c = rng.integers(0, 90, 5000)
    save("lexops", a=a, b=b, c=c,
         argunique=u.lexargunique(T(a), T(b), T(c)).numpy(),
         argsort=u.lexargsort(T(a), T(b)).numpy(),
         unique_a=u.lexunique(T(a)).numpy(),
         argunique_empty=u.lexargunique(T(a[:0]), T(b[:0])).numpy())


# ---------------------------------------------------------------- from_dense
def gen_from_dense():
    image = ref["image"]
    rng = np.random.default_rng(8)
    P, n_pts, n_img = 2000, 300, 4
    pid = rng.integers(0, n_pts - 20, P)      # the last 20 points are unseen
    iid = rng.integers(0, n_img, P)
    pix = rng.integers(0, 300, (P, 2)).astype(np.int16)
    feat = rng.random((P, 3)).astype(np.float32)
    m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n_pts)
    save("from_dense", point_ids=pid, image_ids=iid, pixels=pix, features=feat,
         num_points=n_pts, pointers=m.pointers.numpy(), images=m.images.numpy(),
         view_pointers=m.values[1].pointers.numpy(), out_pixels=m.pixels.numpy(),
         out_features=m.features.numpy())
    # CSRBatch round trip on three mappings
    csr = ref["csr"]
    ms = []
    for k in range(3):
        pid = rng.integers(0, 50, 200)
        iid = rng.integers(0, 3, 200)
        pix = rng.integers(0, 100, (200, 2)).astype(np.int16)
        feat = rng.random((200, 2)).astype(np.float32)
        ms.append((pid, iid, pix, feat))
    mm = [image.ImageMapping.from_dense(T(a), T(b), T(c), T(d), num_points=60)
          for a, b, c, d in ms]
    batch = csr.CSRBatch.from_csr_list(mm)
    arrs = {}
    for k, (a, b, c, d) in enumerate(ms):
        arrs.update({f"pid{k}": a, f"iid{k}": b, f"pix{k}": c, f"feat{k}": d})
    save("csr_batch", num_points=60, pointers=batch.pointers.numpy(),
         images=batch.values[0].numpy(),
         view_pointers=batch.values[1].pointers.numpy(),
         pixels=batch.values[1].values[0].numpy(), features=batch.values[2].numpy(),
         **arrs)


# ----------------------------------------------------------- mapped features
def gen_mapped_features():
    """What SameSettingImageData.get_mapped_features (core/multimodal/image.py:
    1272-1297) is made of: ImageMapping.rescale_images / downscale_images /
    upscale_images (:1911-2037), feature_map_indexing (:1881-1895) on a
    [B, C, H, W] feature map, and sparse_interpolation (:105-170)."""
    image = ref["image"]
    rng = np.random.default_rng(31)
    P, n_pts, n_img = 4000, 500, 5
    Wm, Hm = 96, 64                                   # mapping size (W, H)
    pid = rng.integers(0, n_pts - 30, P)
    iid = rng.integers(0, n_img, P)
    pix = np.stack([rng.integers(0, Wm, P), rng.integers(0, Hm, P)], 1).astype(np.int16)
    feat = rng.random((P, 2)).astype(np.float32)
    fresh = lambda: image.ImageMapping.from_dense(
        T(pid), T(iid), T(pix), T(feat), num_points=n_pts)
    m = fresh()
    # the mapping's own arrays: the order of the pixels inside a view depends on
    # an unstable sort, so consumers start from these rather than from_dense
    arrs = dict(num_points=n_pts, mapping_size=np.array([Wm, Hm]),
                pointers=m.pointers.numpy(), images=m.images.numpy(),
                view_pointers=m.values[1].pointers.numpy(), pixels=m.pixels.numpy(),
                features=m.features.numpy())
    # feature maps at the mapping resolution and at half of it
    x1 = rng.random((n_img, 7, Hm, Wm)).astype(np.float32)
    x2 = rng.random((n_img, 7, Hm // 2, Wm // 2)).astype(np.float32)
    arrs.update(x1=x1, x2=x2)
    # interpolate=False: mappings rescaled to the feature map, then x[idx]
    arrs["gather_scale1"] = T(x1)[m.rescale_images(1).feature_map_indexing].numpy()
    down = m.rescale_images(1 / 2)
    arrs["down_pixels"] = down.pixels.numpy()
    arrs["down_view_pointers"] = down.values[1].pointers.numpy()
    arrs["gather_scale_half"] = T(x2)[down.feature_map_indexing].numpy()
    for center in (True, False):
        # on a fresh mapping: the reference's upscale_images writes the new
        # pixels into the nested CSR it shares with its source (shallow clone)
        up = fresh().upscale_images(2, center=center)
        arrs[f"up_pixels_center{int(center)}"] = up.pixels.numpy()
    # interpolate=True with scale != 1 (:1288-1293)
    resolution = torch.Tensor([[Wm, Hm]])
    coords = (m.pixels / (resolution - 1))[:, [1, 0]]
    batch = m.feature_map_indexing[0]
    arrs["coords"] = coords.numpy()
    arrs["batch"] = batch.numpy()
    for mode in ("border", "zeros", "reflection"):
        arrs["interp_" + mode] = image.sparse_interpolation(
            T(x2), coords, batch, padding_mode=mode).numpy()
    save("mapped_features", **arrs)


# ------------------------------------------------------------- image batches
def mapping_arrays(m, prefix):
    out = {prefix + "pointers": m.pointers.numpy(), prefix + "images": m.images.numpy(),
           prefix + "view_pointers": m.values[1].pointers.numpy(),
           prefix + "pixels": m.pixels.numpy()}
    if m.has_features:
        out[prefix + "features"] = m.features.numpy()
    return out


def gen_image_batch():
    """SameSettingImageBatch / ImageBatch collation and ImageMapping.select_images /
    select_views (core/multimodal/image.py:1300-1416, 1608-1735, 2039-2175)."""
    image = ref["image"]

    def item(n_pts, n_img, P, seed, size):
        r = np.random.default_rng(seed)
        pid = r.integers(0, n_pts - 5, P)
        iid = np.concatenate([np.arange(n_img), r.integers(0, n_img, P - n_img)])
        pix = r.integers(0, size, (P, 2)).astype(np.int16)
        feat = r.random((P, 2)).astype(np.float32)
        m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n_pts)
        f32 = lambda *shape: T(r.random(shape).astype(np.float32))
        return image.SameSettingImageData(
            path=np.array([f"s{seed}_{i}" for i in range(n_img)], dtype='O'),
            pos=T(r.random((n_img, 3))), extrinsic=f32(n_img, 4, 4), fx=f32(n_img),
            fy=f32(n_img), mx=f32(n_img), my=f32(n_img), ref_size=(size, size),
            proj_upscale=1, downscale=1, mappings=m, x=f32(n_img, 3, size, size))

    # three samples; each has images of two settings (64 px and 32 px)
    n_pts = [50, 70, 40]
    big = [item(n, k, P, 10 + i, 64) for i, (n, k, P) in enumerate(zip(n_pts, [3, 4, 2], [200, 300, 150]))]
    small = [item(n, k, P, 20 + i, 32) for i, (n, k, P) in enumerate(zip(n_pts, [2, 2, 3], [90, 120, 100]))]
    arrs = dict(n_pts=np.array(n_pts))
    for tag, items in (("big", big), ("small", small)):
        for i, im in enumerate(items):
            arrs.update(mapping_arrays(im.mappings, f"{tag}{i}_"))
            arrs[f"{tag}{i}_pos"] = im.pos.numpy()
            arrs[f"{tag}{i}_x"] = im.x.numpy()
    b = image.SameSettingImageBatch.from_data_list(big)
    arrs.update(mapping_arrays(b.mappings, "ssb_"))
    arrs.update(ssb_pos=b.pos.numpy(), ssb_x=b.x.numpy(), ssb_sizes=b.batch_items_sizes,
                ssb_path=np.array([str(p) for p in b.path]))
    arrs.update(mapping_arrays(b.to_data_list()[1].mappings, "ssb_item1_"))
    # select_images / select_views on the second sample's mapping
    m = big[1].mappings
    sel = np.array([2, 0], np.int64)
    arrs["sel_idx"] = sel
    arrs.update(mapping_arrays(m.select_images(T(sel)), "sel_"))
    mask = np.random.default_rng(5).random(m.num_items) < 0.5
    mask[m.images.numpy() == 1] = False          # image 1 loses all its views
    out, img_idx = m.select_views(T(mask))
    arrs["view_mask"] = mask
    arrs["views_img_idx"] = img_idx.numpy()
    arrs.update(mapping_arrays(out, "views_"))
    # ImageBatch: [big_i, small_i] per sample
    ib = image.ImageBatch.from_data_list([image.ImageData([a, c]) for a, c in zip(big, small)])
    for k, ssb in enumerate(ib):
        arrs[f"ib{k}_ref_size"] = np.array(ssb.ref_size)
        arrs[f"ib{k}_pos"] = ssb.pos.numpy()
        arrs.update(mapping_arrays(ssb.mappings, f"ib{k}_"))
    back = ib.to_data_list()
    for i, il in enumerate(back):
        for k, im in enumerate(il):
            src = (big, small)[k][i]
            assert torch.equal(im.mappings.pointers, src.mappings.pointers)
            assert torch.equal(im.mappings.pixels, src.mappings.pixels)
    save("image_batch", **arrs)


# ------------------------------------------------------------- select_points
def gen_select_points():
    """ImageMapping.select_points (core/multimodal/image.py:2177-2287), the CSR
    re-indexing after a 3D sampling ('pick') or a strided convolution ('merge')."""
    image = ref["image"]
    rng = np.random.default_rng(21)
    P, n_pts, n_img = 3000, 400, 5
    pid = rng.integers(0, n_pts - 30, P)
    iid = rng.integers(0, n_img, P)
    pix = rng.integers(0, 40, (P, 2)).astype(np.int16)      # small images: duplicates after merging
    feat = rng.random((P, 3)).astype(np.float32)
    m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n_pts)
    pick = rng.permutation(n_pts)[:150].astype(np.int64)
    out_pick = m.select_points(T(pick), mode='pick')
    n_vox = 120
    merge = np.concatenate([np.arange(n_vox), rng.integers(0, n_vox, n_pts - n_vox)]).astype(np.int64)
    merge = merge[rng.permutation(n_pts)]
    out_merge = m.select_points(T(merge), mode='merge')
    arrs = dict(point_ids=pid, image_ids=iid, pixels=pix, features=feat, num_points=n_pts,
                pick=pick, merge=merge)
    for tag, o in (("pick", out_pick), ("merge", out_merge)):
        arrs.update({tag + "_pointers": o.pointers.numpy(), tag + "_images": o.images.numpy(),
                     tag + "_view_pointers": o.values[1].pointers.numpy(),
                     tag + "_pixels": o.pixels.numpy(), tag + "_features": o.features.numpy()})
    save("select_points", **arrs)


# ------------------------------------------------------------------- pooling
def random_ptr(rng, nseg, mean, long_at=None):
    sizes = rng.poisson(mean, nseg)
    sizes[rng.random(nseg) < 0.15] = 0
    sizes[rng.random(nseg) < 0.1] = 1
    if long_at is not None:
        sizes[long_at] = 1100
    ptr = np.zeros(nseg + 1, np.int64)
    np.cumsum(sizes, out=ptr[1:])
    return ptr


def gen_pooling():
    pooling = ref["pooling"]
    rng = np.random.default_rng(9)
    arrs = {}
    for C in (1, 13, 64, 96):
        ptr = random_ptr(rng, 500, 4, long_at=17)
        x = (rng.random((ptr[-1], C)) * 2 - 1).astype(np.float32)
        arrs[f"ptr_{C}"] = ptr
        arrs[f"x_{C}"] = x
        for mode in ("max", "mean", "min", "sum"):
            pool = pooling.BimodalCSRPool(mode=mode)
            arrs[f"out_{C}_{mode}"] = pool(None, T(x), None, T(ptr)).numpy()
    # the docstring case of pooling.py:185-205
    ptr = np.array([0, 4, 4, 5, 10, 20], np.int64)
    x = rng.random((20, 7)).astype(np.float32)
    arrs["ptr_doc"], arrs["x_doc"] = ptr, x
    for mode in ("max", "mean", "min", "sum"):
        arrs[f"out_doc_{mode}"] = pooling.BimodalCSRPool(mode=mode)(
            None, T(x), None, T(ptr)).numpy()
    save("segment_csr", **arrs)

    arrs = {}
    src = torch.arange(15).float().view(-1, 1).repeat_interleave(2, dim=1)
    ptr = torch.LongTensor([0, 5, 10, 15])
    arrs["doc_src"], arrs["doc_ptr"] = src.numpy(), ptr.numpy()
    arrs["doc_out"] = pooling.segment_softmax_csr(src.clone(), ptr).numpy()
    arrs["doc_out_scaled"] = pooling.segment_softmax_csr(src.clone(), ptr, scaling=True).numpy()
    for G in (1, 3, 4):
        ptr = random_ptr(rng, 800, 4)
        x = (rng.normal(size=(ptr[-1], G)) * 3).astype(np.float32)
        arrs[f"ptr_{G}"], arrs[f"x_{G}"] = ptr, x
        arrs[f"out_{G}"] = pooling.segment_softmax_csr(T(x.copy()), T(ptr)).numpy()
        arrs[f"out_{G}_scaled"] = pooling.segment_softmax_csr(
            T(x.copy()), T(ptr), scaling=True).numpy()
        arrs[f"gather_{G}"] = pooling.gather_csr(
            T(rng.random((800, G)).astype(np.float32)), T(ptr)).numpy()
        arrs[f"seggather_{G}"] = pooling.segment_gather_csr(T(x), T(ptr), reduce="max").numpy()
        arrs[f"expand_{G}"] = pooling.expand_group_feat(T(x[:50]), G, 10).numpy()
    arrs["gather_src_seedcheck"] = np.zeros(0)
    save("segment_softmax", **arrs)


def randomise_bn(module, gen):
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.3)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.2)


def gen_attention_pools():
    import contextlib
    import io
    pooling = ref["pooling"]
    rng = np.random.default_rng(10)
    N, F_main, F_map, C = 500, 32, 8, 64
    ptr = random_ptr(rng, N, 4)
    Vw = int(ptr[-1])
    x_main = rng.random((N, F_main)).astype(np.float32)
    x_mod = rng.random((Vw, C)).astype(np.float32)
    x_map = rng.random((Vw, F_map)).astype(np.float32)
    variants = {
        "qkv_default": dict(num_groups=1, nc_qk=16, gating=True, dim_scaling=True,
                            group_scaling=False, use_num=True),
        "qkv_groups4": dict(num_groups=4, nc_qk=16, gating=True, dim_scaling=True,
                            group_scaling=True, use_num=True),
        "qkv_nogate": dict(num_groups=1, nc_qk=8, gating=False, dim_scaling=False,
                           group_scaling=False, use_num=False),
    }
    for name, kw in variants.items():
        gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)   # (hash() is salted per process)
        torch.manual_seed(sum(map(ord, name)))
        mod = pooling.QKVBimodalCSRPool(
            in_main=F_main, in_map=F_map, in_mod=C, out_mod=C, save_last=True,
            map_encoder="DeepSetFeat", **kw).eval()
        with torch.no_grad():
            randomise_bn(mod, gen)
            if mod.G is not None:
                mod.G.weight.copy_(torch.rand(1, kw["num_groups"], generator=gen) + 0.5)
                mod.G.bias.copy_(torch.randn(1, kw["num_groups"], generator=gen) * 0.1)
            with contextlib.redirect_stdout(io.StringIO()):
                out = mod(T(x_main), T(x_mod), T(x_map), T(ptr))
        sd = {"sd." + k: v.numpy() for k, v in mod.state_dict().items()}
        last = dict(C=mod._last_C.numpy(), A=mod._last_A.numpy(),
                    G=np.zeros(0) if mod.G is None else mod._last_G.numpy())
        # the same module evaluated in float64: how far the reference's own float32 result is
        # from the exact one (a nearly closed gate leaves small, cancelling outputs)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out64 = mod.double()(T(x_main).double(), T(x_mod).double(), T(x_map).double(), T(ptr))
        save(name, x_main=x_main, x_mod=x_mod, x_map=x_map, ptr=ptr,
             x_pool=out.numpy(), x_pool64=out64.numpy(), **last,
             **{k: np.asarray(v) for k, v in kw.items()}, **sd)

    # flagship-like Group pool (conf/models/.../sparseconv3d.yaml:7320-7328)
    kw = dict(num_groups=4, gating=True, group_scaling=True, use_num=True)
    torch.manual_seed(77)
    gen = torch.Generator().manual_seed(78)
    mod = pooling.GroupBimodalCSRPool(
        in_map=F_map, in_mod=C, out_mod=C, use_mod=False,
        map_encoder="DeepSetFeat", **kw).eval()
    with torch.no_grad():
        randomise_bn(mod, gen)
        mod.G.weight.copy_(torch.rand(1, 4, generator=gen) + 0.5)
        mod.G.bias.copy_(torch.randn(1, 4, generator=gen) * 0.1)
        with contextlib.redirect_stdout(io.StringIO()):
            out = mod(None, T(x_mod), T(x_map), T(ptr))
    sd = {"sd." + k: v.numpy() for k, v in mod.state_dict().items()}
    save("group_pool", x_mod=x_mod, x_map=x_map, ptr=ptr, x_pool=out.numpy(),
         **{k: np.asarray(v) for k, v in kw.items()}, **sd)


def gen_training():
    """One training step of the attentive pools in the reference (``.train()``:
    BatchNorm with batch statistics): output, gradients of sum(out * w) with
    respect to the inputs and to every parameter, running statistics after the
    step.  The torch_scatter stand-in of the loader is plain differentiable
    torch code."""
    import contextlib
    import io
    pooling = ref["pooling"]
    rng = np.random.default_rng(12)
    N, F_main, F_map, C = 300, 32, 8, 64
    ptr = random_ptr(rng, N, 4)
    Vw = int(ptr[-1])
    x_main = rng.random((N, F_main)).astype(np.float32)
    x_mod = rng.random((Vw, C)).astype(np.float32)
    x_map = rng.random((Vw, F_map)).astype(np.float32)
    w = rng.normal(size=(N, C)).astype(np.float32)
    cases = {
        "train_qkv": ("QKVBimodalCSRPool", dict(
            in_main=F_main, num_groups=1, nc_qk=16, gating=True, dim_scaling=True,
            group_scaling=False, use_num=True, map_encoder="DeepSetFeat")),
        "train_qkv_groups4_modk": ("QKVBimodalCSRPool", dict(
            in_main=F_main, num_groups=4, nc_qk=8, gating=True, dim_scaling=True,
            group_scaling=True, use_mod_k=True, map_encoder="MinMaxDiffSetFeat", use_num=True)),
        "train_group": ("GroupBimodalCSRPool", dict(
            num_groups=4, gating=True, group_scaling=True, use_num=True,
            map_encoder="DeepSetFeat")),
    }
    for name, (cls, kw) in cases.items():
        torch.manual_seed(sum(map(ord, name)))
        gen = torch.Generator().manual_seed(len(name))
        mod = getattr(pooling, cls)(in_map=F_map, in_mod=C, out_mod=C, **kw)
        with torch.no_grad():
            randomise_bn(mod, gen)
            mod.G.weight.copy_(torch.rand(1, kw["num_groups"], generator=gen) + 0.5)
            mod.G.bias.copy_(torch.randn(1, kw["num_groups"], generator=gen) * 0.1)
        sd0 = {"sd." + k: v.clone().numpy() for k, v in mod.state_dict().items()}
        mod.train()
        xs = [T(a.copy()).requires_grad_(True) for a in (x_main, x_mod, x_map)]
        with contextlib.redirect_stdout(io.StringIO()):
            out = mod(xs[0] if cls.startswith("QKV") else None, xs[1], xs[2], T(ptr))
        (out * T(w)).sum().backward()
        arrs = dict(x_main=x_main, x_mod=x_mod, x_map=x_map, ptr=ptr, w=w, x_pool=out.detach().numpy(),
                    g_x_mod=xs[1].grad.numpy(), g_x_map=xs[2].grad.numpy(), cls=cls)
        if cls.startswith("QKV"):
            arrs["g_x_main"] = xs[0].grad.numpy()
        for k, p_ in mod.named_parameters():
            arrs["grad." + k] = p_.grad.numpy()
        for k, v in mod.state_dict().items():
            if "running_" in k:
                arrs["after." + k] = v.numpy()
        save(name, **arrs, **{"kw." + k: np.asarray(v) for k, v in kw.items()}, **sd0)


# --------------------------------------------------- neighbourhood features
class DenseLazy:
    """Dense stand-in for the three pykeops LazyTensor operations the transform
    uses (image.py:511-517): broadcast difference, square, sum over the
    coordinate axis and argKmin (ascending distance, ties by index)."""

    def __init__(self, t):
        self.t = t

    def __sub__(self, other):
        return DenseLazy(self.t - other.t)

    def __pow__(self, p):
        return DenseLazy(self.t ** p)

    def sum(self, dim):
        return DenseLazy(self.t.sum(dim=dim))

    def argKmin(self, k, dim):
        return torch.argsort(self.t, dim=dim, stable=True)[:, :k]


def gen_neighborhood_features():
    """NeighborhoodBasedMappingFeatures (data_transform/multimodal/image.py:
    431-612) behind MapImages-like mappings, k = [5, 20], density + occlusion."""
    image, tr = ref["image"], ref["transforms"]
    tr.LazyTensor = DenseLazy
    sc = S.make_scene("c1", seed=9, n_points=3000, n_views=5)
    xyz = sc["xyz"]
    rng = np.random.default_rng(90)
    n, n_img = xyz.shape[0], 5
    sees = rng.random((n, n_img)) < 0.35
    sees[:100] = False
    pid, iid = np.nonzero(sees)
    pix = rng.integers(0, 200, (pid.shape[0], 2)).astype(np.int16)
    feat = rng.random((pid.shape[0], 2)).astype(np.float32)
    m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n)
    images = image.SameSettingImageData(
        path=np.array([f"v{i}" for i in range(n_img)], dtype='O'), pos=T(sc["pos"]),
        ref_size=(200, 200), proj_upscale=1, downscale=1, mappings=m)
    data = _refshim.Data(pos=T(xyz), mapping_index=torch.arange(n))
    feats_before = m.features.numpy().copy()
    t = tr.NeighborhoodBasedMappingFeatures(k=[5, 20], voxel=0.05, density=True, occlusion=True)
    _, out = t(data, images)
    save("neighborhood_features", xyz=xyz, pointers=m.pointers.numpy(), images=m.images.numpy(),
         view_pointers=m.values[1].pointers.numpy(), pixels=m.pixels.numpy(),
         features_before=feats_before, features_after=out.mappings.features.numpy(),
         k_list=np.array([5, 20]), voxel=0.05)


# ------------------------------------------------- image-selection transforms
def gen_image_selection():
    """SameSettingImageData.select_points / select_views and the transforms
    SelectMappingFromPointId, DropImagesOutsideDataBoundingBox,
    PickImagesFromMappingArea (core/multimodal/image.py:826-950,
    core/data_transform/multimodal/image.py:615-762) on one sample."""
    image, tr = ref["image"], ref["transforms"]
    rng = np.random.default_rng(71)
    n, n_img, size = 600, 6, 64
    sees = rng.random((n, n_img)) < np.array([0.5, 0.02, 0.3, 0.0, 0.4, 0.05])
    sees[3, 3] = True                          # image 3 is seen by a single point
    pid, iid = np.nonzero(sees)
    pix = rng.integers(0, size, (pid.shape[0], 2)).astype(np.int16)
    feat = rng.random((pid.shape[0], 2)).astype(np.float32)
    pos_img = (rng.random((n_img, 3)) * 4 - 1).astype(np.float64)
    xyz = rng.random((n, 3)).astype(np.float32) * 2

    def fresh():
        m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n)
        return image.SameSettingImageData(
            path=np.array([f"v{i}" for i in range(n_img)], dtype='O'), pos=T(pos_img),
            ref_size=(size, size), proj_upscale=1, downscale=1, mappings=m)

    base = fresh()
    arrs = dict(xyz=xyz, pos_img=pos_img, size=size)
    arrs.update(mapping_arrays(base.mappings, "in_"))

    def record(tag, im):
        arrs[tag + "_pos"] = im.pos.numpy()
        arrs[tag + "_path"] = np.array([str(p) for p in im.path])
        arrs.update(mapping_arrays(im.mappings, tag + "_"))

    pick = rng.permutation(n)[:150].astype(np.int64)
    pick[0] = 3
    arrs["pick"] = pick
    record("pick", fresh().select_points(T(pick), mode='pick'))
    n_vox = 200
    merge = np.concatenate([np.arange(n_vox), rng.integers(0, n_vox, n - n_vox)]).astype(np.int64)
    merge = merge[rng.permutation(n)]
    arrs["merge"] = merge
    record("merge", fresh().select_points(T(merge), mode='merge'))
    im = fresh()
    mask = rng.random(im.mappings.num_items) < 0.6
    mask[im.mappings.images.numpy() == 2] = False
    arrs["view_mask"] = mask
    record("views", im.select_views(T(mask)))
    # transforms
    sub = np.sort(rng.permutation(n)[:250]).astype(np.int64)
    data = _refshim.Data(pos=T(xyz[sub]), mapping_index=T(sub))
    d2, out = tr.SelectMappingFromPointId()(data, fresh())
    arrs["sub"] = sub
    arrs["smfp_index"] = d2.mapping_index.numpy()
    record("smfp", out)
    data = _refshim.Data(pos=T(xyz), mapping_index=torch.arange(n))
    _, out = tr.DropImagesOutsideDataBoundingBox(margin=0.5)(data, fresh())
    record("bbox", out)
    for tag, kw in (("area", dict(area_ratio=0.02, n_max=3)),
                    ("area_bbox", dict(area_ratio=0.3, use_bbox=True))):
        _, out = tr.PickImagesFromMappingArea(**kw)(data, fresh())
        record(tag, out)
    save("image_selection", **arrs)


def gen_cropping():
    """ImageMapping.crop, SameSettingImageData.update_cropping and the
    downscale bookkeeping of the x setter (core/multimodal/image.py:688-787,
    2289-2352)."""
    image = ref["image"]
    rng = np.random.default_rng(81)
    n, n_img, W, H = 400, 4, 64, 48
    sees = rng.random((n, n_img)) < 0.4
    pid, iid = np.nonzero(sees)
    pix = np.stack([rng.integers(0, W, pid.shape[0]), rng.integers(0, H, pid.shape[0])], 1).astype(np.int16)
    feat = rng.random((pid.shape[0], 2)).astype(np.float32)
    x = rng.random((n_img, 3, H, W)).astype(np.float32)
    m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n)
    arrs = dict(x=x, size=np.array([W, H]))
    arrs.update(mapping_arrays(m, "in_"))
    images = image.SameSettingImageData(
        path=np.array([f"v{i}" for i in range(n_img)], dtype='O'), pos=T(rng.random((n_img, 3))),
        ref_size=(W, H), proj_upscale=1, downscale=1, mappings=m, x=T(x))
    crop_size = (32, 24)
    offsets = np.stack([rng.integers(0, W - 32, n_img), rng.integers(0, H - 24, n_img)], 1).astype(np.int64)
    arrs.update(crop_size=np.array(crop_size), crop_offsets=offsets)
    images.update_cropping(crop_size, T(offsets))
    arrs.update(mapping_arrays(images.mappings, "crop_"))
    arrs.update(crop_x=images.x.numpy(), crop_state_size=np.array(images.crop_size),
                crop_state_offsets=images.crop_offsets.numpy(), crop_img_size=np.array(images.img_size))
    # a feature map at half resolution: the setter doubles the downscale
    images.x = T(images.x.numpy()[:, :, ::2, ::2].copy())
    arrs.update(half_downscale=float(images.downscale), half_img_size=np.array(images.img_size))
    # a crop box no mapping falls into: pixels shifted, nothing dropped
    far = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n)
    out = far.crop((4, 4), T(np.full((n_img, 2), 1000, np.int64)))
    arrs["far_pixels"] = out.pixels.numpy()
    arrs["far_pointers"] = out.pointers.numpy()
    save("cropping", **arrs)


def gen_crop_groups():
    """CropImageGroups (with ImageMapping.bounding_boxes underneath) and
    PickMappingsFromMappingFeatures
    (core/data_transform/multimodal/image.py:877-931, 1040-1141) on one sample
    whose images' mappings span boxes of very different sizes."""
    image, tr = ref["image"], ref["transforms"]
    rng = np.random.default_rng(91)
    n, n_img, W, H = 500, 7, 256, 192
    # per image: a box the mapped pixels are drawn from (every image is seen: the
    # reference's crop asserts one offset per image present in the mappings)
    boxes = [(10, 40, 20, 50), (0, 256, 0, 192), (100, 200, 60, 130), (200, 256, 150, 192),
             (30, 150, 5, 60), (250, 256, 0, 10), (0, 70, 0, 192)]
    pid, iid, pix = [], [], []
    for i, b in enumerate(boxes):
        if b is None:
            continue
        pts = np.sort(rng.permutation(n)[:rng.integers(20, 120)])
        pid.append(pts)
        iid.append(np.full(pts.shape[0], i))
        pix.append(np.stack([rng.integers(b[0], b[1], pts.shape[0]),
                             rng.integers(b[2], b[3], pts.shape[0])], 1))
    pid, iid = np.concatenate(pid).astype(np.int64), np.concatenate(iid).astype(np.int64)
    pix = np.concatenate(pix).astype(np.int16)
    feat = rng.random((pid.shape[0], 3)).astype(np.float32)
    # feature maps whose values name their own position (exact in float32), so that
    # the fixture only keeps the shape and the sum of every cropped map
    x = np.arange(n_img * 2 * H * W, dtype=np.float32).reshape(n_img, 2, H, W)
    pos_img = rng.random((n_img, 3))

    def fresh():
        m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n)
        return image.SameSettingImageData(
            path=np.array([f"v{i}" for i in range(n_img)], dtype='O'), pos=T(pos_img),
            ref_size=(W, H), proj_upscale=1, downscale=1, mappings=m, x=T(x))

    base = fresh()
    arrs = dict(pos_img=pos_img, size=np.array([W, H]), n_img=n_img)
    arrs.update(mapping_arrays(base.mappings, "in_"))
    for k, v in zip(("w_min", "w_max", "h_min", "h_max"), base.mappings.bounding_boxes):
        arrs["bbox_" + k] = v.numpy()
    data = _refshim.Data(pos=T(rng.random((n, 3)).astype(np.float32)))
    for tag, kw in (("pad0", dict(padding=0, min_size=32)), ("pad8", dict(padding=8, min_size=64))):
        _, out = tr.CropImageGroups(**kw)(data, fresh())
        arrs[tag + "_n"] = len(out)
        for j, im in enumerate(out):
            t = f"{tag}_{j}"
            arrs[t + "_path"] = np.array([str(p) for p in im.path])
            arrs[t + "_crop_size"] = np.array(im.crop_size)
            arrs[t + "_crop_offsets"] = im.crop_offsets.numpy()
            arrs[t + "_x_shape"] = np.array(im.x.shape)
            arrs[t + "_x_sum"] = im.x.double().sum(dim=(1, 2, 3)).numpy()
            arrs.update(mapping_arrays(im.mappings, t + "_"))
    _, out = tr.PickMappingsFromMappingFeatures(feat=[0, 2], lower=[0.2, None], upper=[None, 0.7])(data, fresh())
    arrs["pmf_path"] = np.array([str(p) for p in out.path])
    arrs.update(mapping_arrays(out.mappings, "pmf_"))
    save("crop_groups", **arrs)


def gen_roll_and_credit():
    """CenterRoll with SameSettingImageData.update_rollings underneath, and
    PickImagesFromMemoryCredit with and without the coverage weights
    (core/data_transform/multimodal/image.py:765-874, 962-1037;
    core/multimodal/image.py:578-627) on spherical images whose mapped columns
    sit in different places, some across the seam."""
    image, tr = ref["image"], ref["transforms"]
    rng = np.random.default_rng(101)
    n, n_img, W, H = 400, 6, 512, 256
    # per image: centre column and half width of the mapped band (columns wrap around)
    bands = [(20, 60), (256, 40), (500, 90), (130, 200), (384, 10), (0, 255)]
    pid, iid, pix = [], [], []
    for i, (c, hw) in enumerate(bands):
        pts = np.sort(rng.permutation(n)[:rng.integers(30, 150)])
        pid.append(pts)
        iid.append(np.full(pts.shape[0], i))
        cols = (c + rng.integers(-hw, hw + 1, pts.shape[0])) % W
        pix.append(np.stack([cols, rng.integers(0, H, pts.shape[0])], 1))
    pid, iid = np.concatenate(pid).astype(np.int64), np.concatenate(iid).astype(np.int64)
    pix = np.concatenate(pix).astype(np.int16)
    feat = rng.random((pid.shape[0], 2)).astype(np.float32)
    x = np.arange(n_img * 2 * H * W, dtype=np.float32).reshape(n_img, 2, H, W) % 1021
    pos_img = rng.random((n_img, 3))

    def fresh(with_x=True, views=None, size=(W, H)):
        sel = np.arange(n_img) if views is None else np.asarray(views)
        keep = np.isin(iid, sel)
        new_id = np.searchsorted(sel, iid[keep])
        px = pix[keep].copy()
        px[:, 0] %= size[0]
        px[:, 1] %= size[1]
        m = image.ImageMapping.from_dense(T(pid[keep]), T(new_id), T(px), T(feat[keep]), num_points=n)
        return image.SameSettingImageData(
            path=np.array([f"s{size[0]}_v{i}" for i in sel], dtype='O'), pos=T(pos_img[sel]),
            ref_size=size, proj_upscale=1, downscale=1, mappings=m,
            x=T(x[sel][:, :, :size[1], :size[0]].copy()) if with_x else None)

    arrs = dict(pos_img=pos_img, size=np.array([W, H]), n_img=n_img, x_mod=1021)
    arrs.update(mapping_arrays(fresh().mappings, "in_"))
    data = _refshim.Data(pos=T(rng.random((n, 3)).astype(np.float32)))
    for res in (16, 4, 256):
        _, out = tr.CenterRoll(angular_res=res)(data, fresh())
        t = f"roll{res}"
        arrs[t + "_rollings"] = out.rollings.numpy()
        arrs[t + "_x_head"] = out.x[:, 0, 0, :8].numpy()
        arrs[t + "_x_sum_cols"] = out.x.double().sum(dim=(1, 2)).numpy()[:, ::37]
        arrs.update(mapping_arrays(out.mappings, t + "_"))
    # memory credit: two settings (full and half size images), seeded draws
    for tag, kw, seed in (("credit", dict(credit=3 * W * H), 5),
                          ("credit_cov", dict(img_size=[W, H], n_img=2, k_coverage=2), 6),
                          ("credit_small", dict(credit=W * H // 4 + 5, k_coverage=0.5), 7)):
        ims = image.ImageData([fresh(False, [0, 2, 3, 5]), fresh(False, [1, 4], size=(W // 2, H // 2))])
        np.random.seed(seed)
        _, out = tr.PickImagesFromMemoryCredit(**kw)(data, ims)
        arrs[tag + "_n"] = len(out)
        for j, im in enumerate(out):
            arrs[f"{tag}_{j}_path"] = np.array([str(p) for p in im.path])
            arrs.update(mapping_arrays(im.mappings, f"{tag}_{j}_"))
    save("roll_and_credit", **arrs)


def gen_image_side():
    """The image-side transforms between MapImages and the model that touch the
    feature maps or the mappings: ToFloatImage, AddPixelHeightFeature,
    AddPixelWidthFeature, RandomHorizontalFlip
    (core/data_transform/multimodal/image.py:1163-1232) on small uint8 images."""
    image, tr = ref["image"], ref["transforms"]
    rng = np.random.default_rng(202)
    n, n_img, W, H = 300, 4, 40, 24
    pid = np.sort(rng.integers(0, n, 500)).astype(np.int64)
    iid = rng.integers(0, n_img, 500).astype(np.int64)
    pix = np.stack([rng.integers(0, W, 500), rng.integers(0, H, 500)], 1).astype(np.int16)
    keys = np.unique(np.stack([pid, iid, pix[:, 0], pix[:, 1]], 1), axis=0)
    pid, iid, pix = keys[:, 0], keys[:, 1], keys[:, 2:].astype(np.int16)
    feat = rng.random((pid.shape[0], 2)).astype(np.float32)
    x = rng.integers(0, 256, (n_img, 3, H, W)).astype(np.uint8)

    def fresh():
        m = image.ImageMapping.from_dense(T(pid), T(iid), T(pix.copy()), T(feat), num_points=n)
        return image.SameSettingImageData(
            path=np.array([f"v{i}" for i in range(n_img)], dtype='O'), pos=T(rng.random((n_img, 3))),
            ref_size=(W, H), proj_upscale=1, downscale=1, mappings=m, x=T(x.copy()))

    data = _refshim.Data(pos=T(rng.random((n, 3)).astype(np.float32)))
    arrs = dict(x=x, size=np.array([W, H]), n=n)
    arrs.update(mapping_arrays(fresh().mappings, "in_"))
    _, out = tr.ToFloatImage()(data, fresh())
    arrs["float_x"] = out.x.numpy()
    _, out = tr.AddPixelHeightFeature()(data, tr.ToFloatImage()(data, fresh())[1])
    arrs["height_x"] = out.x.numpy()
    _, out = tr.AddPixelWidthFeature()(data, tr.ToFloatImage()(data, fresh())[1])
    arrs["width_x"] = out.x.numpy()
    # the flip: always, never, and with the probability decided by torch's seeded draw
    for tag, p, seed in (("flip1", 1.0, 0), ("flip0", 0.0, 0), ("flip_s3", 0.5, 3), ("flip_s4", 0.5, 4),
                         ("flip_s5", 0.5, 5), ("flip_s6", 0.5, 6)):
        torch.manual_seed(seed)
        _, out = tr.RandomHorizontalFlip(p=p)(data, fresh())
        arrs[tag + "_seed"] = seed
        arrs[tag + "_p"] = p
        arrs[tag + "_x"] = out.x.numpy()
        arrs.update(mapping_arrays(out.mappings, tag + "_"))
    save("image_side", **arrs)


def gen_attention_variants():
    """Option combinations of the attentive pools beyond the defaults: modality
    features mixed into keys / queries / scores, the other map encoders, the
    other DeepSetFeat fusions and pools.  Small on purpose (state dicts are
    stored with the vectors)."""
    import contextlib
    import io
    pooling = ref["pooling"]
    rng = np.random.default_rng(21)
    N, F_main, F_map, C = 300, 12, 8, 24
    ptr = random_ptr(rng, N, 4)
    Vw = int(ptr[-1])
    x_main = rng.random((N, F_main)).astype(np.float32)
    x_mod = rng.random((Vw, C)).astype(np.float32)
    x_map = rng.random((Vw, F_map)).astype(np.float32)
    variants = {
        "qkv_usemod": ("QKVBimodalCSRPool", dict(
            in_main=F_main, in_map=F_map, in_mod=C, out_mod=C, num_groups=2, nc_qk=4,
            use_mod_q=True, use_mod_k=True, gating=True, dim_scaling=True, group_scaling=True)),
        "qkv_minmax": ("QKVBimodalCSRPool", dict(
            in_main=F_main, in_map=F_map, in_mod=C, out_mod=16, num_groups=1, nc_qk=8,
            map_encoder="MinMaxDiffSetFeat", use_min=True, use_max=True, use_num=True)),
        "qkv_residual": ("QKVBimodalCSRPool", dict(
            in_main=F_main, in_map=F_map, in_mod=C, num_groups=C, nc_qk=2,
            map_encoder="DeepSetFeat", pool="max_mean_min_sum", fusion="residual")),
        "qkv_both": ("QKVBimodalCSRPool", dict(
            in_main=F_main, in_map=F_map, in_mod=C, num_groups=3, nc_qk=4,
            map_encoder="DeepSetFeat", pool="mean", fusion="both", use_num=True)),
        "group_usemod": ("GroupBimodalCSRPool", dict(
            in_map=F_map, in_mod=C, out_mod=C, num_groups=4, use_mod=True, gating=True,
            group_scaling=False, map_encoder="MLPSetFeat")),
    }
    for name, (cls, kw) in variants.items():
        torch.manual_seed(sum(map(ord, name)))
        gen = torch.Generator().manual_seed(len(name))
        mod = getattr(pooling, cls)(save_last=True, **kw).eval()
        with torch.no_grad():
            randomise_bn(mod, gen)
            if mod.G is not None:
                g = mod.G.weight.shape[1]
                mod.G.weight.copy_(torch.rand(1, g, generator=gen) + 0.5)
                mod.G.bias.copy_(torch.randn(1, g, generator=gen) * 0.1)
            with contextlib.redirect_stdout(io.StringIO()):
                out = mod(T(x_main), T(x_mod), T(x_map), T(ptr))
        sd = {"sd." + k: v.numpy() for k, v in mod.state_dict().items()}
        save(name, x_main=x_main, x_mod=x_mod, x_map=x_map, ptr=ptr, x_pool=out.numpy(),
             C=mod._last_C.numpy(), A=mod._last_A.numpy(), cls=np.asarray(cls),
             kwargs=np.asarray(repr(kw)), **sd)


if __name__ == "__main__":
    which = sys.argv[1:] or [
        "projection", "visibility", "wide_cameras", "model", "depth_map", "map_images", "lexops",
        "from_dense", "mapped_features", "image_batch", "select_points", "pooling", "attention_pools", "attention_variants", "training", "neighborhood_features", "image_selection", "cropping", "crop_groups", "roll_and_credit", "image_side"]
    for w in which:
        globals()["gen_" + w]()
