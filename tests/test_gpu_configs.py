"""GPU parity on the two multi-sample / dense workload shapes that are not the
bench line: a batch of KITTI-360-shaped cylinders mapped and pooled as one
concatenated ImageMapping, and the dense 4 MP depth-map case.  The CPU oracle
is the checker on the same seeded inputs; view counts are reduced so that the
oracle finishes in seconds, the per-view sizes are the full ones."""
import numpy as np
import pytest
import torch

import mvf_oracle as orc
from mvfusion import synthetic as S
from test_gpu_attentive_pools import _close
from test_gpu_parity import VIS, Data, check_mapping, make_inputs, npy

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_calls():
    """Forward-only workloads: autograd off, so that eval-mode attentive pools
    run their fused inference kernels."""
    with torch.no_grad():
        yield

T = torch.from_numpy
DEV = torch.device("cuda", 0)


def render_depth_map(xyz, sc, v):
    """S3DIS-style depth map of view v rendered from the cloud itself: nearest
    splat per pixel, quantised to 1/512 m as the 16-bit PNGs are, -1 = empty
    (visibility.py:1328-1358)."""
    W, H = sc["img_size"]
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = sc["fx"][v], sc["fy"][v], sc["mx"][v], sc["my"][v]
    kw = {k: VIS[k] for k in ("voxel", "k_swell", "d_swell", "camera")}
    idx, dist, x, y = orc.camera_projection(
        xyz, sc["pos"][v].astype(np.float32), img_intrinsic_pinhole=K,
        img_extrinsic=sc["extrinsic"][v], img_size=(W, H), r_max=VIS["r_max"], r_min=VIS["r_min"],
        camera=VIS["camera"])
    vi, px, py = orc.visibility_from_splatting(
        x, y, dist, img_intrinsic_pinhole=K, img_size=(W, H), exact=False, **kw)
    dm = np.full((W, H), -1.0, np.float32)
    q = np.round(dist[vi].astype(np.float64) * 512.0)
    dm[px, py] = (q / 512.0).astype(np.float32)
    return dm


def test_c5_depth_map_visibility_full_size():
    """2 M points against 4 MP depth maps: projection + depth-map visibility
    (visibility.py:592-623, 1361-1388) and the MapImages pass built on them."""
    from mvfusion.transforms import MapImages
    from mvfusion.visibility import camera_projection, visibility_from_depth_map
    V = 2
    sc = S.make_scene("c5", seed=4, n_views=V)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    assert N == 2_000_000 and tuple(sc["img_size"]) == (2304, 1728)
    W, H = sc["img_size"]
    depth_maps = [render_depth_map(xyz, sc, v) for v in range(V)]

    # the two visibility steps on one view
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = sc["fx"][0], sc["fy"][0], sc["mx"][0], sc["my"][0]
    ref = orc.camera_projection(
        xyz, sc["pos"][0].astype(np.float32), img_intrinsic_pinhole=K,
        img_extrinsic=sc["extrinsic"][0], img_size=(W, H), r_max=VIS["r_max"], r_min=VIS["r_min"],
        camera=VIS["camera"])
    got = camera_projection(
        T(xyz).to(DEV), T(sc["pos"][0].astype(np.float32)).to(DEV), img_intrinsic_pinhole=T(K).to(DEV),
        img_extrinsic=T(sc["extrinsic"][0]).to(DEV), img_size=(W, H), r_max=VIS["r_max"],
        r_min=VIS["r_min"], camera=VIS["camera"])
    for g_, r_ in zip(got, ref):
        assert np.array_equal(npy(g_), r_)
    ridx, rx, ry = orc.visibility_from_depth_map(ref[2], ref[3], ref[1], depth_map=depth_maps[0],
                                                 img_size=(W, H), depth_threshold=0.05)
    gidx, gx, gy = visibility_from_depth_map(got[2], got[3], got[1], depth_map=T(depth_maps[0]).to(DEV),
                                             img_size=(W, H), depth_threshold=0.05)
    assert ridx.shape[0] > 10000
    assert np.array_equal(npy(gidx), ridx)
    assert np.array_equal(npy(gx), rx) and np.array_equal(npy(gy), ry)

    # MapImages with DepthBasedVisibility on both views
    vis = {k: VIS[k] for k in ("r_max", "r_min", "camera")}
    want = orc.map_images(xyz, np.arange(N), sc, order=None, method="depth_map", ref_size=(W, H),
                          cylinder=True, depth_maps=depth_maps, depth_threshold=0.05, **vis)
    data, images = make_inputs(xyz, np.arange(N), sc, (W, H), attrs=False)
    images.depth_maps = [T(d).to(DEV) for d in depth_maps]
    t = MapImages(method="DepthBasedVisibility", cylinder=True, point_order="index",
                  depth_threshold=0.05, **vis)
    _, out = t(data, images)
    check_mapping(out.mappings, want)


def test_c5_fused_depth_map_16_views_with_pools():
    """The per-GPU share of the dense scene: 2 M points, 16 views of 4 MP,
    depth-map visibility for all views in one launch set (no per-view loop),
    then the pixel -> view -> point pools; mapping bit-exact against the
    oracle's per-view loop, pools against its segment reductions."""
    from mvfusion.pooling import segment_csr_two_level
    from mvfusion.transforms import MapImages
    from mvfusion.visibility import render_depth_maps
    V, C_feat = 16, 8
    sc = S.make_scene("c5", seed=4, n_views=V)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    W, H = sc["img_size"]
    assert N == 2_000_000 and (W, H) == (2304, 1728)
    K = np.tile(np.eye(4, dtype=np.float32), (V, 1, 1))
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = sc["fx"], sc["fy"], sc["mx"], sc["my"]
    kw = {k: VIS[k] for k in ("voxel", "k_swell", "d_swell", "camera", "r_max", "r_min")}
    dm = render_depth_maps(T(xyz).to(DEV), T(sc["pos"]), T(K), T(sc["extrinsic"]), (W, H), **kw)
    dm_host = dm.cpu().numpy()
    vis = {k: VIS[k] for k in ("r_max", "r_min", "camera")}
    want = orc.map_images(xyz, np.arange(N), sc, order=None, method="depth_map", ref_size=(W, H),
                          cylinder=True, depth_maps=dm_host, depth_threshold=0.05, **vis)
    data, images = make_inputs(xyz, np.arange(N), sc, (W, H), attrs=False)
    images.depth_maps = dm
    t = MapImages(method="DepthBasedVisibility", cylinder=True, depth_threshold=0.05, **vis)
    model = t._settle(data, images)
    assert t._fusable(images, model)
    _, out = t(data, images)
    m = out.mappings
    assert m.images.shape[0] > 500_000
    check_mapping(m, want)
    # pools: one pixel per (point, view); pixel -> view max, view -> point max and mean
    rng = np.random.default_rng(5)
    Pn = m.images.shape[0]
    x_pix = rng.random((Pn, C_feat), dtype=np.float32)
    atomic_ptr = torch.arange(Pn + 1, dtype=torch.int64, device=DEV)
    g_max, g_mean = segment_csr_two_level(T(x_pix).to(DEV), atomic_ptr, m.pointers.to(DEV),
                                          inner="max", outer=("max", "mean"))
    xv = orc.segment_csr(x_pix, want["view_pointers"], "max")
    assert np.array_equal(npy(g_max), orc.segment_csr(xv, want["pointers"], "max"))
    np.testing.assert_allclose(npy(g_mean), orc.segment_csr(xv, want["pointers"], "mean"), rtol=1e-5, atol=0)


def test_c4_batch_of_cylinders():
    """Per-GPU share of the 64-cylinder batch: every cylinder mapped on its own,
    the mappings concatenated (csr.py:347-416, image.py:1627-1682), pixel
    features gathered from the stacked feature maps and pooled to views and
    points in one launch set; against the oracle per cylinder."""
    from mvfusion.mapping import ImageMappingBatch
    from mvfusion.pooling import BimodalCSRPool, gather_atomic_pool
    from mvfusion.transforms import MapImages, kdtree_order
    n_samples, C_feat = 3, 5
    rng = np.random.default_rng(11)
    maps, feats, refs = [], [], []
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, **VIS)
    for k in range(n_samples):
        sc = S.make_scene("c1", seed=20 + k)
        xyz = sc["xyz"]
        N = xyz.shape[0]
        W, H = sc["img_size"]
        order = kdtree_order(T(xyz), True).numpy()
        data, images = make_inputs(xyz, np.arange(N), sc, (W, H), attrs=False, order=order)
        _, out = t(data, images)
        ref = orc.map_images(xyz, np.arange(N), sc, order=order, ref_size=(W, H), cylinder=True,
                             exact=True, **VIS)
        check_mapping(out.mappings, ref)
        x = rng.random((out.num_views, C_feat, H, W)).astype(np.float32)
        maps.append(out.mappings.to(DEV))
        feats.append(x)
        # oracle pooling of this cylinder: gather, pixel -> view max, view -> point max
        pix = ref["pixels"].astype(np.int64)
        img_of_pixel = np.repeat(ref["images"], np.diff(ref["view_pointers"]))
        dense = x[img_of_pixel, :, pix[:, 1], pix[:, 0]]
        xv = orc.segment_csr(dense, ref["view_pointers"], "max")
        refs.append(orc.segment_csr(xv, ref["pointers"], "max"))
    batch = ImageMappingBatch.from_csr_list(maps)
    x_all = T(np.concatenate(feats, 0)).to(DEV)
    assert int(batch.images.max()) == x_all.shape[0] - 1
    x_view = gather_atomic_pool(x_all, batch, "max")
    x_point = BimodalCSRPool(mode="max")(None, x_view, None, batch.view_csr_indexing)
    assert np.array_equal(npy(x_point), np.concatenate(refs, 0))


@pytest.mark.parametrize("n_views", [33, 70])
def test_fused_map_images_many_views(n_views):
    """More views than one 32-bit word of the per-point view masks / one 64-byte
    row of seen flags: fused exact MapImages against the oracle."""
    from mvfusion.transforms import MapImages, kdtree_order
    sc = S.make_scene("c1", seed=40 + n_views, n_points=20000, n_views=n_views)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    order = kdtree_order(T(xyz), True).numpy()
    data, images = make_inputs(xyz, np.arange(N), sc, sc["img_size"], attrs=True, order=order)
    nrm, lin, pla, sca = S.surface_normals(xyz)
    ref = orc.map_images(xyz, np.arange(N), sc, order=order, ref_size=sc["img_size"], cylinder=True,
                         exact=True, linearity=lin, planarity=pla, scattering=sca, normals=nrm, **VIS)
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, **VIS)
    _, out = t(data, images)
    check_mapping(out.mappings, ref)


@pytest.mark.parametrize("case", ["scannet", "mask_crop_rmin", "upscale_downscale_crop", "sphere"])
def test_fused_map_images_options(case):
    """Option combinations of the fused exact path against the oracle on a small
    scene: world-to-camera extrinsics ('scannet'), image mask + crop_top /
    crop_bottom + r_min, proj_upscale / downscale / per-view crop offsets, and
    spherical instead of cylindrical sampling."""
    from mvfusion.images import SameSettingImageData
    from mvfusion.transforms import MapImages, kdtree_order
    sc = S.make_scene("c1", seed=77, n_points=30000, n_views=5)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    W, H = sc["img_size"]
    vis = dict(VIS)
    kw_img, kw_orc, cylinder, mask = {}, {}, True, None
    extr = sc["extrinsic"]
    if case == "scannet":
        vis["camera"] = "scannet"
        extr = np.stack([np.linalg.inv(e.astype(np.float64)).astype(np.float32) for e in extr])
    elif case == "mask_crop_rmin":
        rng = np.random.default_rng(1)
        mask = rng.random((W, H)) > 0.3
        vis.update(r_min=2.0, crop_top=40, crop_bottom=25)
    elif case == "upscale_downscale_crop":
        ref_size = (W // 2, H // 2)
        offs = np.array([[10 * k, 3 * k] for k in range(5)], np.int64)
        kw_img = dict(proj_upscale=2, downscale=2, crop_size=(ref_size[0] - 64, ref_size[1] - 16),
                      crop_offsets=T(offs))
        kw_orc = dict(proj_upscale=2, downscale=2, crop_size=(ref_size[0] - 64, ref_size[1] - 16),
                      crop_offsets=offs)
    elif case == "sphere":
        cylinder = False
        vis["r_max"] = 5
    ref_size = (W // 2, H // 2) if case == "upscale_downscale_crop" else (W, H)
    cams = dict(sc, extrinsic=extr)
    order = kdtree_order(T(xyz), cylinder).numpy()
    ref = orc.map_images(xyz, np.arange(N), cams, order=order, ref_size=ref_size, cylinder=cylinder,
                         img_mask=mask, exact=True, **kw_orc, **vis)
    data = Data()
    data.pos = T(xyz)
    data.mapping_index = torch.arange(N)
    data.mvf_point_order = T(order)
    images = SameSettingImageData(
        pos=T(sc["pos"]), extrinsic=T(extr), fx=T(sc["fx"]), fy=T(sc["fy"]), mx=T(sc["mx"]),
        my=T(sc["my"]), ref_size=ref_size, mask=None if mask is None else T(mask),
        **(kw_img or dict(proj_upscale=1)))
    t = MapImages(method="SplattingVisibility", cylinder=cylinder, exact=True, **vis)
    _, out = t(data, images)
    assert out.mappings.pointers.shape[0] == N + 1 and ref["images"].shape[0] > 500
    check_mapping(out.mappings, ref)


def test_map_images_repeated_mapping_ids():
    """Several points share a mapping id (voxelised clouds): several pixels per
    (id, view) group, the path that cannot use the fused kernels."""
    from mvfusion.transforms import MapImages, kdtree_order
    sc = S.make_scene("c1", seed=78, n_points=20000, n_views=3)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    ids = np.arange(N) // 3
    order = kdtree_order(T(xyz), True).numpy()
    ref = orc.map_images(xyz, ids, sc, order=order, ref_size=sc["img_size"], cylinder=True,
                         exact=True, **VIS)
    data, images = make_inputs(xyz, ids, sc, sc["img_size"], attrs=False, order=order)
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, **VIS)
    _, out = t(data, images)
    assert np.diff(ref["view_pointers"]).max() > 1
    check_mapping(out.mappings, ref)


def test_c2_non_exact_splat_visibility_full_size():
    """The WildScenes-shaped scan with exact=False (plain splat visibility: every
    pixel a point wins is a mapping), all 16 views in one launch set, against the
    oracle's per-view loop: 32 M pixels in 319 k views; then the pixel -> view ->
    point pools over them."""
    from mvfusion.pooling import segment_csr_two_level
    from mvfusion.transforms import MapImages
    sc = S.make_scene("c2", seed=2)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    assert N == 300_000 and sc["pos"].shape[0] == 16
    ref = orc.map_images(xyz, np.arange(N), sc, order=None, ref_size=sc["img_size"], cylinder=True,
                         exact=False, **VIS)
    data, images = make_inputs(xyz, np.arange(N), sc, sc["img_size"], attrs=False)
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=False, point_order="index", **VIS)
    assert t._fusable(images, t._settle(data, images))
    _, out = t(data, images)
    assert ref["pixels"].shape[0] > 30_000_000
    m = out.mappings
    check_mapping(m, ref)
    # the fused pass emits the pixels of a view in ascending (x, y) order already
    px = npy(m.pixels).astype(np.int64)
    key = px[:, 0] * 65536 + px[:, 1]
    seg = np.repeat(np.arange(ref["view_pointers"].shape[0] - 1), np.diff(ref["view_pointers"]))
    assert np.all((np.diff(key) > 0) | (np.diff(seg) > 0))
    rng = np.random.default_rng(6)
    x_pix = rng.random((px.shape[0], 3), dtype=np.float32)
    g_max, g_mean = segment_csr_two_level(T(x_pix).to(DEV), m.atomic_csr_indexing.to(DEV),
                                          m.view_csr_indexing.to(DEV), inner="max", outer=("max", "mean"))
    xv = orc.segment_csr(x_pix, ref["view_pointers"], "max")
    assert np.array_equal(npy(g_max), orc.segment_csr(xv, ref["pointers"], "max"))
    np.testing.assert_allclose(npy(g_mean), orc.segment_csr(xv, ref["pointers"], "mean"), rtol=1e-5, atol=0)


def test_non_exact_fused_options_and_overflow():
    """Plain splat visibility on the fused path with an image mask, crops, crop
    offsets and a crop window smaller than the image (pairs whose pixels are all
    cropped away vanish, their view still counts as seen), mapping features from
    point attributes; then the same pass from deliberately small buffers."""
    from mvfusion import visibility as visibility_module
    from mvfusion.images import SameSettingImageData
    from mvfusion.transforms import MapImages, kdtree_order
    sc = S.make_scene("c1", seed=79, n_points=30000, n_views=5)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    W, H = sc["img_size"]
    rng = np.random.default_rng(2)
    mask = rng.random((W, H)) > 0.2
    vis = dict(VIS, exact=False, r_min=1.0, crop_top=30, crop_bottom=20)
    offs = np.array([[40 * k, 5 * k] for k in range(5)], np.int64)
    crop_size = (W - 400, H - 60)
    order = kdtree_order(T(xyz), True).numpy()
    nrm, lin, pla, sca = S.surface_normals(xyz)
    ref = orc.map_images(xyz, np.arange(N), sc, order=order, ref_size=(W, H), cylinder=True,
                         img_mask=mask, crop_size=crop_size, crop_offsets=offs, linearity=lin,
                         planarity=pla, scattering=sca, normals=nrm, **vis)
    data, _ = make_inputs(xyz, np.arange(N), sc, (W, H), attrs=True, order=order)
    images = SameSettingImageData(
        pos=T(sc["pos"]), extrinsic=T(sc["extrinsic"]), fx=T(sc["fx"]), fy=T(sc["fy"]), mx=T(sc["mx"]),
        my=T(sc["my"]), ref_size=(W, H), proj_upscale=1, mask=T(mask), crop_size=crop_size,
        crop_offsets=T(offs))
    t = MapImages(method="SplattingVisibility", cylinder=True, **vis)
    model = t._settle(data, images)
    assert t._fusable(images, model)
    _, out = t(data, images)
    assert np.diff(ref["view_pointers"]).max() > 4 and ref["images"].shape[0] > 500
    check_mapping(out.mappings, ref)
    plan = t.prepare_fused(data, images, model, DEV)
    plan.out_cap, plan.pair_cap, plan.pix_cap, plan.mask_cap = 16, 16, 16, 8
    res = plan.run()
    assert plan.pix_cap >= ref["pixels"].shape[0] and plan.out_cap >= ref["images"].shape[0]
    assert plan.mask_cap > ref["pixels"].shape[0] // 16
    from mvfusion.mapping import ImageMapping
    check_mapping(ImageMapping.from_csr_arrays(res[0], res[1], res[2], res[3], res[5]), ref)


def test_view_sharded_scene_equals_single_pass():
    """The multi-GPU layout of one big scene: two blocks of views mapped and
    pooled separately against the whole cloud (what two ranks do), partial
    per-point pools merged as the all-reduces merge them; against one pass over
    all views.  Max pool bit-exact, mean pool to 1e-5."""
    from mvfusion import distributed as D
    from mvfusion.pooling import gather_atomic_pool, segment_csr
    from mvfusion.transforms import MapImages
    sc = S.make_scene("c1", seed=31)
    xyz = sc["xyz"]
    N, (W, H), V, C_feat = xyz.shape[0], sc["img_size"], sc["pos"].shape[0], 6
    rng = np.random.default_rng(32)
    x_all = T(rng.normal(size=(V, C_feat, H, W)).astype(np.float32) - 0.5).to(DEV)
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True,
                  point_order="index", **VIS)

    def run(view_ids):
        """map + pixel -> view pool of a block of views: (x_view, view_ptr)"""
        data, images = make_inputs(xyz, np.arange(N), sc, (W, H), attrs=False)
        sub = images[torch.as_tensor(list(view_ids))]
        n_sub = sub.num_views
        _, out = t(data, sub)
        # views that saw nothing were dropped: recover their ids in the block
        kept = [k for k in range(n_sub) if any(torch.equal(sub.pos[k], p) for p in out.pos)]
        x = x_all[torch.as_tensor(list(view_ids))[kept].to(DEV)]
        m = out.mappings.to(DEV)
        assert m.pointers.shape[0] == N + 1
        return gather_atomic_pool(x, m, "max"), m.view_csr_indexing

    xv, ptr = run(range(V))
    want_max = segment_csr(xv, ptr, reduce="max")
    want_mean = segment_csr(xv, ptr, reduce="mean")
    parts = [D.local_view_pools(*run(D.partition_views(V, r, 2))) for r in range(2)]
    mx = torch.maximum(parts[0][0], parts[1][0])
    sm, cnt = parts[0][1] + parts[1][1], parts[0][2] + parts[1][2]
    got_max, got_mean = D.finish_view_pools(mx, sm, cnt)
    assert torch.equal(cnt.long(), ptr[1:] - ptr[:-1])
    assert torch.equal(got_max, want_max)
    assert (want_max < 0).any()
    torch.testing.assert_close(got_mean, want_mean, rtol=1e-5, atol=1e-6)
    # single process: the exchange step is the identity
    one_max, one_mean = D.view_sharded_pool(xv, ptr)
    assert torch.equal(one_max, want_max)
    torch.testing.assert_close(one_mean, want_mean, rtol=1e-5, atol=1e-6)


def test_map_images_list_of_samples_multi_stream():
    """MapImages on a list of cylinders (the per-GPU batch of C4) spreads the
    fused passes over three streams; results equal the one-by-one calls."""
    from mvfusion.transforms import MapImages, kdtree_order
    scenes = [S.make_scene("c1", seed=40 + k, n_points=60000) for k in range(5)]

    def inputs():
        pairs = [make_inputs(sc["xyz"], np.arange(sc["xyz"].shape[0]), sc, sc["img_size"],
                             attrs=(k % 2 == 0)) for k, sc in enumerate(scenes)]
        return [p[0] for p in pairs], [p[1] for p in pairs]

    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, **VIS)
    data, images = inputs()
    _, out_list = t(data, images)
    one = MapImages(method="SplattingVisibility", cylinder=True, exact=True, lanes=1, **VIS)
    data, images = inputs()
    assert len(out_list) == len(scenes)
    for k, (da, im) in enumerate(zip(data, images)):
        _, want = one(da, im)
        got = out_list[k]
        assert got.num_views == want.num_views
        for a, b in ((got.mappings.pointers, want.mappings.pointers),
                     (got.mappings.images, want.mappings.images),
                     (got.mappings.pixels, want.mappings.pixels),
                     (got.mappings.features, want.mappings.features)):
            assert torch.equal(a, b), k
    # and against the oracle for one of them
    sc = scenes[1]
    ref = orc.map_images(sc["xyz"], np.arange(sc["xyz"].shape[0]), sc,
                         order=kdtree_order(T(sc["xyz"]), True).numpy(), ref_size=sc["img_size"], cylinder=True, exact=True, **VIS)
    check_mapping(out_list[1].mappings, ref, features_atol=1e-6)


def test_map_images_process_batch_equals_collated_samples():
    """One batched pass over a list of cylinders (mvf_map_images_exact_batch:
    clouds and views concatenated, every point facing the views of its own
    sample) returns the batch that mapping the samples one by one and collating
    them gives: samples of different sizes and view counts (up to 35, so that the
    seen flags of a point span three 16-view chunks and its kept-view mask two
    words), one with a camera that sees
    nothing (dropped from the batch) and mapping ids that leave gaps."""
    from mvfusion.images import SameSettingImageBatch
    from mvfusion.mapping import ImageMappingBatch
    from mvfusion.transforms import MapImages
    specs = [(30000, 8), (777, 3), (41000, 35), (20000, 5)]
    scenes = [S.make_scene("c1", seed=60 + k, n_points=n, n_views=v) for k, (n, v) in enumerate(specs)]
    # a camera far away from the second sample: one unseen view
    scenes[1]["pos"] = scenes[1]["pos"].copy()
    scenes[1]["pos"][1] += 500.0
    scenes[1]["extrinsic"] = scenes[1]["extrinsic"].copy()
    scenes[1]["extrinsic"][1, :3, 3] += 500.0

    W, H = scenes[0]["img_size"]
    mask = torch.ones((W, H), dtype=torch.bool)
    mask[:, H - 40:] = False                       # the images share one mask (the car's bonnet, say)

    def inputs():
        out = []
        for k, sc in enumerate(scenes):
            n = sc["xyz"].shape[0]
            ids = np.arange(n) if k != 3 else np.random.default_rng(5).permutation(n + 300)[:n]
            out.append(make_inputs(sc["xyz"], ids, sc, sc["img_size"], attrs=True))
            out[-1][1].mask = mask
        return [o[0] for o in out], [o[1] for o in out]

    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, lanes=1, **VIS)
    data, images = inputs()
    want = SameSettingImageBatch.from_data_list([t(d, im)[1] for d, im in zip(data, images)])
    data, images = inputs()
    _, got = t.process_batch(data, images)
    assert t.prepare_batch(*inputs()) is not None          # the batched pass, not the fallback
    assert isinstance(got, SameSettingImageBatch) and isinstance(got.mappings, ImageMappingBatch)
    assert got.num_views == want.num_views == sum(v for _, v in specs) - 1
    assert np.array_equal(got.batch_items_sizes, want.batch_items_sizes)
    assert torch.equal(got.pos, want.pos) and torch.equal(got.extrinsic, want.extrinsic)
    gm, wm = got.mappings, want.mappings
    assert torch.equal(gm.pointers, wm.pointers)
    assert torch.equal(gm.images, wm.images)
    assert torch.equal(gm.values[1].pointers, wm.values[1].pointers)
    assert torch.equal(gm.pixels, wm.pixels)
    assert torch.equal(gm.features, wm.features)
    assert torch.equal(gm.batch_items_sizes, wm.batch_items_sizes)
    assert torch.equal(gm.values[1].batch_items_sizes, wm.values[1].batch_items_sizes)
    # the batch splits back into the samples
    back = got.mappings.to_csr_list()
    data, images = inputs()
    for k, m in enumerate(back):
        _, one = t(data[k], images[k])
        assert torch.equal(m.pointers, one.mappings.pointers), k
        assert torch.equal(m.images, one.mappings.images), k
        assert torch.equal(m.pixels, one.mappings.pixels), k
        assert torch.equal(m.features, one.mappings.features), k
    # samples that cannot share a pass fall back to the loop + collation
    data, images = inputs()
    images[2].mask = None                          # not the mask the other samples share
    assert t.prepare_batch(data, images) is None


def test_neighborhood_mapping_features():
    """NeighborhoodBasedMappingFeatures on the GPU: against the reference's
    output (fixture) and, on a C1-sized cloud mapped by MapImages, against the
    oracle (exact k-NN compared index by index, features to 1e-6)."""
    from mvfusion.images import SameSettingImageData
    from mvfusion.mapping import ImageMapping
    from mvfusion.transforms import MapImages, NeighborhoodBasedMappingFeatures, knn
    import os
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "neighborhood_features.npz")))
    m = ImageMapping.from_csr_arrays(
        T(g["pointers"]).to(DEV), T(g["images"]).to(DEV), T(g["pixels"]).to(DEV),
        T(g["features_before"]).to(DEV), view_pointers=T(g["view_pointers"]).to(DEV))
    images = SameSettingImageData(pos=torch.zeros(5, 3), ref_size=(200, 200), proj_upscale=1,
                                  mappings=m)
    data = Data()
    data.pos = T(g["xyz"])
    t = NeighborhoodBasedMappingFeatures(k=[5, 20], voxel=float(g["voxel"]))
    _, out = t(data, images)
    np.testing.assert_allclose(npy(out.mappings.features), g["features_after"], rtol=1e-6, atol=1e-7)

    # C1-sized: 100k points, mappings from MapImages
    sc = S.make_scene("c1", seed=51)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    sub = np.random.default_rng(3).choice(N, 400, replace=False)
    got = npy(knn(T(xyz).to(DEV), 20))
    assert np.array_equal(got[:, 0], np.arange(N))
    # brute-force check of a sample of the queries against all points
    p64 = xyz.astype(np.float64)
    d = p64[sub, None, :] - p64[None, :, :]
    d2 = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
    want = np.argsort(d2, axis=1, kind="stable")[:, :20]
    assert np.array_equal(got[sub], want)
    data, images = make_inputs(xyz, np.arange(N), sc, sc["img_size"], attrs=False)
    tm = MapImages(method="SplattingVisibility", cylinder=True, exact=True,
                   point_order="index", **VIS)
    data, images = tm(data, images)
    before = npy(images.mappings.features)
    t = NeighborhoodBasedMappingFeatures(k=20, voxel=0.05)
    _, images = t(data, images)
    cols = orc.neighborhood_features(xyz, got, npy(images.mappings.pointers),
                                     npy(images.mappings.images), [20], voxel=0.05)
    after = npy(images.mappings.features)
    assert after.shape[1] == before.shape[1] + 2
    assert np.array_equal(after[:, :before.shape[1]], before)
    np.testing.assert_allclose(after[:, before.shape[1]:], cols, rtol=1e-6, atol=1e-7)


def test_view_sharded_attentive_pool_equals_single_pass():
    """Attentive pool on a view-sharded scene: two blocks of views mapped and
    pooled to view rows separately (two ranks), the rows of every block of
    points merged as the all-to-all delivers them to the block's owner, the
    QKV pool run per owner; against one pass over all views.  The merged rows
    equal the single-pass rows bit for bit; the pooled features agree to 1e-6 (the
    32-row tiles of the pool's last launch fall differently on the owner's rows)."""
    from mvfusion import distributed as D
    from mvfusion import pooling
    from mvfusion.pooling import gather_atomic_pool
    from mvfusion.transforms import MapImages
    sc = S.make_scene("c1", seed=41)
    xyz = sc["xyz"]
    N, (W, H), V, C_feat = xyz.shape[0], sc["img_size"], sc["pos"].shape[0], 8
    rng = np.random.default_rng(42)
    x_all = T(rng.normal(size=(V, C_feat, H, W)).astype(np.float32)).to(DEV)
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True,
                  point_order="index", **VIS)

    def run(view_ids):
        data, images = make_inputs(xyz, np.arange(N), sc, (W, H), attrs=False)
        sub = images[torch.as_tensor(list(view_ids))]
        _, out = t(data, sub)
        kept = [k for k in range(sub.num_views) if any(torch.equal(sub.pos[k], p) for p in out.pos)]
        x = x_all[torch.as_tensor(list(view_ids))[kept].to(DEV)]
        m = out.mappings.to(DEV)
        return gather_atomic_pool(x, m, "max"), m.features, m.view_csr_indexing

    xv, xm, ptr = run(range(V))
    torch.manual_seed(3)
    pool = pooling.QKVBimodalCSRPool(in_main=8, in_map=xm.shape[1], in_mod=C_feat, out_mod=C_feat,
                                     num_groups=1, nc_qk=8, gating=False).to(DEV).eval()
    x_main = T(rng.random((N, 8)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        want = pool(x_main, xv, xm, ptr)
    assert float(want.abs().max()) > 0
    parts = [run(D.partition_views(V, r, 2)) for r in range(2)]
    outs = []
    for o in range(2):
        own = D.partition_points(N, o, 2)
        counts = [(p[2][1:] - p[2][:-1])[own.start:own.stop] for p in parts]
        rows = [[p[0][int(p[2][own.start]):int(p[2][own.stop])],
                 p[1][int(p[2][own.start]):int(p[2][own.stop])]] for p in parts]
        optr, (ox, om) = D.merge_view_rows(counts, rows)
        a, b = int(ptr[own.start]), int(ptr[own.stop])
        assert torch.equal(optr, ptr[own.start:own.stop + 1] - a)
        assert torch.equal(ox, xv[a:b]) and torch.equal(om, xm[a:b])
        outs.append(pool(x_main[own.start:own.stop], ox, om, optr))
    got = torch.cat(outs)
    scale = float(want.abs().max())
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6 * scale)
    # single process: the exchange is the identity
    one = D.view_sharded_attentive_pool(pool, x_main, xv, xm, ptr)
    assert torch.equal(one, pool(x_main, xv, xm, ptr)) or torch.allclose(one, want, rtol=1e-6, atol=1e-6 * scale)


def test_c3_qkv_pool_full_size(mlp_precision):
    """C3: QKVBimodalCSRPool on 500 k points x min(Poisson(4), 32) views (2 M view rows), 64
    modality channels, 16-channel keys.  The full-size forward (value encoder and element
    encoder with their pooling epilogues) against (i) the separate launches that save_last
    keeps, (ii) the same module run on the second half of the points alone (points are
    independent; the 32-row tiles of the epilogues fall differently), (iii) the CPU oracle
    on the first 3000 points."""
    import mvf_oracle as orc
    from mvfusion import pooling
    rng = np.random.default_rng(3)
    N, C = 500_000, 64
    counts = np.minimum(rng.poisson(4, N), 32)
    ptr_np = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    Vw = int(ptr_np[-1])
    x_main = rng.random((N, 32), dtype=np.float32)
    x_mod = rng.random((Vw, C), dtype=np.float32) - 0.5
    x_map = rng.random((Vw, 8), dtype=np.float32)
    torch.manual_seed(7)
    mod = pooling.QKVBimodalCSRPool(in_main=32, in_map=8, in_mod=C, out_mod=C, num_groups=1,
                                    nc_qk=16, gating=True)
    gen = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
        mod.G.weight.fill_(1.2)
        mod.G.bias.fill_(0.3)
    sd = {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}
    mod = mod.to(DEV).eval()
    T = lambda a: torch.from_numpy(a).to(DEV)
    xm, xo, xp, ptr = T(x_main), T(x_mod), T(x_map), T(ptr_np)
    out = mod(xm, xo, xp, ptr)
    assert out.shape == (N, C) and bool(torch.isfinite(out).all())
    scale = float(out.abs().max())
    assert scale > 1e-3
    assert float(out[T(counts == 0)].abs().max()) == 0.0
    mod.save_last = True
    plain = mod(xm, xo, xp, ptr)
    mod.save_last = False
    _close(out, plain.cpu().numpy())
    h = N // 2 + 13
    a = int(ptr_np[h])
    half = mod(xm[h:], xo[a:], xp[a:], ptr[h:] - a)
    _close(half, out[h:].cpu().numpy())
    k = 3000
    b = int(ptr_np[k])
    want = orc.qkv_pool(x_main[:k], x_mod[:b], x_map[:b], ptr_np[:k + 1], sd, num_groups=1, nc_qk=16,
                        gating=True, dim_scaling=True, group_scaling=False)
    _close(out[:k], want)


def test_attention_pool_long_segments_64_views():
    """Points with 64 views: their rows span three 32-row tiles of the fused pool
    epilogue, whose per-tile partial sums are added in tile order by a second
    launch.  The fused forward stays within 1e-5 (element-wise, relative) of the
    launches that fold every segment in row order, and two runs of the same forward
    give the same bits."""
    from mvfusion import pooling
    rng = np.random.default_rng(9)
    N, C, views = 20_000, 64, 64
    ptr_np = (np.arange(N + 1) * views).astype(np.int64)
    x_main = rng.random((N, 32), dtype=np.float32)
    x_mod = rng.random((N * views, C), dtype=np.float32) - 0.5
    x_map = rng.random((N * views, 8), dtype=np.float32)
    torch.manual_seed(11)
    mod = pooling.QKVBimodalCSRPool(in_main=32, in_map=8, in_mod=C, out_mod=C, num_groups=1,
                                    nc_qk=16, gating=False).to(DEV).eval()
    T = lambda a: torch.from_numpy(a).to(DEV)
    args = (T(x_main), T(x_mod), T(x_map), T(ptr_np))
    fused = mod(*args)
    again = mod(*args)
    mod.save_last = True
    ordered = mod(*args)                 # separate launches: every segment folded in row order
    scale = float(ordered.abs().max())
    assert scale > 1e-3
    _close(fused, ordered.cpu().numpy())
    assert torch.equal(fused, again)


def test_c4_full_share_batched_pass_vs_oracle_per_cylinder():
    """C4's real per-GPU share: 8 full KITTI-360-shaped cylinders (100 k points x 8 views
    each) mapped by ONE batched pass (MapImages.process_batch); every cylinder of the
    collated batch against the oracle's MapImages of that cylinder alone (KD-tree order,
    as the reference samples it)."""
    from mvfusion.mapping import ImageMappingBatch
    from mvfusion.transforms import MapImages, kdtree_order
    scenes = [S.make_scene("c1", seed=2 + k) for k in range(8)]
    pairs = [make_inputs(sc["xyz"], np.arange(sc["xyz"].shape[0]), sc, sc["img_size"], attrs=False)
             for sc in scenes]
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, **VIS)
    assert t.prepare_batch([p[0] for p in pairs], [p[1] for p in pairs]) is not None
    _, batch = t.process_batch([p[0] for p in pairs], [p[1] for p in pairs])
    assert isinstance(batch.mappings, ImageMappingBatch)
    assert batch.mappings.pointers.shape[0] == 8 * 100_000 + 1
    parts = batch.mappings.to_csr_list()
    assert len(parts) == 8
    total = 0
    for k, (sc, m) in enumerate(zip(scenes, parts)):
        n = sc["xyz"].shape[0]
        assert n == 100_000 and sc["pos"].shape[0] == 8
        ref = orc.map_images(sc["xyz"], np.arange(n), sc, order=kdtree_order(T(sc["xyz"]), True).numpy(),
                             ref_size=sc["img_size"], cylinder=True, exact=True, **VIS)
        check_mapping(m, ref, features_atol=1e-6)
        total += ref["images"].shape[0]
    assert total == batch.mappings.images.shape[0] > 8 * 20_000


def test_c5_128_views_mapping_on_a_point_subsample():
    """The dense scene with all of its 128 views of 4 MP in one depth-map pass (what one GPU
    holds when the scene is not sharded): the mappings of every 97th point against the oracle
    run on those points alone.  The depth test of a point depends on nothing but the point
    and the maps, so the rows of the subsample must agree exactly."""
    from mvfusion.transforms import MapImages
    from mvfusion.visibility import render_depth_maps
    V = 128
    sc = S.make_scene("c5", seed=4, n_views=V)
    xyz = sc["xyz"]
    N = xyz.shape[0]
    W, H = sc["img_size"]
    assert N == 2_000_000 and (W, H) == (2304, 1728)
    K = np.tile(np.eye(4, dtype=np.float32), (V, 1, 1))
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = sc["fx"], sc["fy"], sc["mx"], sc["my"]
    kw = {k: VIS[k] for k in ("voxel", "k_swell", "d_swell", "camera", "r_max", "r_min")}
    dm = render_depth_maps(T(xyz).to(DEV), T(sc["pos"]), T(K), T(sc["extrinsic"]), (W, H), **kw)
    vis = {k: VIS[k] for k in ("r_max", "r_min", "camera")}
    data, images = make_inputs(xyz, np.arange(N), sc, (W, H), attrs=False)
    images.depth_maps = dm
    t = MapImages(method="DepthBasedVisibility", cylinder=True, depth_threshold=0.05, **vis)
    _, out = t(data, images)
    m = out.mappings
    assert out.num_views == V and m.images.shape[0] > 5_000_000
    sub = np.arange(0, N, 97)
    # the oracle on the subsample; the maps go view by view so that the host copy stays small
    want_rows = [[] for _ in sub]
    for v in range(V):
        one = {k: (sc[k][v:v + 1] if isinstance(sc[k], np.ndarray) and sc[k].shape[:1] == (V,) else sc[k])
               for k in sc}
        try:
            ref = orc.map_images(xyz[sub], np.arange(sub.shape[0]), one, order=None, method="depth_map",
                                 ref_size=(W, H), cylinder=True, depth_maps=dm[v:v + 1].cpu().numpy(),
                                 depth_threshold=0.05, **vis)
        except ValueError:               # no point of the subsample is mapped in this view
            continue
        cnt = np.diff(ref["pointers"])
        for q in np.nonzero(cnt)[0]:
            r = ref["pointers"][q]
            want_rows[q].append((v, int(ref["pixels"][r, 0]), int(ref["pixels"][r, 1]),
                                 ref["features"][r].copy()))
    ptr = npy(m.pointers)
    imgs, pix, feat = npy(m.images), npy(m.pixels), npy(m.features)
    checked = 0
    for q, p in enumerate(sub):
        a, b = ptr[p], ptr[p + 1]
        assert b - a == len(want_rows[q]), (p, b - a, len(want_rows[q]))
        for r, (v, x, y, f) in zip(range(a, b), want_rows[q]):
            assert imgs[r] == v and pix[r, 0] == x and pix[r, 1] == y
            np.testing.assert_allclose(feat[r], f, rtol=0, atol=1e-6)
            checked += 1
    assert checked > 50_000
