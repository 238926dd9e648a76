"""GPU parity: the HIP path (through the Python host layer and the C ABI)
against the golden vectors of the reference and against the CPU oracle on the
same seeded inputs.  Integer outputs must be bit-exact; float32 distances and
projections bit-exact; mapping features to 1e-6 abs; pooled features to 1e-5
relative (max / min bit-exact)."""
import numpy as np
import pytest
import torch

import mvf_oracle as orc
from mvfusion import synthetic as S

pytestmark = pytest.mark.gpu

T = torch.from_numpy
VIS = dict(voxel=0.05, r_max=20, r_min=0, k_swell=1.5, d_swell=1e6,
           camera="kitti360_perspective")


def dev(a):
    return T(np.ascontiguousarray(a)).cuda()


def npy(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ projection
@pytest.mark.parametrize("name", ["plain", "crop_mask", "scannet", "tiny"])
def test_camera_projection_golden(golden, name):
    from mvfusion.visibility import camera_projection
    g = golden("proj_" + name)
    mask = dev(g["mask"]) if g["mask"].size else None
    idx, dist, x, y = camera_projection(
        dev(g["xyz"]), T(g["img_xyz"]), img_intrinsic_pinhole=T(g["intrinsic"]),
        img_extrinsic=T(g["extrinsic"]), img_mask=mask, img_size=tuple(g["img_size"]),
        crop_top=int(g["crop_top"]), crop_bottom=int(g["crop_bottom"]),
        r_max=float(g["r_max"]), r_min=float(g["r_min"]), camera=str(g["camera"]))
    assert idx.is_cuda and idx.dtype == torch.int64 and x.dtype == torch.float64
    assert np.array_equal(npy(idx), g["indices"])
    assert np.array_equal(npy(dist).view(np.uint32), g["dist"].view(np.uint32))
    assert np.array_equal(npy(x), g["x_proj"])
    assert np.array_equal(npy(y), g["y_proj"])


def test_camera_projection_empty_and_cpu_input(golden):
    from mvfusion.visibility import camera_projection
    g = golden("proj_plain")
    # nothing in range -> empty outputs
    out = camera_projection(
        dev(g["xyz"]), T(g["img_xyz"]), img_intrinsic_pinhole=T(g["intrinsic"]),
        img_extrinsic=T(g["extrinsic"]), img_size=tuple(g["img_size"]),
        r_max=1e-3, r_min=0, camera="kitti360_perspective")
    assert all(o.shape[0] == 0 for o in out)
    # CPU tensors are staged to the GPU and results come back on the CPU
    idx, dist, x, y = camera_projection(
        T(g["xyz"]), T(g["img_xyz"]), img_intrinsic_pinhole=T(g["intrinsic"]),
        img_extrinsic=T(g["extrinsic"]), img_size=tuple(g["img_size"]),
        r_max=20, r_min=0, camera="kitti360_perspective")
    assert not idx.is_cuda and np.array_equal(idx.numpy(), g["indices"])
    with pytest.raises(ValueError):
        camera_projection(dev(g["xyz"]), T(g["img_xyz"]), camera="nope")


# ------------------------------------------------------------------- z-buffer
@pytest.mark.parametrize("name", ["nocrop", "crop"])
@pytest.mark.parametrize("exact", [False, True])
def test_splat_visibility_golden(golden, name, exact):
    from mvfusion.visibility import visibility_from_splatting
    g = golden("vis_" + name)
    idx, x, y = visibility_from_splatting(
        dev(g["x_proj"]), dev(g["y_proj"]), dev(g["dist"]), None,
        img_intrinsic_pinhole=T(g["intrinsic"]), img_size=tuple(g["img_size"]),
        crop_top=int(g["crop_top"]), crop_bottom=int(g["crop_bottom"]),
        voxel=float(g["voxel"]), k_swell=float(g["k_swell"]), d_swell=float(g["d_swell"]),
        exact=exact, camera="kitti360_perspective")
    tag = "exact" if exact else "splat"
    assert np.array_equal(npy(idx), g[tag + "_idx"])
    assert np.array_equal(npy(x), g[tag + "_x"])
    assert np.array_equal(npy(y), g[tag + "_y"])


def test_splat_visibility_deterministic_and_retry(golden):
    """Atomics make the kernel order-free: two runs are bit-identical; a too
    small pair buffer is detected and the call repeated."""
    from mvfusion import visibility as V
    g = golden("vis_nocrop")
    args = (dev(g["x_proj"]), dev(g["y_proj"]), dev(g["dist"]), None)
    kw = dict(img_intrinsic_pinhole=T(g["intrinsic"]), img_size=tuple(g["img_size"]),
              voxel=0.05, k_swell=1.5, d_swell=1e6, exact=False, camera="kitti360_perspective")
    a = V.visibility_from_splatting(*args, **kw)
    old = V._pair_factor[0]
    V._pair_factor[0] = 0.01            # force the overflow + retry path
    try:
        b = V.visibility_from_splatting(*args, **kw)
    finally:
        V._pair_factor[0] = max(old, V._pair_factor[0])
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    assert np.array_equal(npy(a[0]), g["splat_idx"])


@pytest.mark.parametrize("name", ["full", "bare"])
def test_visibility_model_golden(golden, name):
    from mvfusion.visibility import SplattingVisibility
    g = golden("model_" + name)
    nrm, lin, pla, sca = S.surface_normals(g["xyz"])
    use = bool(g["use_attrs"])
    model = SplattingVisibility(
        voxel=0.05, k_swell=1.5, d_swell=1e6, exact=True, img_size=tuple(g["img_size"]),
        crop_top=0, crop_bottom=0, r_max=20, r_min=0, camera="kitti360_perspective")
    out = model(
        dev(g["xyz"]), T(g["img_xyz"]), linearity=dev(lin) if use else None,
        planarity=dev(pla) if use else None, scattering=dev(sca) if use else None,
        normals=dev(nrm) if use else None, img_intrinsic_pinhole=T(g["intrinsic"]),
        img_extrinsic=T(g["extrinsic"]))
    assert np.array_equal(npy(out["idx"]), g["idx"])
    assert np.array_equal(npy(out["x"]), g["x"])
    assert np.array_equal(npy(out["y"]), g["y"])
    assert np.array_equal(npy(out["depth"]), g["depth"])
    np.testing.assert_allclose(npy(out["features"]), g["features"], rtol=0, atol=1e-6)


def test_depth_map_visibility_golden(golden):
    from mvfusion.visibility import visibility_from_depth_map
    g = golden("depth_map")
    idx, x, y = visibility_from_depth_map(
        dev(g["x_proj"]), dev(g["y_proj"]), dev(g["dist"]), depth_map=dev(g["depth_map"]),
        img_size=tuple(g["img_size"]), depth_threshold=float(g["depth_threshold"]))
    assert np.array_equal(npy(idx), g["indices"])
    assert np.array_equal(npy(x), g["x"]) and np.array_equal(npy(y), g["y"])


def test_depth_map_visibility_png_path(golden, tmp_path):
    from PIL import Image
    from mvfusion.visibility import visibility_from_depth_map
    g = golden("depth_map")
    path = str(tmp_path / "v_depth.png")
    Image.fromarray(g["raw_png"]).save(path)
    idx, x, y = visibility_from_depth_map(
        dev(g["x_proj"]), dev(g["y_proj"]), dev(g["dist"]), depth_map_path=path,
        img_size=tuple(g["img_size"]), depth_threshold=float(g["depth_threshold"]))
    assert np.array_equal(npy(idx), g["indices"])


# ------------------------------------------------------------------ MapImages
class Data:
    def clone(self):
        d = Data()
        d.__dict__.update(self.__dict__)
        return d


def make_inputs(xyz, mapping_index, cams, ref_size, proj_upscale=1, downscale=1,
                crop_size=None, crop_offsets=None, attrs=True, order=None):
    from mvfusion.images import SameSettingImageData
    data = Data()
    data.pos = T(xyz)
    data.mapping_index = T(np.asarray(mapping_index, np.int64))
    if attrs:
        nrm, lin, pla, sca = S.surface_normals(xyz)
        data.linearity, data.planarity, data.scattering, data.norm = T(lin), T(pla), T(sca), T(nrm)
    if order is not None:
        data.mvf_point_order = T(np.asarray(order, np.int64))
    images = SameSettingImageData(
        pos=T(cams["pos"]), extrinsic=T(cams["extrinsic"]), fx=T(cams["fx"]), fy=T(cams["fy"]),
        mx=T(cams["mx"]), my=T(cams["my"]), ref_size=tuple(ref_size), proj_upscale=proj_upscale,
        downscale=downscale, crop_size=crop_size,
        crop_offsets=None if crop_offsets is None else T(np.asarray(crop_offsets, np.int64)))
    return data, images


def canonical_pixels(view_ptr, pixels):
    out = pixels.copy()
    for a, b in zip(view_ptr[:-1], view_ptr[1:]):
        if b - a > 1:
            seg = pixels[a:b]
            out[a:b] = seg[np.lexsort((seg[:, 1], seg[:, 0]))]
    return out


def check_mapping(m, ref, features_atol=1e-6):
    assert np.array_equal(npy(m.pointers), ref["pointers"])
    assert np.array_equal(npy(m.images), ref["images"])
    assert np.array_equal(npy(m.values[1].pointers), ref["view_pointers"])
    assert npy(m.pixels).dtype == ref["pixels"].dtype
    assert np.array_equal(npy(m.pixels), canonical_pixels(ref["view_pointers"], ref["pixels"]))
    np.testing.assert_allclose(npy(m.features), ref["features"], rtol=0, atol=features_atol)


@pytest.mark.parametrize("name", ["mapimages_exact", "mapimages_exact_sphere", "mapimages_splat"])
def test_map_images_golden(golden, name):
    from mvfusion.transforms import MapImages, kdtree_order
    g = golden(name)
    cams = dict(pos=g["cam_pos"], extrinsic=g["extrinsic"], fx=g["fx"], fy=g["fy"],
                mx=g["mx"], my=g["my"])
    data, images = make_inputs(
        g["xyz"], g["mapping_index"], cams, g["ref_size"], g["proj_upscale"].item(),
        g["downscale"].item(), tuple(g["crop_size"]), g["crop_offsets"],
        attrs=bool(g["use_attrs"]))
    # the KD-tree traversal order is rebuilt on the host, as the reference does
    assert np.array_equal(kdtree_order(data.pos, bool(g["cylinder"])).numpy(), g["order"])
    t = MapImages(method="SplattingVisibility", cylinder=bool(g["cylinder"]),
                  exact=bool(g["exact"]), **VIS)
    _, out = t(data, images)
    assert out.num_views == int(g["out_num_views"])
    assert np.array_equal(npy(out.pos), g["out_pos"])
    check_mapping(out.mappings, dict(
        pointers=g["pointers"], images=g["images"], view_pointers=g["view_pointers"],
        pixels=g["pixels"], features=g["features"]))
    assert out.mappings.pixels.dtype == torch.int16


@pytest.mark.parametrize("cfg,order_kind", [("c1", "kdtree"), ("c1", "index"), ("c2", "index")])
def test_map_images_vs_oracle_full_size(cfg, order_kind):
    """BASELINE-sized samples: fused HIP MapImages against the CPU oracle."""
    from mvfusion.transforms import MapImages, kdtree_order
    sc = S.make_scene(cfg, seed=1 if cfg == "c1" else 2)
    N = sc["xyz"].shape[0]
    order = kdtree_order(T(sc["xyz"]), True).numpy() if order_kind == "kdtree" else None
    ref = orc.map_images(sc["xyz"], np.arange(N), sc, order=order, ref_size=sc["img_size"],
                         cylinder=True, exact=True, **VIS)
    data, images = make_inputs(sc["xyz"], np.arange(N), sc, sc["img_size"], attrs=False,
                               order=order)
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True,
                  point_order="index", **VIS)
    _, out = t(data, images)
    check_mapping(out.mappings, ref)
    # second call on the same transform reuses the persistent workspace
    data2, images2 = make_inputs(sc["xyz"], np.arange(N), sc, sc["img_size"], attrs=False,
                                 order=order)
    _, out2 = t(data2, images2)
    assert torch.equal(out.mappings.pointers, out2.mappings.pointers)
    assert torch.equal(out.mappings.pixels, out2.mappings.pixels)
    assert torch.equal(out.mappings.features, out2.mappings.features)


def test_fused_plan_async_matches_sync():
    """Pipelined passes (two in flight, alternating output buffers, sizes read
    from pinned memory) return exactly what the synchronous pass returns,
    including after an overflow of deliberately small buffers."""
    from mvfusion import visibility as visibility_module
    from mvfusion.transforms import MapImages
    sc = S.make_scene("c1", seed=5, n_points=30000, n_views=4)
    N = sc["xyz"].shape[0]
    data, images = make_inputs(sc["xyz"], np.arange(N), sc, sc["img_size"], attrs=True)
    t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, **VIS)
    model = visibility_module.SplattingVisibility(img_size=images.proj_size, exact=True, **VIS)
    plan = t.prepare_fused(data, images, model, torch.device("cuda", 0))
    want = [x.clone() for x in plan.run()]
    a = plan.run_async()
    b = plan.run_async()
    ra = [x.clone() for x in a.result()]
    c = plan.run_async()
    rb = [x.clone() for x in b.result()]
    rc = [x.clone() for x in c.result()]
    for got in (ra, rb, rc):
        assert len(got) == len(want)
        for g_, w_ in zip(got, want):
            assert torch.equal(g_, w_)
    # too-small buffers: the pending pass reports the overflow and is repeated
    plan.out_cap, plan.pair_cap = 16, 16
    rd = [x.clone() for x in plan.run_async().result()]
    for g_, w_ in zip(rd, want):
        assert torch.equal(g_, w_)
    assert plan.out_cap > 16 and plan.pair_cap > 16
    # a centre-pixel table far too small for the seen pairs: reported, grown, repeated
    plan.centre_slots = 1
    re_ = [x.clone() for x in plan.run()]
    for g_, w_ in zip(re_, want):
        assert torch.equal(g_, w_)
    assert plan.centre_slots >= want[1].shape[0]


def test_fused_plans_on_concurrent_streams():
    """Three plans over two different samples driven on three streams at once
    (what bench.py does with consecutive steps): every pass returns exactly
    what a lone synchronous pass returns.  The plans own separate workspaces
    (tile bins, seen flags, centre-pixel table), so passes in flight share
    nothing but their inputs."""
    from mvfusion import visibility as visibility_module
    from mvfusion.transforms import MapImages
    dev0 = torch.device("cuda", 0)
    plans, wants = [], []
    for seed in (6, 7):
        sc = S.make_scene("c1", seed=seed)
        N = sc["xyz"].shape[0]
        data, images = make_inputs(sc["xyz"], np.arange(N), sc, sc["img_size"], attrs=False)
        t = MapImages(method="SplattingVisibility", cylinder=True, exact=True, **VIS)
        model = visibility_module.SplattingVisibility(img_size=images.proj_size, exact=True, **VIS)
        plan = t.prepare_fused(data, images, model, dev0)
        plans.append(plan)
        wants.append([x.clone() for x in plan.run()])
    plans.append(plans[0].twin(lane=2))
    wants.append(wants[0])
    plans[1].lane = 1
    assert len({p.lane for p in plans}) == 3
    streams = [torch.cuda.Stream(dev0) for _ in plans]
    torch.cuda.synchronize()
    for _ in range(4):
        pending = []
        for plan, s in zip(plans, streams):
            with torch.cuda.stream(s):
                pending.append((plan.run_async(), plan.run_async()))
        for (p0, p1), s, want in zip(pending, streams, wants):
            with torch.cuda.stream(s):
                for p in (p0, p1):
                    got = [x.clone() for x in p.result()]
                    s.synchronize()
                    for g_, w_ in zip(got, want):
                        assert torch.equal(g_, w_)
    torch.cuda.synchronize()


@pytest.mark.parametrize("exact", [True, False])
def test_map_images_no_mapping_raises(exact):
    """No point in range of any camera: the fused passes (exact and plain splat: no
    tile, no pair, no pixel) come back empty and MapImages raises as the reference does."""
    from mvfusion.transforms import MapImages
    sc = S.make_scene("c1", seed=3, n_points=2000, n_views=2)
    xyz = sc["xyz"] + np.float32(500.0)          # far away from every camera
    data, images = make_inputs(xyz, np.arange(2000), sc, sc["img_size"], attrs=False)
    t = MapImages(method="SplattingVisibility", cylinder=True, point_order="index",
                  **dict(VIS, exact=exact))
    model = t._settle(data, images)
    assert t._fusable(images, model)
    plan = t.prepare_fused(data, images, model, torch.device("cuda", 0))
    res = plan.run()
    assert plan.last_counts["mappings"] == 0 and plan.last_counts["pixels"] == 0
    assert int(res[0][-1]) == 0 and res[1].shape[0] == 0 and res[2].shape[0] == 0
    with pytest.raises(ValueError):
        t(data, images)


# -------------------------------------------------------------------- pooling
def test_segment_csr_golden(golden):
    from mvfusion.pooling import BimodalCSRPool
    g = golden("segment_csr")
    for C in ("1", "13", "64", "96", "doc"):
        x, ptr = dev(g[f"x_{C}"]), dev(g[f"ptr_{C}"])
        for mode in ("max", "mean", "min", "sum"):
            out = npy(BimodalCSRPool(mode=mode)(None, x, None, ptr))
            ref = g[f"out_{C}_{mode}"]
            if mode in ("max", "min"):
                assert np.array_equal(out, ref), (C, mode)
            else:
                np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
            # rows are folded in order: identical to the sequential CPU fold
            assert np.array_equal(out, orc.segment_csr(g[f"x_{C}"], g[f"ptr_{C}"], mode))


def test_segment_csr_edge_cases():
    from mvfusion.pooling import segment_csr
    x = torch.rand(10, 8, device="cuda")
    # all-empty segments, 1-D source, pointers not covering all rows
    out = segment_csr(x, torch.zeros(5, dtype=torch.int64, device="cuda"), reduce="max")
    assert out.shape == (4, 8) and float(out.abs().sum()) == 0.0
    out = segment_csr(x[:, 0].contiguous(), torch.tensor([0, 3, 3, 10]).cuda(), reduce="sum")
    ref = orc.segment_csr(npy(x[:, 0]), np.array([0, 3, 3, 10]), "sum")
    assert np.array_equal(npy(out), ref)
    out = segment_csr(x, torch.tensor([0], device="cuda"), reduce="mean")
    assert out.shape == (0, 8)
    with pytest.raises(RuntimeError):
        segment_csr(x.cpu(), torch.tensor([0, 10]))


def test_segment_softmax_golden(golden):
    from mvfusion import pooling as P
    g = golden("segment_softmax")
    np.testing.assert_allclose(
        npy(P.segment_softmax_csr(dev(g["doc_src"]), dev(g["doc_ptr"]))), g["doc_out"],
        rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(
        npy(P.segment_softmax_csr(dev(g["doc_src"]), dev(g["doc_ptr"]), scaling=True)),
        g["doc_out_scaled"], rtol=1e-5, atol=1e-9)
    for G in (1, 3, 4):
        x, ptr = dev(g[f"x_{G}"]), dev(g[f"ptr_{G}"])
        np.testing.assert_allclose(npy(P.segment_softmax_csr(x, ptr)), g[f"out_{G}"],
                                   rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(npy(P.segment_softmax_csr(x, ptr, scaling=True)),
                                   g[f"out_{G}_scaled"], rtol=1e-5, atol=1e-9)
        assert np.array_equal(npy(P.segment_gather_csr(x, ptr, reduce="max")), g[f"seggather_{G}"])
        assert np.array_equal(npy(P.expand_group_feat(x[:50], G, 10)), g[f"expand_{G}"])


def test_pooling_full_size_properties():
    """C2-sized view pooling (300k points, C = 13 and 64): vs the oracle, plus
    size-independent properties (sum linearity, max >= mean >= min)."""
    from mvfusion.pooling import segment_csr
    rng = np.random.default_rng(5)
    ptr = S.make_csr(300_000, 4, 16, rng)
    for C in (13, 64):
        x = rng.random((int(ptr[-1]), C), dtype=np.float32)
        xd, pd = dev(x), dev(ptr)
        outs = {m: npy(segment_csr(xd, pd, reduce=m)) for m in ("max", "mean", "min", "sum")}
        for m in outs:
            ref = orc.segment_csr(x, ptr, m)
            assert np.array_equal(outs[m], ref), (C, m)
        assert (outs["max"] >= outs["mean"] - 1e-6).all() and (outs["mean"] >= outs["min"] - 1e-6).all()
        two = npy(segment_csr(xd * 2, pd, reduce="sum"))
        assert np.array_equal(two, outs["sum"] * 2)
