"""GPU parity of the 'kitti360_fisheye' and 's3dis_equirectangular' cameras:
the HIP path against the CPU oracle on the same inputs.  Integer outputs and
float32 distances are compared exactly; the float64 pixel coordinates may
differ in the last bits (atan2 / acos / sin / exp come from different math
libraries on the two sides), which is bounded here at 1e-11 pixel."""
import numpy as np
import pytest
import torch

import mvf_oracle as orc
from mvfusion import synthetic as S
from test_oracle_golden import wide_camera_kwargs

pytestmark = pytest.mark.gpu

T = torch.from_numpy
FISHEYE = np.array([2.2134, 0.016, 0.4325, 1336.3, 1335.8, 716.9, 705.8], np.float32)
SPLAT = dict(voxel=0.05, k_swell=1.5, d_swell=1e6)


def dev(a):
    return None if a is None else T(np.ascontiguousarray(a)).cuda()


def npy(t):
    return t.detach().cpu().numpy()


def tens(kw):
    """numpy keyword arguments -> torch tensors where the host layer expects them"""
    return {k: (T(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
            for k, v in kw.items()}


def close_px(a, b):
    return a.shape == b.shape and (a.size == 0 or np.abs(a - b).max() <= 1e-11)


@pytest.mark.parametrize("name", ["fisheye", "equirect"])
def test_projection_and_splat_vs_oracle(golden, name):
    from mvfusion.visibility import camera_projection, visibility_from_splatting
    g = golden("wide_" + name)
    proj, splat_kw = wide_camera_kwargs(g)
    rng = np.random.default_rng(5)
    W, H = proj["img_size"]
    for mask in (None, rng.random((W, H)) > 0.3):
        ref = orc.camera_projection(g["xyz"], g["img_xyz"], img_mask=mask, r_max=20.0,
                                    r_min=0.5, **proj)
        out = camera_projection(dev(g["xyz"]), T(g["img_xyz"]), img_mask=dev(mask),
                                r_max=20.0, r_min=0.5, **tens(proj))
        idx, dist, x, y = (npy(o) for o in out)
        assert np.array_equal(idx, ref[0])
        assert np.array_equal(dist.view(np.uint32), ref[1].view(np.uint32))
        assert close_px(x, ref[2]) and close_px(y, ref[3])
        assert np.array_equal(np.floor(x), np.floor(ref[2]))
        assert np.array_equal(np.floor(y), np.floor(ref[3]))
    # z-buffer on the oracle's projections (identical inputs on both sides)
    idx1, dist, x, y = ref
    xyz1 = g["xyz"][idx1]
    for exact in (False, True):
        want = orc.visibility_from_splatting(x, y, dist, xyz1, exact=exact, **proj, **splat_kw)
        got = visibility_from_splatting(dev(x), dev(y), dev(dist), dev(xyz1), exact=exact,
                                        **tens(proj), **splat_kw)
        for a, b in zip(got, want):
            assert np.array_equal(npy(a), b), (name, exact)


def scene(cam, n_points, n_views, seed):
    """Cylinder scene of the synthetic generator with the views re-labelled as
    fisheye or equirectangular cameras at the same poses."""
    sc = S.make_scene("c1", seed=seed, n_points=n_points, n_views=n_views)
    V = sc["pos"].shape[0]
    cams = dict(pos=sc["pos"])
    if cam == "kitti360_fisheye":
        cams["extrinsic"] = sc["extrinsic"]
        cams["fisheye"] = np.tile(FISHEYE, (V, 1))
        size = (1400, 1400)
    else:
        yaw = np.linspace(0, 2 * np.pi, V, endpoint=False)
        cams["opk"] = np.stack([0.03 * np.cos(yaw), 0.05 * np.sin(yaw), yaw], 1).astype(np.float32)
        size = (2048, 1024)
    return sc["xyz"], cams, size


def model_kwargs(cams, v):
    kw = {}
    if "extrinsic" in cams:
        kw["img_extrinsic"] = cams["extrinsic"][v]
    if "fisheye" in cams:
        kw["img_intrinsic_fisheye"] = cams["fisheye"][v]
    if "opk" in cams:
        kw["img_opk"] = cams["opk"][v]
    return kw


@pytest.mark.parametrize("cam", ["kitti360_fisheye", "s3dis_equirectangular"])
def test_splatting_model_vs_oracle(cam):
    from mvfusion.visibility import SplattingVisibility
    xyz, cams, size = scene(cam, 60000, 4, seed=11)
    nrm, lin, pla, sca = S.surface_normals(xyz)
    vis = dict(img_size=size, crop_top=16, crop_bottom=8, r_max=20, r_min=0.5, camera=cam)
    model = SplattingVisibility(exact=True, **SPLAT, **vis)
    v = 1
    img_xyz = cams["pos"][v].astype(np.float32)
    ref = orc.visibility_model(xyz, img_xyz, method="splatting", linearity=lin, planarity=pla,
                               scattering=sca, normals=nrm, exact=True, **SPLAT, **vis,
                               **model_kwargs(cams, v))
    out = model(dev(xyz), T(img_xyz), linearity=dev(lin), planarity=dev(pla),
                scattering=dev(sca), normals=dev(nrm), **tens(model_kwargs(cams, v)))
    assert ref["idx"].shape[0] > 1000
    for k in ("idx", "x", "y"):
        assert np.array_equal(npy(out[k]), ref[k]), k
    assert np.array_equal(npy(out["depth"]).view(np.uint32), ref["depth"].view(np.uint32))
    np.testing.assert_allclose(npy(out["features"]), ref["features"], rtol=0, atol=1e-6)


def test_depth_map_model_equirectangular():
    """DepthBasedVisibility on an S3DIS-like equirectangular view: the depth
    map is rendered from the oracle's own non-exact splat z-buffer and
    quantised like the dataset's 16-bit PNGs (1/512 m)."""
    from mvfusion.visibility import DepthBasedVisibility
    cam = "s3dis_equirectangular"
    xyz, cams, size = scene(cam, 60000, 4, seed=12)
    v = 2
    img_xyz = cams["pos"][v].astype(np.float32)
    vis = dict(img_size=size, crop_top=0, crop_bottom=0, r_max=20, r_min=0.5, camera=cam)
    idx1, dist, x, y = orc.camera_projection(xyz, img_xyz, **model_kwargs(cams, v), **vis)
    i2, xp, yp = orc.visibility_from_splatting(x, y, dist, xyz[idx1], exact=False, **SPLAT, **vis)
    depth_map = np.full(size, -1.0, np.float32)
    depth_map[xp, yp] = np.round(dist[i2] * np.float32(512)) / np.float32(512)
    ref = orc.visibility_model(xyz, img_xyz, method="depth_map", depth_map=depth_map,
                               depth_threshold=0.05, **vis, **model_kwargs(cams, v))
    model = DepthBasedVisibility(depth_threshold=0.05, **vis)
    out = model(dev(xyz), T(img_xyz), depth_map=dev(depth_map), **tens(model_kwargs(cams, v)))
    assert ref["idx"].shape[0] > 1000
    for k in ("idx", "x", "y"):
        assert np.array_equal(npy(out[k]).astype(np.int64), np.asarray(ref[k]).astype(np.int64)), k
    np.testing.assert_allclose(npy(out["features"]), ref["features"], rtol=0, atol=1e-6)


class Data:
    def clone(self):
        d = Data()
        d.__dict__.update(self.__dict__)
        return d


@pytest.mark.parametrize("cam", ["kitti360_fisheye", "s3dis_equirectangular"])
def test_map_images_wide_cameras(cam):
    """MapImages with fisheye / equirectangular images (fused all-view pass,
    then the view-by-view path) against the oracle's MapImages: nested CSR,
    pixels and features."""
    from mvfusion.images import SameSettingImageData
    from mvfusion.transforms import MapImages
    from test_gpu_parity import check_mapping
    xyz, cams, size = scene(cam, 40000, 4, seed=13)
    N = xyz.shape[0]
    ref_size = (size[0] // 2, size[1] // 2)
    vis = dict(r_max=20, r_min=0.5, camera=cam, exact=True, **SPLAT)
    ref = orc.map_images(xyz, np.arange(N), cams, order=None, ref_size=ref_size, proj_upscale=2,
                         cylinder=True, **vis)
    data = Data()
    data.pos = T(xyz)
    data.mapping_index = torch.arange(N)
    extra = {}
    if "fisheye" in cams:
        f = T(cams["fisheye"])
        extra = dict(extrinsic=T(cams["extrinsic"]), xi=f[:, 0], k1=f[:, 1], k2=f[:, 2],
                     gamma1=f[:, 3], gamma2=f[:, 4], u0=f[:, 5], v0=f[:, 6])
    else:
        extra = dict(opk=T(cams["opk"]))
    images = SameSettingImageData(pos=T(cams["pos"]), ref_size=ref_size, proj_upscale=2, **extra)
    t = MapImages(method="SplattingVisibility", cylinder=True, point_order="index", **vis)
    assert t._fusable(images, t._settle(data, images))       # all views in one launch set
    _, out = t(data, images)
    assert out.num_views == ref["seen_images"].shape[0]
    check_mapping(out.mappings, ref)
    # the view-by-view path gives the same mappings
    images2 = SameSettingImageData(pos=T(cams["pos"]), ref_size=ref_size, proj_upscale=2, **extra)
    m2, seen = t._process_per_view(data, images2, t._settle(data, images2), torch.device("cuda"))
    check_mapping(m2, ref)
