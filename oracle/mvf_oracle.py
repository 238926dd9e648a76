"""CPU oracle for the multi-view image->point fusion path (numpy + C).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by ``__graft_entry__.smoke()``
and by the ``cpu_baseline`` leg of ``bench.py`` -- never by the shipped package.
The heavy loops live in ``mvf_oracle.c`` (built into ``_build/libmvf_oracle.so``
by ``oracle/Makefile``); this file restates the reference's numpy / torch glue
around them.  Citations are relative to
/root/reference/DeepViewAgg-Wildscenes/torch_points3d/.

Parity pin: the fixtures of tests/golden (made by tests/golden/make_golden.py,
which executes the reference's own source files).  Third-party arithmetic the
reference relies on and that is absent from its tree -- numba, torch_scatter
(``segment_csr``), sklearn ``KDTree`` -- is restated from its published
semantics; the KD-tree traversal order is an *input* here (``order``).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmvf_oracle.so")

CAMERAS = {"kitti360_perspective": 0, "scannet": 1, "kitti360_fisheye": 2,
           "s3dis_equirectangular": 3}
PINHOLE = ("kitti360_perspective", "scannet")
REDUCE = {"sum": 0, "add": 0, "mean": 1, "min": 2, "max": 3}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    try:
        return _LIB
    except NameError:
        pass
    if not os.path.exists(_SO):
        build()
    _LIB = C.CDLL(_SO)
    _LIB.orc_camera_projection.restype = C.c_longlong
    _LIB.orc_visibility_from_splatting.restype = C.c_longlong
    _LIB.orc_visibility_from_depth_map.restype = C.c_longlong
    _LIB.orc_postprocess_features.restype = C.c_int
    _LIB.orc_field_of_view.restype = C.c_longlong
    _LIB.orc_zbuffer_from_splat.restype = C.c_longlong
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# --------------------------------------------------------------------------- #
# visibility.py                                                               #
# --------------------------------------------------------------------------- #

def pose_to_rotation_matrix(opk):
    """pose_to_rotation_matrix_cpu, visibility.py:57-90 (float32 throughout)."""
    opk = np.asarray(opk, dtype=np.float32).reshape(3)
    co, so = np.cos(opk[0]), np.sin(opk[0])
    cp, sp = np.cos(opk[1]), np.sin(opk[1])
    ck, sk = np.cos(opk[2]), np.sin(opk[2])
    M_o = np.array([[1.0, 0.0, 0.0], [0.0, co, -so], [0.0, so, co]], dtype=np.float32)
    M_p = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]], dtype=np.float32)
    M_k = np.array([[ck, -sk, 0.0], [sk, ck, 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    return np.dot(M_o, np.dot(M_p, M_k))


def _norm(v):
    """norm_cpu, visibility.py:129-137: float32, squares added in column order."""
    v = _c(v, np.float32)
    s = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
    return np.sqrt(s)


def fisheye_projection(xyz, img_extrinsic, img_intrinsic_fisheye):
    """fisheye_projection_cpu, visibility.py:288-339 (numba typing)."""
    xyz = _c(xyz, np.float32)
    n = xyz.shape[0]
    x, y, z = (np.empty(n, np.float64) for _ in range(3))
    _lib().orc_fisheye_projection(
        C.c_longlong(n), _p(xyz), _p(_c(img_extrinsic, np.float32)),
        _p(_c(img_intrinsic_fisheye, np.float32)), _p(x), _p(y), _p(z))
    return x, y, z


def equirectangular_projection(xyz_to_img, radius, img_opk, img_size):
    """equirectangular_projection_cpu, visibility.py:150-182 (numba typing)."""
    d = _c(xyz_to_img, np.float32)
    n = d.shape[0]
    R = np.ascontiguousarray(pose_to_rotation_matrix(img_opk), dtype=np.float32)
    x, y = np.empty(n, np.float64), np.empty(n, np.float64)
    _lib().orc_equirectangular_projection(
        C.c_longlong(n), _p(d), _p(_c(radius, np.float32)), _p(R),
        int(img_size[0]), int(img_size[1]), _p(x), _p(y))
    return x, y


def _camera_projection_wide(
        xyz, img_xyz, img_opk, img_intrinsic_fisheye, img_extrinsic, img_mask,
        img_size, crop_top, crop_bottom, r_max, r_min, camera):
    """camera_projection_cpu, visibility.py:478-538, fisheye / equirectangular
    branches (:519-527)."""
    xyz = _c(xyz, np.float32)
    img_xyz = _c(img_xyz, np.float32).reshape(3)
    indices = np.arange(xyz.shape[0])
    dist = _norm(xyz - img_xyz)
    in_range = np.where(np.logical_and(r_min < dist.astype(np.float64),
                                       dist.astype(np.float64) < r_max))[0]
    xyz, dist, indices = xyz[in_range], dist[in_range], indices[in_range]
    if camera == "kitti360_fisheye":
        E = np.eye(4, dtype=np.float32) if img_extrinsic is None else img_extrinsic
        f = np.ones(7, np.float32) if img_intrinsic_fisheye is None else img_intrinsic_fisheye
        x, y, z = fisheye_projection(xyz, E, f)
    else:
        opk = np.zeros(3, np.float32) if img_opk is None else img_opk
        x, y = equirectangular_projection(xyz - img_xyz, dist, opk, img_size)
        z = None
    W, H = int(img_size[0]), int(img_size[1])
    mask = None
    if img_mask is not None:
        assert tuple(img_mask.shape) == (W, H)
        mask = _c(img_mask, np.uint8)
    keep = np.empty(x.shape[0], np.int64)
    m = _lib().orc_field_of_view(
        C.c_longlong(x.shape[0]), _p(x), _p(y), _p(z), _p(mask), W, H,
        int(crop_top), int(crop_bottom), _p(keep))
    keep = keep[:m]
    return indices[keep].astype(np.int64), dist[keep], x[keep], y[keep]


def camera_projection(
        xyz, img_xyz, img_opk=None, img_intrinsic_pinhole=None, img_intrinsic_fisheye=None,
        img_extrinsic=None, img_mask=None, img_size=(1024, 512), crop_top=0, crop_bottom=0,
        r_max=30, r_min=0.5, camera="kitti360_perspective", **kwargs):
    """camera_projection_cpu, visibility.py:478-538."""
    if camera not in PINHOLE:
        if camera not in CAMERAS:
            raise ValueError(camera)
        return _camera_projection_wide(
            xyz, img_xyz, img_opk, img_intrinsic_fisheye, img_extrinsic, img_mask,
            img_size, crop_top, crop_bottom, r_max, r_min, camera)
    xyz = _c(xyz, np.float32)
    n = xyz.shape[0]
    K = np.eye(4, dtype=np.float32) if img_intrinsic_pinhole is None \
        else _c(img_intrinsic_pinhole, np.float32)
    E = np.eye(4, dtype=np.float32) if img_extrinsic is None \
        else _c(img_extrinsic, np.float32)
    if camera == "scannet":
        E = np.ascontiguousarray(np.linalg.inv(E))          # visibility.py:233
    W, H = int(img_size[0]), int(img_size[1])
    mask = None
    if img_mask is not None:
        assert tuple(img_mask.shape) == (W, H)
        mask = _c(img_mask, np.uint8)
    idx = np.empty(n, np.int64)
    dist = np.empty(n, np.float32)
    x = np.empty(n, np.float64)
    y = np.empty(n, np.float64)
    m = _lib().orc_camera_projection(
        C.c_longlong(n), _p(xyz), _p(_c(img_xyz, np.float32)), _p(E),
        C.c_float(K[0, 0]), C.c_float(K[1, 1]), C.c_float(K[0, 2]), C.c_float(K[1, 2]),
        _p(mask), W, H, int(crop_top), int(crop_bottom),
        C.c_double(r_max), C.c_double(r_min), CAMERAS[camera],
        _p(idx), _p(dist), _p(x), _p(y))
    return idx[:m].copy(), dist[:m].copy(), x[:m].copy(), y[:m].copy()


def pinhole_splat(
        x_proj, y_proj, dist, img_intrinsic_pinhole, img_size=(1024, 512),
        crop_top=0, crop_bottom=0, voxel=0.02, k_swell=1.0, d_swell=1000):
    """pinhole_splat_cpu, visibility.py:761-827."""
    K = _c(img_intrinsic_pinhole, np.float32)
    m = x_proj.shape[0]
    splat = np.empty((m, 4), np.int32)
    _lib().orc_pinhole_splat(
        C.c_longlong(m), _p(_c(x_proj, np.float64)), _p(_c(y_proj, np.float64)),
        _p(_c(dist, np.float32)), C.c_float(K[0, 0]), C.c_float(K[1, 1]),
        int(img_size[0]), int(img_size[1]), int(crop_top), int(crop_bottom),
        C.c_double(voxel), C.c_double(k_swell), C.c_double(d_swell), _p(splat))
    return splat


def equirectangular_splat(
        x_proj, y_proj, dist, img_size=(1024, 512), crop_top=0, crop_bottom=0,
        voxel=0.02, k_swell=1.0, d_swell=1000):
    """equirectangular_splat_cpu, visibility.py:630-704."""
    m = x_proj.shape[0]
    splat = np.empty((m, 4), np.int32)
    _lib().orc_equirectangular_splat(
        C.c_longlong(m), _p(_c(x_proj, np.float64)), _p(_c(y_proj, np.float64)),
        _p(_c(dist, np.float32)), int(img_size[0]), int(img_size[1]), int(crop_top),
        int(crop_bottom), C.c_double(voxel), C.c_double(k_swell), C.c_double(d_swell),
        _p(splat))
    return splat


def fisheye_splat(
        x_proj, y_proj, xyz, img_extrinsic, img_intrinsic_fisheye, img_size=(1024, 512),
        crop_top=0, crop_bottom=0, voxel=0.02, k_swell=1.0, d_swell=1000):
    """fisheye_splat_cpu, visibility.py:876-953."""
    m = x_proj.shape[0]
    splat = np.empty((m, 4), np.int32)
    _lib().orc_fisheye_splat(
        C.c_longlong(m), _p(_c(x_proj, np.float64)), _p(_c(y_proj, np.float64)),
        _p(_c(xyz, np.float32)), _p(_c(img_extrinsic, np.float32)),
        _p(_c(img_intrinsic_fisheye, np.float32)), int(img_size[0]), int(img_size[1]),
        int(crop_top), int(crop_bottom), C.c_double(voxel), C.c_double(k_swell),
        C.c_double(d_swell), _p(splat))
    return splat


def _zbuffer(x_proj, y_proj, dist, splat, img_size, crop_top, crop_bottom, exact):
    """z-buffer part of visibility_from_splatting_cpu, visibility.py:1134-1195."""
    m = x_proj.shape[0]
    W, H = int(img_size[0]), int(img_size[1])
    cap = min(m, W * (H - crop_top - crop_bottom)) if exact \
        else W * (H - crop_top - crop_bottom)
    idx, xo, yo = (np.empty(cap, np.int64) for _ in range(3))
    k = _lib().orc_zbuffer_from_splat(
        C.c_longlong(m), _p(_c(x_proj, np.float64)), _p(_c(y_proj, np.float64)),
        _p(_c(dist, np.float32)), _p(np.ascontiguousarray(splat, np.int32)), W, H,
        int(crop_top), int(crop_bottom), int(bool(exact)), _p(idx), _p(xo), _p(yo))
    if k < 0:
        raise MemoryError
    return idx[:k].copy(), xo[:k].copy(), yo[:k].copy()


def visibility_from_splatting(
        x_proj, y_proj, dist, xyz=None, img_extrinsic=None,
        img_intrinsic_pinhole=None, img_intrinsic_fisheye=None, img_size=(1024, 512),
        crop_top=0, crop_bottom=0, voxel=0.1, k_swell=1.0, d_swell=1000, exact=False,
        camera="kitti360_perspective", **kwargs):
    """visibility_from_splatting_cpu, visibility.py:1073-1195."""
    assert camera in CAMERAS
    m = x_proj.shape[0]
    assert m > 0 and y_proj.shape[0] == m and dist.shape[0] == m
    if camera == "s3dis_equirectangular":
        splat = equirectangular_splat(x_proj, y_proj, dist, img_size, crop_top, crop_bottom,
                                      voxel, k_swell, d_swell)
        return _zbuffer(x_proj, y_proj, dist, splat, img_size, crop_top, crop_bottom, exact)
    if camera == "kitti360_fisheye":
        E = np.eye(4, dtype=np.float32) if img_extrinsic is None else img_extrinsic
        f = np.ones(7, np.float32) if img_intrinsic_fisheye is None else img_intrinsic_fisheye
        splat = fisheye_splat(x_proj, y_proj, xyz, E, f, img_size, crop_top, crop_bottom,
                              voxel, k_swell, d_swell)
        return _zbuffer(x_proj, y_proj, dist, splat, img_size, crop_top, crop_bottom, exact)
    K = np.eye(4, dtype=np.float32) if img_intrinsic_pinhole is None \
        else _c(img_intrinsic_pinhole, np.float32)
    W, H = int(img_size[0]), int(img_size[1])
    cap = min(m, W * (H - crop_top - crop_bottom)) if exact \
        else W * (H - crop_top - crop_bottom)
    idx = np.empty(cap, np.int64)
    xo = np.empty(cap, np.int64)
    yo = np.empty(cap, np.int64)
    k = _lib().orc_visibility_from_splatting(
        C.c_longlong(m), _p(_c(x_proj, np.float64)), _p(_c(y_proj, np.float64)),
        _p(_c(dist, np.float32)), C.c_float(K[0, 0]), C.c_float(K[1, 1]), W, H,
        int(crop_top), int(crop_bottom), C.c_double(voxel), C.c_double(k_swell),
        C.c_double(d_swell), int(bool(exact)), _p(idx), _p(xo), _p(yo))
    if k < 0:
        raise MemoryError
    return idx[:k].copy(), xo[:k].copy(), yo[:k].copy()


def visibility_from_depth_map(
        x_proj, y_proj, dist, depth_map=None, img_size=(1024, 512),
        depth_threshold=0.05, **kwargs):
    """visibility_from_depth_map, visibility.py:1361-1388, on a decoded
    [W, H] float32 map (what read_s3dis_depth_map :1328-1358 returns)."""
    m = x_proj.shape[0]
    assert m > 0
    W, H = int(img_size[0]), int(img_size[1])
    dm = _c(depth_map, np.float32)
    assert dm.shape == (W, H)
    x_proj = _c(x_proj, np.float64)
    y_proj = _c(y_proj, np.float64)
    idx = np.empty(m, np.int64)
    k = _lib().orc_visibility_from_depth_map(
        C.c_longlong(m), _p(x_proj), _p(y_proj), _p(_c(dist, np.float32)),
        _p(dm), W, H, C.c_float(np.float32(depth_threshold)), _p(idx))
    idx = idx[:k].copy()
    return idx, x_proj[idx], y_proj[idx]


def decode_s3dis_depth(raw_u16_hw):
    """read_s3dis_depth_map, visibility.py:1349-1356, from the decoded PNG
    array [H, W] uint16: transpose, /512 in float32, 65535 -> -1."""
    im = np.asarray(raw_u16_hw).T
    out = (im.astype(np.float32) / np.float32(512)).astype(np.float32)
    out[im == 2 ** 16 - 1] = -1
    return np.ascontiguousarray(out)


def postprocess_features(
        xyz_to_img, y_proj, dist, linearity, planarity, scattering, normals,
        img_size=(1024, 512), r_max=30, r_min=0.5, **kwargs):
    """postprocess_features, visibility.py:1548-1600."""
    k = dist.shape[0]
    F = 2 + sum(a is not None for a in (linearity, planarity, scattering, normals))
    out = np.empty((k, F), np.float32)
    f32 = lambda a: None if a is None else _c(a, np.float32)
    lin, pla, sca, nrm = f32(linearity), f32(planarity), f32(scattering), f32(normals)
    _lib().orc_postprocess_features(
        C.c_longlong(k), _p(_c(xyz_to_img, np.float32)), _p(_c(y_proj, np.float64)),
        _p(_c(dist, np.float32)), _p(lin), _p(pla), _p(sca), _p(nrm),
        int(img_size[1]), C.c_double(r_max), C.c_double(r_min), _p(out))
    return out


def visibility_model(
        xyz, img_xyz, method="splatting", linearity=None, planarity=None,
        scattering=None, normals=None, **kw):
    """VisibilityModel.__call__, visibility.py:1717-1775 (and the functional
    twin :1607-1692).  ``kw`` holds the model's attributes and per-image
    arguments.  Returns dict idx, x, y, depth, features."""
    xyz = _c(xyz, np.float32)
    img_xyz = _c(img_xyz, np.float32)
    idx_1, dist, x_proj, y_proj = camera_projection(xyz, img_xyz, **kw)
    if x_proj.shape[0] == 0:
        z = np.empty(0, np.int64)
        return dict(idx=z, x=z.copy(), y=z.copy(),
                    depth=np.empty(0, np.float32), features=np.empty(0, np.float32))
    if method == "splatting":
        idx_2, x_pix, y_pix = visibility_from_splatting(
            x_proj, y_proj, dist, xyz[idx_1], **kw)
    elif method == "depth_map":
        idx_2, x_pix, y_pix = visibility_from_depth_map(x_proj, y_proj, dist, **kw)
    else:
        raise NotImplementedError(method)
    idx = idx_1[idx_2]
    dist = dist[idx_2]
    y_proj = y_proj[idx_2]
    g = lambda a: None if a is None else np.asarray(a)[idx]
    feats = postprocess_features(
        xyz[idx] - img_xyz, y_proj, dist, g(linearity), g(planarity),
        g(scattering), g(normals), **kw)
    return dict(idx=idx, x=x_pix, y=y_pix, depth=dist, features=feats)


# --------------------------------------------------------------------------- #
# utils/multimodal.py                                                         #
# --------------------------------------------------------------------------- #

def composite(*arrays):
    """CompositeNDArray, utils/multimodal.py:182-228 (int64 inputs)."""
    arrays = [np.asarray(a, dtype=np.int64) for a in arrays]
    if arrays[0].shape[0] == 0:
        return np.zeros(0, np.int64)
    max_list = [int(np.abs(a).max()) + 1 for a in arrays]
    assert np.prod([float(x) for x in max_list]) < np.iinfo(np.int64).max
    out = np.zeros_like(arrays[0])
    for i, a in enumerate(arrays):
        base = int(np.prod(max_list[i + 1:], dtype=object)) if i + 1 < len(arrays) else 1
        out = out + a * np.int64(base)
    return out


def lexargunique(*arrays):
    """cpu_lex_op(op='argunique'), utils/multimodal.py:308-312."""
    return np.unique(composite(*arrays), return_index=True)[1]


def lexunique(*arrays):
    """cpu_lex_op(op='unique') for a single array, utils/multimodal.py:306-307."""
    assert len(arrays) == 1
    return np.unique(np.asarray(arrays[0], dtype=np.int64))


def lexargsort(*arrays):
    """cpu_lex_op(op='argsort'), utils/multimodal.py:315-319.  The reference's
    np.argsort is unstable; the oracle uses the stable kind, i.e. the canonical
    representative of the reference's set of valid answers."""
    return np.argsort(composite(*arrays), kind="stable")


# --------------------------------------------------------------------------- #
# torch_scatter.segment_csr and the pooling helpers                           #
# --------------------------------------------------------------------------- #

def segment_csr(src, ptr, reduce="sum"):
    src = _c(src, np.float32)
    squeeze = src.ndim == 1
    src2 = src.reshape(src.shape[0], -1)
    ptr = _c(ptr, np.int64)
    nseg = ptr.shape[0] - 1
    out = np.empty((nseg, src2.shape[1]), np.float32)
    _lib().orc_segment_csr(
        _p(src2), int(src2.shape[1]), _p(ptr), C.c_longlong(nseg),
        REDUCE[reduce], _p(out))
    return out[:, 0] if squeeze else out


def gather_csr(src, ptr):
    """pooling.py:825-853."""
    ptr = np.asarray(ptr, np.int64)
    return np.repeat(src, ptr[1:] - ptr[:-1], axis=0)


def segment_gather_csr(src, ptr, reduce="sum"):
    """pooling.py:856-868."""
    return gather_csr(segment_csr(src, ptr, reduce), ptr)


def segment_softmax_csr(src, ptr, eps=1e-12, scaling=False):
    """pooling.py:770-822."""
    src = _c(src, np.float32)
    squeeze = src.ndim == 1
    src2 = src.reshape(src.shape[0], -1)
    ptr = _c(ptr, np.int64)
    out = np.empty_like(src2)
    _lib().orc_segment_softmax_csr(
        _p(src2), int(src2.shape[1]), _p(ptr), C.c_longlong(ptr.shape[0] - 1),
        C.c_float(np.float32(eps)), int(bool(scaling)), _p(out))
    return out[:, 0] if squeeze else out


def group_sizes(num_elements, num_groups):
    """pooling.py:749-757."""
    sizes = np.full(num_groups, math.floor(num_elements / num_groups), np.int64)
    sizes += np.arange(num_groups) < num_elements - sizes.sum()
    return sizes


def expand_group_feat(A, num_groups, num_channels):
    """pooling.py:760-767."""
    if num_groups == 1:
        return A.reshape(-1, 1)
    if num_groups < num_channels:
        return np.repeat(A, group_sizes(num_channels, num_groups), axis=1)
    return A


def linear(x, weight, bias=None):
    x = _c(x, np.float32)
    W = _c(weight, np.float32)
    y = np.empty((x.shape[0], W.shape[0]), np.float32)
    _lib().orc_linear_bn_act(
        _p(x), C.c_longlong(x.shape[0]), int(W.shape[1]), int(W.shape[0]), _p(W),
        _p(None if bias is None else _c(bias, np.float32)), None, None, None,
        None, C.c_float(0), C.c_float(0), _p(y))
    return y


def mlp(x, sd, prefix, n_layers=2, eps=1e-5, slope=0.2):
    """MLP(channels, bias=False) in eval mode, base_modules.py:38-48, with
    parameters taken from a state-dict-like mapping of numpy arrays under
    ``prefix`` ('<prefix>.<i>.0.weight', '<prefix>.<i>.1.batch_norm.*')."""
    x = _c(x, np.float32)
    for i in range(n_layers):
        W = _c(sd[f"{prefix}.{i}.0.weight"], np.float32)
        b = sd.get(f"{prefix}.{i}.0.bias")
        bn = f"{prefix}.{i}.1.batch_norm."
        y = np.empty((x.shape[0], W.shape[0]), np.float32)
        _lib().orc_linear_bn_act(
            _p(x), C.c_longlong(x.shape[0]), int(W.shape[1]), int(W.shape[0]),
            _p(W), _p(None if b is None else _c(b, np.float32)),
            _p(_c(sd[bn + "weight"], np.float32)), _p(_c(sd[bn + "bias"], np.float32)),
            _p(_c(sd[bn + "running_mean"], np.float32)),
            _p(_c(sd[bn + "running_var"], np.float32)),
            C.c_float(eps), C.c_float(slope), _p(y))
        x = y
    return x


def deep_set_feat(x, ptr, sd, prefix, pool=("max",), fusion="concatenation",
                  use_num=False):
    """DeepSetFeat.forward, pooling.py:658-681."""
    ptr = np.asarray(ptr, np.int64)
    x = mlp(x, sd, prefix + ".mlp_elt_1")
    x_set = np.concatenate([segment_csr(x, ptr, p) for p in pool], axis=-1)
    if use_num:
        num = np.sqrt(np.float32(1) / ((ptr[1:] - ptr[:-1]).astype(np.float32)
                                       + np.float32(1e-3))).astype(np.float32)
        x_set = np.concatenate([x_set, num.reshape(-1, 1)], axis=1)
    x_set = mlp(x_set, sd, prefix + ".mlp_set")
    x_set = gather_csr(x_set, ptr)
    if fusion == "residual":
        x_out = x + x_set
    elif fusion == "concatenation":
        x_out = np.concatenate([x, x_set], axis=-1)
    else:
        x_out = np.concatenate([x, x + x_set], axis=-1)
    return mlp(x_out, sd, prefix + ".mlp_elt_2")


def min_max_diff_set_feat(x, ptr, sd, prefix, use_min=True, use_max=True, use_num=False):
    """MinMaxDiffSetFeat.forward, pooling.py:577-597."""
    ptr = np.asarray(ptr, np.int64)
    parts = [x]
    if use_min:
        parts.append(x - segment_gather_csr(x, ptr, "min"))
    if use_max:
        parts.append(x - segment_gather_csr(x, ptr, "max"))
    if use_num:
        cnt = ptr[1:] - ptr[:-1]
        num = np.sqrt(np.float32(1) / (cnt.astype(np.float32) + np.float32(1e-3))).astype(np.float32)
        parts.append(np.repeat(num, cnt).reshape(-1, 1))
    return mlp(np.concatenate(parts, axis=1), sd, prefix + ".mlp")


def map_encoder(x, ptr, sd, prefix, encoder="DeepSetFeat", **kw):
    """The E_map module of the pools (pooling.py:240-241, 420-421)."""
    if encoder == "DeepSetFeat":
        pool = kw.get("pool", ("max",))
        pool = tuple(pool.split("_")) if isinstance(pool, str) else tuple(pool)
        return deep_set_feat(x, ptr, sd, prefix, pool, kw.get("fusion", "concatenation"),
                             kw.get("use_num", False))
    if encoder == "MinMaxDiffSetFeat":
        return min_max_diff_set_feat(x, ptr, sd, prefix, kw.get("use_min", True),
                                     kw.get("use_max", True), kw.get("use_num", False))
    if encoder == "MLPSetFeat":
        return mlp(x, sd, prefix + ".mlp")              # pooling.py:688-699
    raise ValueError(encoder)


def _gating(x, sd):
    """Gating.forward, pooling.py:717-723."""
    x = x * sd["G.weight"].astype(np.float32) + sd["G.bias"].astype(np.float32)
    return np.tanh(np.maximum(x, np.float32(0))).astype(np.float32)


def _attend(xv, comp, ptr, sd, num_groups, out_mod, gating, group_scaling):
    """pooling.py:284-300 / 514-530."""
    att = segment_softmax_csr(comp, ptr, scaling=group_scaling)
    x_pool = segment_csr(xv * expand_group_feat(att, num_groups, out_mod), ptr, "sum")
    gate = None
    if gating:
        gate = _gating(segment_csr(comp, ptr, "max"), sd)
        x_pool = x_pool * expand_group_feat(gate, num_groups, out_mod)
    return x_pool, att, gate


def qkv_pool(x_main, x_mod, x_map, ptr, sd, num_groups=1, nc_qk=8, gating=True,
             dim_scaling=True, group_scaling=False, use_num=False,
             pool=("max",), fusion="concatenation", return_parts=False,
             use_mod_q=False, use_mod_k=False, map_encoder_name="DeepSetFeat", **enc_kw):
    """QKVBimodalCSRPool.forward (eval), pooling.py:454-547."""
    ptr = np.asarray(ptr, np.int64)
    out_mod = sd["E_mod.1.0.weight"].shape[0]
    enc = dict(pool=pool, fusion=fusion, use_num=use_num)
    enc.update(enc_kw)
    xm = mlp(x_main, sd, "E_main")
    xp = map_encoder(x_map, ptr, sd, "E_map", map_encoder_name, **enc)
    xv = mlp(x_mod, sd, "E_mod")
    if use_mod_k:
        keys = linear(mlp(np.concatenate([xp, xv], axis=1), sd, "E_mix_K"),
                      sd["K.weight"], sd["K.bias"])
    else:
        keys = linear(xp, sd["K.weight"], sd["K.bias"])
    if use_mod_q:
        xq = np.concatenate([gather_csr(xm, ptr), xv], axis=1)
        queries = linear(mlp(xq, sd, "E_mix_Q"), sd["Q.weight"], sd["Q.bias"])
    else:
        queries = gather_csr(linear(xm, sd["Q.weight"], sd["Q.bias"]), ptr)
    V = keys.shape[0]
    prod = (keys.reshape(V, num_groups, nc_qk) * queries.reshape(V, num_groups, nc_qk))
    comp = prod[:, :, 0].copy()
    for k in range(1, nc_qk):
        comp = comp + prod[:, :, k]
    if dim_scaling:
        comp = (comp / np.float32(math.sqrt(nc_qk))).astype(np.float32)
    x_pool, att, gate = _attend(xv, comp, ptr, sd, num_groups, out_mod, gating, group_scaling)
    if return_parts:
        return dict(x_pool=x_pool, K=keys, Q=queries, C=comp, A=att, G=gate,
                    x_map=xp, x_mod=xv)
    return x_pool


def group_pool(x_mod, x_map, ptr, sd, num_groups=1, gating=True,
               group_scaling=True, use_num=False, pool=("max",),
               fusion="concatenation", use_mod=False, map_encoder_name="DeepSetFeat",
               return_parts=False, **enc_kw):
    """GroupBimodalCSRPool.forward (eval), pooling.py:263-315."""
    ptr = np.asarray(ptr, np.int64)
    out_mod = sd["E_mod.1.0.weight"].shape[0]
    enc = dict(pool=pool, fusion=fusion, use_num=use_num)
    enc.update(enc_kw)
    xp = map_encoder(x_map, ptr, sd, "E_map", map_encoder_name, **enc)
    xv = mlp(x_mod, sd, "E_mod")
    if use_mod:
        xp = mlp(np.concatenate([xp, xv], axis=1), sd, "E_mix")
    comp = linear(xp, sd["E_score.weight"], sd["E_score.bias"])
    x_pool, att, gate = _attend(xv, comp, ptr, sd, num_groups, out_mod, gating, group_scaling)
    if return_parts:
        return dict(x_pool=x_pool, C=comp, A=att, G=gate)
    return x_pool


# --------------------------------------------------------------------------- #
# csr.py / core/multimodal/image.py                                           #
# --------------------------------------------------------------------------- #

def sorted_indices_to_pointers(indices):
    """CSRData._sorted_indices_to_pointers, csr.py:158-172."""
    indices = np.asarray(indices, np.int64)
    return np.concatenate([
        [0], np.where(indices[1:] > indices[:-1])[0] + 1, [indices.shape[0]]]).astype(np.int64)


def insert_empty_groups(pointers, group_indices, num_groups=None):
    """CSRData.insert_empty_groups, csr.py:197-229."""
    group_indices = np.asarray(group_indices, np.int64)
    n = int(group_indices.max()) + 1
    if num_groups is not None:
        n = max(n, int(num_groups))
    starts = np.concatenate([[-1], group_indices])
    ends = np.concatenate([group_indices, [n]])
    return np.repeat(pointers, ends - starts)


def mapping_from_dense(point_ids, image_ids, pixels, features, num_points=None):
    """ImageMapping.from_dense, core/multimodal/image.py:1738-1805.  Returns a
    dict: pointers [N+1], images [Vw], view_pointers [Vw+1], pixels [P,2],
    features [Vw,F].  Pixels of one (point, image) group are put in (x, y)
    order: the reference's unstable argsort leaves that order unspecified."""
    point_ids = np.asarray(point_ids, np.int64)
    image_ids = np.asarray(image_ids, np.int64)
    pixels = np.asarray(pixels)
    order = np.lexsort((pixels[:, 1], pixels[:, 0], image_ids, point_ids))
    point_ids, image_ids, pixels = point_ids[order], image_ids[order], pixels[order]
    comp = composite(point_ids, image_ids)
    view_ptr = sorted_indices_to_pointers(comp)
    last = view_ptr[1:] - 1
    images = image_ids[last]
    pids = point_ids[last]
    feats = None
    if features is not None:
        feats = segment_csr(np.asarray(features, np.float32)[order], view_ptr, "mean")
    ptr = sorted_indices_to_pointers(pids)
    if num_points is None or num_points < pids.max() + 1:
        num_points = int(pids.max()) + 1
    ptr = insert_empty_groups(ptr, pids[ptr[1:] - 1], num_groups=num_points)
    return dict(pointers=ptr.astype(np.int64), images=images,
                view_pointers=view_ptr, pixels=pixels, features=feats)


# --------------------------------------------------------------------------- #
# core/data_transform/multimodal/image.py :: MapImages._process               #
# --------------------------------------------------------------------------- #

def select_points(m, idx, mode="pick"):
    """ImageMapping.select_points, core/multimodal/image.py:2177-2287, on a
    mapping dict as returned by mapping_from_dense (with features)."""
    idx = np.asarray(idx, np.int64)
    ptr, vptr = m["pointers"], m["view_pointers"]
    if mode == "pick":
        # CSRData.__getitem__, csr.py:235-294
        views = np.concatenate([np.arange(ptr[i], ptr[i + 1]) for i in idx] + [np.zeros(0, np.int64)]).astype(np.int64)
        px = np.concatenate([np.arange(vptr[v], vptr[v + 1]) for v in views] + [np.zeros(0, np.int64)]).astype(np.int64)
        return dict(
            pointers=np.concatenate([[0], np.cumsum(ptr[idx + 1] - ptr[idx])]).astype(np.int64),
            images=m["images"][views],
            view_pointers=np.concatenate([[0], np.cumsum(vptr[views + 1] - vptr[views])]).astype(np.int64),
            pixels=m["pixels"][px], features=m["features"][views])
    assert mode == "merge" and idx.shape[0] == ptr.shape[0] - 1
    point_ids = np.repeat(idx, np.diff(ptr))
    image_ids = m["images"]
    key = composite(point_ids, image_ids)
    feats = m["features"]
    mean = np.zeros_like(feats)
    for k in np.unique(key):                       # scatter_mean + gather, rows in input order
        rows = np.where(key == k)[0]
        acc = feats[rows[0]].copy()
        for r in rows[1:]:
            acc = acc + feats[r]
        mean[rows] = acc / np.float32(len(rows))
    sizes = np.diff(vptr)
    point_ids, image_ids, mean = np.repeat(point_ids, sizes), np.repeat(image_ids, sizes), np.repeat(mean, sizes, axis=0)
    pix = m["pixels"]
    u = lexargunique(point_ids, image_ids, pix[:, 0], pix[:, 1])
    return mapping_from_dense(point_ids[u], image_ids[u], pix[u], mean[u], num_points=int(idx.max()) + 1)


def downscale_pixels(pixels, ratio):
    """ImageMapping.downscale_images, core/multimodal/image.py:1926-1990: the
    de-duplication key starts with the (distinct) atomic row id, so every row
    survives and only the coordinates change."""
    pixels = np.asarray(pixels)
    if ratio == 1:
        return pixels.copy()
    return np.floor(pixels / np.float32(ratio)).astype(np.int64).astype(pixels.dtype)


def upscale_pixels(pixels, ratio, center=True):
    """ImageMapping.upscale_images, core/multimodal/image.py:1992-2037."""
    pixels = np.asarray(pixels)
    if ratio == 1:
        return pixels.copy()
    up = pixels.astype(np.float32) * np.float32(ratio)
    if center:
        up = up + np.float32(ratio / 2)
    return up.astype(np.int64).astype(pixels.dtype)


def feature_map_batch(images, view_ptr):
    """first entry of ImageMapping.feature_map_indexing, image.py:1890-1891"""
    return np.repeat(np.asarray(images, np.int64), np.diff(np.asarray(view_ptr, np.int64)))


def gather_features(x, images, view_ptr, pixels):
    """X[mappings.feature_map_indexing], core/multimodal/image.py:1295,1881-1895."""
    pixels = np.asarray(pixels).astype(np.int64)
    return x[feature_map_batch(images, view_ptr), :, pixels[:, 1], pixels[:, 0]]


def sparse_interpolation(features, coords, batch, padding_mode="border"):
    """sparse_interpolation, core/multimodal/image.py:105-170 (float32)."""
    mode = {"zeros": "constant", "border": "edge", "reflection": "reflect"}[padding_mode]
    pad = np.pad(features, ((0, 0), (0, 0), (1, 1), (1, 1)), mode=mode)
    h, w = features.shape[2:]
    f32 = np.float32
    pix = np.asarray(coords, f32) * np.array([[h, w]], f32) + f32(0.5)
    top, bottom = np.floor(pix[:, 0]), np.floor(pix[:, 0] + f32(1))
    left, right = np.floor(pix[:, 1]), np.floor(pix[:, 1] + f32(1))
    corner = lambda a, b: np.stack([a, b], 1).astype(np.int64)
    tl, tr, bl, br = corner(top, left), corner(top, right), corner(bottom, left), corner(bottom, right)
    wgt = lambda c: np.abs(np.prod(pix - c.astype(f32), axis=1, dtype=f32))[:, None]
    at = lambda c: pad[batch, :, c[:, 0], c[:, 1]]
    out = wgt(br) * at(tl) + wgt(bl) * at(tr)
    out = out + wgt(tr) * at(bl)
    out = out + wgt(tl) * at(br)
    return out.astype(f32)


def pixel_dtype(ref_size):
    """SameSettingImageData.pixel_dtype, core/multimodal/image.py:507-516."""
    for dt in (np.int16, np.int32, np.int64):
        if np.iinfo(dt).max >= max(ref_size):
            return dt


def map_images(
        pos, mapping_index, cams, order=None, method="splatting",
        ref_size=None, proj_upscale=1, downscale=1, crop_size=None,
        crop_offsets=None, img_mask=None, cylinder=False, linearity=None,
        planarity=None, scattering=None, normals=None, depth_maps=None,
        timings=None, workers=1, **vis_kw):
    """MapImages._process, core/data_transform/multimodal/image.py:192-428.

    ``cams``: dict with pos [V,3] float64 and, per camera model, extrinsic
    [V,4,4] + fx, fy, mx, my [V] (pinhole), extrinsic + fisheye [V,7]
    (xi, k1, k2, gamma1, gamma2, u0, v0) or opk [V,3] (equirectangular).
    ``order``: permutation of the points in the traversal order of the
    reference's KDTree(leaf_size=50) -- ``tree.idx_array`` -- which fixes the
    local point order of every per-image sample (transforms.py:332,386); None
    means ascending index.  Returns (mapping dict as mapping_from_dense, plus
    'seen_images' = indices of the views that kept at least one mapping).
    """
    import time
    pos = np.asarray(pos)
    N = pos.shape[0]
    V = cams["pos"].shape[0]
    order = np.arange(N) if order is None else np.asarray(order, np.int64)
    proj_size = tuple(int(x * proj_upscale) for x in ref_size)
    crop_size = tuple(ref_size) if crop_size is None else tuple(crop_size)
    crop_offsets = np.zeros((V, 2), np.int64) if crop_offsets is None \
        else np.asarray(crop_offsets, np.int64)
    r_max = vis_kw["r_max"]
    pos64 = pos.astype(np.float64)
    pos32 = pos.astype(np.float32)
    pdt = pixel_dtype(ref_size)
    T = dict(sample=0.0, visibility=0.0, pixels=0.0, unique=0.0, dense=0.0)

    def one_view(v):
        t0 = time.perf_counter()
        # radius sampling: sklearn KDTree.query_radius keeps points whose
        # float64 distance to the centre is <= r, returned in tree order
        c = cams["pos"][v].astype(np.float64)
        if cylinder:
            d2 = (pos64[order, 0] - c[0]) ** 2 + (pos64[order, 1] - c[1]) ** 2
        else:
            d2 = (pos64[order, 0] - c[0]) ** 2 + (pos64[order, 1] - c[1]) ** 2 \
                + (pos64[order, 2] - c[2]) ** 2
        local = order[d2 <= float(r_max) * float(r_max)]
        g = lambda a: None if a is None else np.asarray(a)[local]
        t1 = time.perf_counter()
        K = None
        if cams.get("fx") is not None:
            K = np.eye(4, dtype=np.float32)
            K[0, 0], K[1, 1] = cams["fx"][v], cams["fy"][v]
            K[0, 2], K[1, 2] = cams["mx"][v], cams["my"][v]
        kw = dict(vis_kw)
        # image.py:276-279: opk / fisheye intrinsics / extrinsic when the images hold them
        if cams.get("opk") is not None:
            kw["img_opk"] = np.asarray(cams["opk"][v], np.float32)
        if cams.get("fisheye") is not None:
            kw["img_intrinsic_fisheye"] = np.asarray(cams["fisheye"][v], np.float32)
        if depth_maps is not None:
            kw["depth_map"] = depth_maps[v]
        out = visibility_model(
            pos32[local], cams["pos"][v].astype(np.float32), method=method,
            linearity=g(linearity), planarity=g(planarity),
            scattering=g(scattering), normals=g(normals),
            img_intrinsic_pinhole=K,
            img_extrinsic=None if cams.get("extrinsic") is None else cams["extrinsic"][v],
            img_mask=img_mask, img_size=proj_size, **kw)
        t2 = time.perf_counter()
        if out["idx"].shape[0] == 0:
            return None, (t1 - t0, t2 - t1, 0.0, 0.0)
        pid = np.asarray(mapping_index, np.int64)[local][out["idx"]]
        px = np.asarray(out["x"]).astype(np.int64)
        py = np.asarray(out["y"]).astype(np.int64)
        ft = out["features"]
        px = px // proj_upscale - crop_offsets[v, 0]
        py = py // proj_upscale - crop_offsets[v, 1]
        keep = (px >= 0) & (py >= 0) & (px < crop_size[0]) & (py < crop_size[1])
        px, py, pid, ft = px[keep], py[keep], pid[keep], ft[keep]
        px = (px // downscale).astype(np.int64)
        py = (py // downscale).astype(np.int64)
        t3 = time.perf_counter()
        u = lexargunique(pid, px, py)
        px, py, pid, ft = px[u], py[u], pid[u], ft[u]
        t4 = time.perf_counter()
        return (pid, ft, np.stack([px, py], 1).astype(pdt)), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)

    # views are independent (the reference loops over them, image.py:238); with
    # workers > 1 they run on a thread pool (the C loops release the GIL)
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as pool:
            results = list(pool.map(one_view, range(V)))
    else:
        results = [one_view(v) for v in range(V)]
    image_ids, point_ids, features, pixels = [], [], [], []
    for v, (res, dt) in enumerate(results):
        for k, t in zip(("sample", "visibility", "pixels", "unique"), dt):
            T[k] += t
        if res is None:
            continue
        image_ids.append(v)
        point_ids.append(res[0])
        features.append(res[1])
        pixels.append(res[2])

    if len(image_ids) == 0:
        raise ValueError("No mappings were found between the 3D points and any "
                         "of the provided images.")
    t5 = time.perf_counter()
    seen = lexunique(np.asarray(image_ids, np.int64))
    ranks = np.searchsorted(seen, np.asarray(image_ids, np.int64))
    img = np.repeat(ranks, [p.shape[0] for p in point_ids])
    out = mapping_from_dense(
        np.concatenate(point_ids), img, np.concatenate(pixels),
        np.concatenate(features),
        num_points=int(np.asarray(mapping_index).max()) + 1)
    T["dense"] += time.perf_counter() - t5
    out["seen_images"] = seen
    if timings is not None:
        timings.update(T)
    return out


# --------------------------------------------------------------------------- #
# data_transform/multimodal/image.py: NeighborhoodBasedMappingFeatures          #
# --------------------------------------------------------------------------- #

def knn(xyz, k):
    """Exact k nearest neighbours of every point among the points (self first):
    float64 squared distances of the float32 coordinates, equal distances by
    ascending index.  The reference searches with FAISS or KeOps
    (image.py:497-518); brute force here, for test-sized clouds."""
    p = np.asarray(xyz, np.float32).astype(np.float64)
    n = p.shape[0]
    out = np.empty((n, k), np.int64)
    for a in range(0, n, 1024):
        d = p[a:a + 1024, None, :] - p[None, :, :]
        d2 = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
        out[a:a + 1024] = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return out


def neighborhood_features(xyz, neighbors, pointers, images, k_list, voxel=1, density=True,
                          occlusion=True):
    """NeighborhoodBasedMappingFeatures._process, image.py:520-610: the columns
    appended to the view-level mapping features."""
    f32 = np.float32
    xyz = np.asarray(xyz, f32)
    pointers = np.asarray(pointers, np.int64)
    images = np.asarray(images, np.int64)
    sizes = np.diff(pointers)
    cols = []
    if density:
        dens = []
        for k in k_list:
            d = xyz - xyz[neighbors[:, k - 1]]
            d2_max = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            v_sphere = f32(3.1416) * d2_max
            with np.errstate(divide="ignore", invalid="ignore"):
                de = (f32(k + 1) / v_sphere) / f32(1 / voxel ** 2)
            de[np.isnan(de)] = 1
            dens.append(de.astype(f32).reshape(-1, 1))
        cols.append(np.repeat(np.concatenate(dens, 1), sizes, axis=0))
    if occlusion:
        n_points = xyz.shape[0]
        n_images = int(images.max()) + 1
        point_ids = np.repeat(np.arange(n_points), sizes)
        views = np.zeros((n_points, n_images), bool)
        views[point_ids, images] = True
        occ = []
        for k in k_list:
            seen = np.ones(images.shape[0], f32)
            for i in range(k):
                seen += views[neighbors[point_ids, i], images]
            occ.append((seen / f32(k + 1)).reshape(-1, 1))
        cols.append(np.concatenate(occ, 1))
    return np.concatenate(cols, 1).astype(f32)
