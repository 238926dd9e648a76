"""Host-side mirror of ``torch_points3d/core/multimodal/visibility.py``: same
function / class names, arguments and return values, with the computation done
by the HIP kernels of libmvfusion.so.  Cameras: 'kitti360_perspective',
'scannet', 'kitti360_fisheye', 's3dis_equirectangular'.

Reference lines (relative to DeepViewAgg-Wildscenes/torch_points3d/):
``camera_projection`` core/multimodal/visibility.py:592-623 (CPU body :478-538),
``visibility_from_splatting`` :1288-1321 (:1073-1195),
``visibility_from_depth_map`` :1361-1388, ``postprocess_features`` :1548-1600,
``visibility`` :1607-1692, ``VisibilityModel`` & co :1695-1805.

Results follow the reference's CPU (numba) path, which its authors designate
as the reliable one: ascending-index outputs, lower index wins a depth tie,
``exact`` keeps one point per centre pixel.  Tensors may live on the CPU or on
a HIP device; CPU inputs are staged to the current HIP device and results are
returned on the input's device.  There is no CPU implementation.
"""
import ctypes as C
import math
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import _lib as L

_PINHOLE = ("kitti360_perspective", "scannet")

# defaults of the reference when an argument is None (visibility.py:13-16)
_OPK_DEFAULT = (0.0, 0.0, 0.0)
_FISHEYE_DEFAULT = (1.0,) * 7


def pose_to_rotation_matrix(opk):
    """pose_to_rotation_matrix_cpu (visibility.py:57-90): float32 rotation from
    an omega, phi, kappa triplet."""
    opk = np.asarray(opk, dtype=np.float32).reshape(3)
    co, so = np.cos(opk[0]), np.sin(opk[0])
    cp, sp = np.cos(opk[1]), np.sin(opk[1])
    ck, sk = np.cos(opk[2]), np.sin(opk[2])
    M_o = np.array([[1.0, 0.0, 0.0], [0.0, co, -so], [0.0, so, co]], dtype=np.float32)
    M_p = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]], dtype=np.float32)
    M_k = np.array([[ck, -sk, 0.0], [sk, ck, 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    return np.dot(M_o, np.dot(M_p, M_k))


def _dev(t, device, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(device).contiguous()


def _np32(t):
    return np.asarray(torch.as_tensor(t).detach().cpu().float())


def make_view(img_xyz, img_intrinsic_pinhole, img_extrinsic, camera, centre=None,
              crop_offsets=(0, 0), img_opk=None, img_intrinsic_fisheye=None):
    """Pack one view into the C struct (host side)."""
    v = L.MvfView()
    xyz = np.asarray(torch.as_tensor(img_xyz).detach().cpu().float()).reshape(3)
    if camera in ("kitti360_fisheye", "s3dis_equirectangular"):
        for i in range(3):
            v.img_xyz[i] = xyz[i]
        if camera == "kitti360_fisheye":
            E = np.eye(4, dtype=np.float32) if img_extrinsic is None else _np32(img_extrinsic)
            f = np.asarray(_FISHEYE_DEFAULT, np.float32) if img_intrinsic_fisheye is None \
                else _np32(img_intrinsic_fisheye).reshape(7)
            M = np.ascontiguousarray(E, dtype=np.float32).reshape(16).copy()
            M[12:15] = f[0:3]                      # xi, k1, k2 in the unused last row
            v.fx, v.fy, v.cx, v.cy = float(f[3]), float(f[4]), float(f[5]), float(f[6])
        else:
            R = pose_to_rotation_matrix(_OPK_DEFAULT if img_opk is None else _np32(img_opk))
            M = np.zeros(16, np.float32)
            M[:9] = R.reshape(9)
        for i, e in enumerate(M):
            v.cam_to_world[i] = e
        c = xyz.astype(np.float64) if centre is None else np.asarray(centre, np.float64).reshape(3)
        for i in range(3):
            v.centre[i] = c[i]
        v.crop_off_x, v.crop_off_y = int(crop_offsets[0]), int(crop_offsets[1])
        return v
    K = np.eye(4, dtype=np.float32) if img_intrinsic_pinhole is None else \
        np.asarray(torch.as_tensor(img_intrinsic_pinhole).detach().cpu().float())
    E = np.eye(4, dtype=np.float32) if img_extrinsic is None else \
        np.asarray(torch.as_tensor(img_extrinsic).detach().cpu().float())
    if camera == "scannet":
        # the reference inverts the float32 matrix (visibility.py:233)
        E = np.linalg.inv(np.ascontiguousarray(E))
    for i in range(3):
        v.img_xyz[i] = xyz[i]
    v.fx, v.fy, v.cx, v.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    for i, e in enumerate(np.ascontiguousarray(E, dtype=np.float32).reshape(16)):
        v.cam_to_world[i] = e
    c = xyz.astype(np.float64) if centre is None else np.asarray(centre, np.float64).reshape(3)
    for i in range(3):
        v.centre[i] = c[i]
    v.crop_off_x, v.crop_off_y = int(crop_offsets[0]), int(crop_offsets[1])
    return v


def make_params(img_size=(1024, 512), crop_top=0, crop_bottom=0, r_max=30, r_min=0.5,
                camera="kitti360_perspective", voxel=0.1, k_swell=1.0, d_swell=1000,
                exact=False, depth_threshold=0.05, **kwargs):
    if camera not in L.CAMERAS:
        raise ValueError(f"Unknown camera '{camera}'")
    p = L.MvfVisParams()
    p.W, p.H = int(img_size[0]), int(img_size[1])
    p.crop_top, p.crop_bottom = int(crop_top), int(crop_bottom)
    p.camera = L.CAMERAS[camera]
    p.exact = int(bool(exact))
    p.r_max, p.r_min = float(r_max), float(r_min)
    p.voxel, p.k_swell = float(voxel), float(k_swell)
    p.ln_d_swell = math.log(d_swell)
    p.depth_threshold = float(depth_threshold)
    return p


def camera_projection(
        xyz, img_xyz, img_opk=None, img_intrinsic_pinhole=None,
        img_intrinsic_fisheye=None, img_extrinsic=None, img_mask=None,
        img_size=(1024, 512), crop_top=0, crop_bottom=0, r_max=30, r_min=0.5,
        camera='s3dis_equirectangular', **kwargs):
    assert img_mask is None or tuple(img_mask.shape) == tuple(img_size), \
        f'Expected img_mask to be a torch.BoolTensor of shape ' \
        f'img_size={img_size} but got size={img_mask.shape}.'
    in_device = xyz.device
    device = L.require_device(xyz)
    lib = L.load()
    params = make_params(img_size=img_size, crop_top=crop_top, crop_bottom=crop_bottom,
                         r_max=r_max, r_min=r_min, camera=camera)
    view = make_view(img_xyz, img_intrinsic_pinhole, img_extrinsic, camera,
                     img_opk=img_opk, img_intrinsic_fisheye=img_intrinsic_fisheye)
    xyz_d = _dev(xyz, device, torch.float32)
    n = xyz_d.shape[0]
    mask_d = _dev(img_mask, device, torch.uint8) if img_mask is not None else None
    indices = torch.empty(n, dtype=torch.int64, device=device)
    dist = torch.empty(n, dtype=torch.float32, device=device)
    x = torch.empty(n, dtype=torch.float64, device=device)
    y = torch.empty(n, dtype=torch.float64, device=device)
    counters = torch.zeros(4, dtype=torch.int64, device=device)
    ws_bytes = lib.mvf_camera_projection_workspace_bytes(n)
    ws = L.workspace(ws_bytes, device)
    with torch.cuda.device(device):
        L.check(lib.mvf_camera_projection(
            L.ptr(xyz_d), n, C.byref(view), C.byref(params), L.ptr(mask_d),
            L.ptr(indices), L.ptr(dist), L.ptr(x), L.ptr(y), L.ptr(counters),
            L.ptr(ws), ws.numel(), L.stream_ptr(device)))
    m = int(counters[0].item())
    out = (indices[:m], dist[:m], x[:m], y[:m])
    return tuple(o.to(in_device) for o in out)


# grown on demand: (splat, tile) pairs per splat, remembered across calls
_pair_factor = [4.0]


def visibility_from_splatting(
        x_proj, y_proj, dist, xyz, img_extrinsic=None, img_intrinsic_pinhole=None,
        img_intrinsic_fisheye=None, img_size=(1024, 512),
        crop_top=0, crop_bottom=0, voxel=0.1, k_swell=1.0, d_swell=1000,
        exact=False, camera='s3dis_equirectangular', **kwargs):
    assert x_proj.shape[0] == y_proj.shape[0] == dist.shape[0] > 0
    in_device = x_proj.device
    device = L.require_device(x_proj)
    lib = L.load()
    params = make_params(img_size=img_size, crop_top=crop_top, crop_bottom=crop_bottom,
                         camera=camera, voxel=voxel, k_swell=k_swell, d_swell=d_swell,
                         exact=exact)
    view = make_view(torch.zeros(3), img_intrinsic_pinhole, img_extrinsic, camera,
                     img_intrinsic_fisheye=img_intrinsic_fisheye)
    xyz_d = None
    if camera == 'kitti360_fisheye':
        # fisheye_splat_cpu projects the lifted points again (visibility.py:900-914)
        assert xyz is not None and xyz.shape[0] == x_proj.shape[0]
        xyz_d = _dev(xyz, device, torch.float32)
    x_d = _dev(x_proj, device, torch.float64)
    y_d = _dev(y_proj, device, torch.float64)
    d_d = _dev(dist, device, torch.float32)
    m = x_d.shape[0]
    W, H = int(img_size[0]), int(img_size[1])
    Hc = H - crop_top - crop_bottom
    out_cap = m if exact else W * Hc
    idx = torch.empty(out_cap, dtype=torch.int64, device=device)
    xo = torch.empty(out_cap, dtype=torch.int64, device=device)
    yo = torch.empty(out_cap, dtype=torch.int64, device=device)
    counters = torch.zeros(4, dtype=torch.int64, device=device)
    pair_cap = int(m * _pair_factor[0]) + 1024
    for _ in range(3):
        ws_bytes = lib.mvf_splat_visibility_workspace_bytes(m, W, H, pair_cap)
        ws = L.workspace(ws_bytes, device)
        with torch.cuda.device(device):
            L.check(lib.mvf_splat_visibility_view(
                L.ptr(x_d), L.ptr(y_d), L.ptr(d_d), L.ptr(xyz_d), m, C.byref(view),
                C.byref(params), pair_cap,
                L.ptr(idx), L.ptr(xo), L.ptr(yo), out_cap, L.ptr(counters),
                L.ptr(ws), ws.numel(), L.stream_ptr(device)))
        k, pairs, overflow, _ = counters.tolist()
        if not overflow:
            break
        pair_cap = pairs + 1024
        _pair_factor[0] = max(_pair_factor[0], 1.25 * pairs / m)
    else:
        raise L.MvfError("visibility_from_splatting: buffers still too small after retries")
    return idx[:k].to(in_device), xo[:k].to(in_device), yo[:k].to(in_device)


def read_s3dis_depth_map(path, img_size=None, empty=-1):
    """Decode an S3DIS-format 16-bit depth PNG into a [W, H] float32 map in
    metres (visibility.py:1328-1358)."""
    from PIL import Image
    im = Image.open(path)
    if img_size is not None:
        im = im.resize(img_size, resample=Image.NEAREST)
    raw = np.array(im).T
    out = torch.from_numpy(raw.astype(np.float32) / np.float32(512))
    out[torch.from_numpy(raw == 2 ** 16 - 1)] = empty
    return out


def render_depth_maps(xyz, img_pos, img_intrinsic_pinhole, img_extrinsic, img_size, r_max=30,
                      r_min=0.5, voxel=0.1, k_swell=1.0, d_swell=1000,
                      camera='kitti360_perspective', quantum=1.0 / 512, img_opk=None):
    """Depth maps [V, W, H] float32 of a cloud seen from V views (pinhole, or
    equirectangular with ``img_opk`` [V, 3] and no intrinsics / extrinsics), in the
    format ``read_s3dis_depth_map`` returns (visibility.py:1328-1358): the
    distance of the nearest splat per pixel (non-exact splat visibility of this
    library), rounded to ``quantum`` metres as the 16-bit PNGs are, -1 where no
    splat lands.  For synthetic inputs of the depth-map visibility model."""
    device = L.require_device(xyz)
    xyz_d = _dev(xyz, device, torch.float32)
    W, H = int(img_size[0]), int(img_size[1])
    V = img_pos.shape[0]
    out = torch.full((V, W, H), -1.0, dtype=torch.float32, device=device)
    for v in range(V):
        cam_kw = dict(
            img_intrinsic_pinhole=None if img_intrinsic_pinhole is None else img_intrinsic_pinhole[v],
            img_extrinsic=None if img_extrinsic is None else img_extrinsic[v],
            img_opk=None if img_opk is None else img_opk[v])
        idx, dist, x, y = camera_projection(
            xyz_d, img_pos[v].float(), img_size=(W, H), r_max=r_max, r_min=r_min, camera=camera, **cam_kw)
        if idx.shape[0] == 0:
            continue
        vi, px, py = visibility_from_splatting(
            x, y, dist, None, img_size=(W, H), voxel=voxel, k_swell=k_swell, d_swell=d_swell,
            exact=False, camera=camera, **cam_kw)
        d = torch.round(dist[vi].double() / quantum) * quantum
        out[v, px, py] = d.float()
    return out


def visibility_from_depth_map(
        x_proj, y_proj, dist, depth_map_path=None, img_size=(1024, 512),
        depth_threshold=0.05, depth_map=None, **kwargs):
    """``depth_map`` (a decoded [W, H] float32 tensor) may be given instead of
    ``depth_map_path``; it spares the PNG decode on the host."""
    assert x_proj.shape[0] == y_proj.shape[0] == dist.shape[0] > 0
    if depth_map is None:
        assert depth_map_path is not None, f'Please provide depth_map_path.'
        depth_map = read_s3dis_depth_map(depth_map_path, img_size=img_size, empty=-1)
    in_device = x_proj.device
    device = L.require_device(x_proj)
    lib = L.load()
    W, H = int(img_size[0]), int(img_size[1])
    assert tuple(depth_map.shape) == (W, H)
    x_d = _dev(x_proj, device, torch.float64)
    y_d = _dev(y_proj, device, torch.float64)
    d_d = _dev(dist, device, torch.float32)
    dm_d = _dev(depth_map, device, torch.float32)
    m = x_d.shape[0]
    idx = torch.empty(m, dtype=torch.int64, device=device)
    xo = torch.empty(m, dtype=torch.float64, device=device)
    yo = torch.empty(m, dtype=torch.float64, device=device)
    counters = torch.zeros(4, dtype=torch.int64, device=device)
    ws = L.workspace(lib.mvf_depth_map_visibility_workspace_bytes(m), device)
    with torch.cuda.device(device):
        L.check(lib.mvf_depth_map_visibility(
            L.ptr(x_d), L.ptr(y_d), L.ptr(d_d), m, L.ptr(dm_d), W, H,
            float(np.float32(depth_threshold)), L.ptr(idx), L.ptr(xo), L.ptr(yo),
            L.ptr(counters), L.ptr(ws), ws.numel(), L.stream_ptr(device)))
    k = int(counters[0].item())
    # the reference returns x_proj[indices] in the dtype of x_proj
    return idx[:k].to(in_device), xo[:k].to(in_device).to(x_proj.dtype), \
        yo[:k].to(in_device).to(y_proj.dtype)


def postprocess_features(
        xyz_to_img, y_proj, dist, linearity, planarity, scattering, normals,
        img_size=(1024, 512), r_max=30, r_min=0.5, **kwargs):
    """Mapping features of already-gathered per-mapping inputs (same argument
    meaning as the reference: ``xyz_to_img = xyz[idx] - img_xyz``)."""
    in_device = dist.device
    device = L.require_device(dist)
    lib = L.load()
    k = dist.shape[0]
    f32 = lambda t: _dev(t, device, torch.float32)
    attrs = [f32(linearity), f32(planarity), f32(scattering), f32(normals)]
    F = 2 + sum(a is not None for a in attrs)
    out = torch.empty((k, F), dtype=torch.float32, device=device)
    idx = torch.arange(k, dtype=torch.int64, device=device)
    zero = (C.c_float * 3)(0.0, 0.0, 0.0)
    with torch.cuda.device(device):
        L.check(lib.mvf_mapping_features(
            L.ptr(f32(xyz_to_img)), L.ptr(idx), L.ptr(_dev(y_proj, device, torch.float64)),
            L.ptr(f32(dist)), k, zero, *[L.ptr(a) for a in attrs], int(img_size[1]),
            float(r_max), float(r_min), L.ptr(out), L.stream_ptr(device)))
    return out.to(in_device)


def _empty_out(in_device):
    out = {}
    out['idx'] = torch.empty((0,), dtype=torch.long, device=in_device)
    out['x'] = torch.empty((0,), dtype=torch.long, device=in_device)
    out['y'] = torch.empty((0,), dtype=torch.long, device=in_device)
    out['depth'] = torch.empty((0,), dtype=torch.float, device=in_device)
    out['features'] = torch.empty((0,), dtype=torch.float, device=in_device)
    return out


def _run_visibility(xyz, img_xyz, method, linearity, planarity, scattering, normals, kw):
    """Shared body of VisibilityModel.__call__ and visibility(): everything
    stays on the HIP device between the steps."""
    in_device = xyz.device
    device = L.require_device(xyz)
    xyz_d = _dev(xyz, device, torch.float32)
    img_xyz_d = torch.as_tensor(img_xyz).float()
    idx_1, dist, x_proj, y_proj = camera_projection(xyz_d, img_xyz_d, **kw)
    if x_proj.shape[0] == 0:
        return _empty_out(in_device)
    if method == 'splatting':
        xyz_1 = xyz_d[idx_1] if kw.get('camera') == 'kitti360_fisheye' else None
        idx_2, x_pix, y_pix = visibility_from_splatting(x_proj, y_proj, dist, xyz_1, **kw)
    elif method == 'depth_map':
        idx_2, x_pix, y_pix = visibility_from_depth_map(x_proj, y_proj, dist, **kw)
    else:
        raise NotImplementedError(method)
    idx = idx_1[idx_2]
    dist = dist[idx_2]
    y_proj = y_proj[idx_2]
    lib = L.load()
    f32 = lambda t: _dev(t, device, torch.float32)
    attrs = [f32(linearity), f32(planarity), f32(scattering), f32(normals)]
    F = 2 + sum(a is not None for a in attrs)
    k = idx.shape[0]
    feats = torch.empty((k, F), dtype=torch.float32, device=device)
    c = np.asarray(img_xyz_d.cpu(), dtype=np.float32).reshape(3)
    h_img = (C.c_float * 3)(*[float(v) for v in c])
    with torch.cuda.device(device):
        L.check(lib.mvf_mapping_features(
            L.ptr(xyz_d), L.ptr(idx.contiguous()), L.ptr(y_proj.contiguous()),
            L.ptr(dist.contiguous()), k, h_img, *[L.ptr(a) for a in attrs],
            int(kw.get('img_size', (1024, 512))[1]), float(kw.get('r_max', 30)),
            float(kw.get('r_min', 0.5)), L.ptr(feats), L.stream_ptr(device)))
    out = {'idx': idx, 'x': x_pix, 'y': y_pix, 'depth': dist, 'features': feats}
    return {k_: v.to(in_device) for k_, v in out.items()}


def visibility(
        xyz, img_xyz, method='splatting', linearity=None, planarity=None,
        scattering=None, normals=None, use_cuda=True, **kwargs):
    METHODS = ['splatting', 'depth_map', 'biasutti']
    assert method in METHODS, \
        f'Unknown method {method}, expected one of {METHODS}.'
    return _run_visibility(
        xyz, img_xyz, method, linearity, planarity, scattering, normals, kwargs)


class VisibilityModel(ABC):

    def __init__(
            self, img_size=(1024, 512), crop_top=0, crop_bottom=0, r_max=30,
            r_min=0.5, camera='s3dis_equirectangular'):
        self.img_size = img_size
        self.crop_top = crop_top
        self.crop_bottom = crop_bottom
        self.r_max = r_max
        self.r_min = r_min
        self.camera = camera

    _method = None

    def _camera_projection(self, *args, **kwargs):
        return camera_projection(*args, **self.__dict__, **kwargs)

    @abstractmethod
    def _visibility(self):
        pass

    def _postprocess_features(self, *args):
        return postprocess_features(*args, **self.__dict__)

    def __call__(
            self, xyz, img_xyz, linearity=None, planarity=None, scattering=None,
            normals=None, **kwargs):
        return _run_visibility(
            xyz, img_xyz, self._method, linearity, planarity, scattering, normals,
            {**self.__dict__, **kwargs})

    def __repr__(self):
        attr_repr = ', '.join([f'{k}={v}' for k, v in self.__dict__.items()])
        return f'{self.__class__.__name__}({attr_repr})'


class SplattingVisibility(VisibilityModel, ABC):
    _method = 'splatting'

    def __init__(
            self, voxel=0.1, k_swell=1.0, d_swell=1000, exact=False, **kwargs):
        super(SplattingVisibility, self).__init__(**kwargs)
        self.voxel = voxel
        self.k_swell = k_swell
        self.d_swell = d_swell
        self.exact = exact

    def _visibility(self, x_proj, y_proj, dist, xyz, **kwargs):
        return visibility_from_splatting(
            x_proj, y_proj, dist, xyz, **self.__dict__, **kwargs)


class DepthBasedVisibility(VisibilityModel, ABC):
    _method = 'depth_map'

    def __init__(self, depth_threshold=0.05, **kwargs):
        super(DepthBasedVisibility, self).__init__(**kwargs)
        self.depth_threshold = depth_threshold

    def _visibility(self, x_proj, y_proj, dist, xyz, **kwargs):
        return visibility_from_depth_map(
            x_proj, y_proj, dist, **self.__dict__, **kwargs)
