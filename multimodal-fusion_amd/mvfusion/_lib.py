"""ctypes binding of libmvfusion.so (the HIP kernels behind a C ABI).

There is no CPU fallback: if the shared library is missing, or no HIP device
is visible, the operators raise.  The library is built in-tree by
``multimodal-fusion_amd/csrc/Makefile`` (``__graft_entry__.build()``).
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVFUSION_SO: an instrumented build of the same library (profiling scripts only)
SO_PATH = os.environ.get("MVFUSION_SO") or os.path.join(_HERE, "libmvfusion.so")

CAMERAS = {"kitti360_perspective": 0, "scannet": 1, "kitti360_fisheye": 2,
           "s3dis_equirectangular": 3}
REDUCE = {"sum": 0, "add": 0, "mean": 1, "min": 2, "max": 3}


class MvfView(C.Structure):
    _fields_ = [("img_xyz", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("cam_to_world", C.c_float * 16),
                ("centre", C.c_double * 3),
                ("crop_off_x", C.c_int32), ("crop_off_y", C.c_int32)]


class MvfVisParams(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32),
                ("crop_top", C.c_int32), ("crop_bottom", C.c_int32),
                ("camera", C.c_int32), ("exact", C.c_int32),
                ("r_max", C.c_double), ("r_min", C.c_double),
                ("voxel", C.c_double), ("k_swell", C.c_double), ("ln_d_swell", C.c_double),
                ("depth_threshold", C.c_float), ("pad_", C.c_int32)]


assert C.sizeof(MvfView) == 128


class MvfMlpLayer(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p),
                ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p),
                ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p),
                ("bn_eps", C.c_float), ("slope", C.c_float),
                ("cin", C.c_int32), ("cout", C.c_int32), ("act", C.c_int32),
                ("pad_", C.c_int32)]


assert C.sizeof(MvfMlpLayer) == 72

_lock = threading.Lock()
_lib = None

_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_F = C.c_float
_D = C.c_double
_SZ = C.c_size_t

_SIGNATURES = {
    "mvf_version": (C.c_int, []),
    "mvf_last_error": (C.c_char_p, []),
    "mvf_profile_enable": (C.c_int, [C.c_int]),
    "mvf_profile_read": (C.c_int, [_P, _P, C.c_int]),
    "mvf_camera_projection_workspace_bytes": (_SZ, [_I64]),
    "mvf_camera_projection": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "mvf_camera_projection_noalloc_max_points": (_I64, []),
    "mvf_camera_projection_noalloc": (C.c_int, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mvf_splat_visibility_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I64]),
    "mvf_splat_visibility": (C.c_int, [_P, _P, _P, _I64, _F, _F, _P, _I64, _P, _P, _P, _I64,
                                       _P, _P, _SZ, _P]),
    "mvf_splat_visibility_view": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P, _P, _I64,
                                            _P, _P, _SZ, _P]),
    "mvf_depth_map_visibility_workspace_bytes": (_SZ, [_I64]),
    "mvf_depth_map_visibility": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _I32, _F, _P, _P, _P,
                                           _P, _P, _SZ, _P]),
    "mvf_mapping_features": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _I32, _D, _D,
                                       _P, _P]),
    "mvf_map_images_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32, _I64, _I64, _I64]),
    "mvf_map_images_exact": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _I32, _P, _I32, _P, _P, _P,
                                       _P, _P, _I32, _I32, _I32, _I32, _I64, _I64,
                                       _P, _P, _P, _P, _I64, _P, _P, _P, _SZ, _P]),
    "mvf_map_images_batch_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32, _I32, _I64, _I64, _I64]),
    "mvf_map_images_exact_batch": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _I32, _I32, _P, _I32, _P,
                                             _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I64, _I64,
                                             _P, _P, _P, _P, _I64, _P, _P, _P, _SZ, _P]),
    "mvf_map_images_splat_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32, _I64, _I64, _I64]),
    "mvf_map_images_splat": (C.c_int, [_P, _P, _I64, _I64, _P, _I32, _P, _I32, _P, _P, _P, _P, _P,
                                       _I32, _I32, _I64, _I64, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P,
                                       _SZ, _P]),
    "mvf_profile_zbuffer_ops": (C.c_int, [_P, _SZ, _I64, _I32, _P, _I64, _I64, _I64, _P, _P]),
    "mvf_map_images_depth_workspace_bytes": (_SZ, [_I64, _I32]),
    "mvf_map_images_depth": (C.c_int, [_P, _P, _I64, _I64, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P,
                                       _I32, _I32, _I32, _I32, _P, _P, _P, _P, _I64, _P, _P, _P, _SZ, _P]),
    "mvf_mapping_from_dense_workspace_bytes": (_SZ, [_I64, _I64]),
    "mvf_mapping_from_dense": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I64, _I64, _I64, _I32, _P, _P, _P,
                                         _P, _P, _P, _P, _SZ, _P]),
    "mvf_group_mean": (C.c_int, [_P, _P, _I32, _I64, _I64, _P, _P, _SZ, _P]),
    "mvf_image_stats": (C.c_int, [_P, _P, _P, _I32, _I64, _I64, _P, _P]),
    "mvf_segment_csr": (C.c_int, [_P, _I64, _I32, _P, _I64, _I32, _P, _P]),
    "mvf_segment_csr_backward": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _I64, _I32, _P, _P]),
    "mvf_segment_softmax_csr_backward": (C.c_int, [_P, _P, _I32, _P, _I64, _I32, _P, _P]),
    "mvf_batch_norm_act_workspace_bytes": (_SZ, [_I32]),
    "mvf_batch_norm_act_forward": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _I32, _F, _F, _I32, _F,
                                             _P, _P, _P, _P, _I32, _P, _SZ, _P]),
    "mvf_batch_norm_act_backward": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P, _I32, _I32, _F,
                                              _P, _P, _P, _P, _SZ, _P]),
    "mvf_xty_workspace_bytes": (_SZ, []),
    "mvf_xty": (C.c_int, [_P, _I32, _I32, _P, _I32, _I32, _I64, _P, _I32, _P, _SZ, _P]),
    "mvf_weighted_segment_sum": (C.c_int, [_P, _P, _I32, _I32, _P, _I64, _P, _P]),
    "mvf_weighted_segment_sum_backward": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _I64, _I64, _P, _P, _P]),
    "mvf_linear_rows_supported": (C.c_int, [_I32, _I32]),
    "mvf_linear_rows": (C.c_int, [_P, _I32, _P, _P, _I32, _I64, _P, _P]),
    "mvf_linear_rows_sums_bytes": (_SZ, [_I32]),
    "mvf_linear_rows_sums": (C.c_int, [_P, _I32, _P, _P, _I32, _I64, _P, _P, _SZ, _P, _P]),
    "mvf_kdtree_order": (C.c_int, [_P, _I64, _I32, _I64, _I64, _P, _I32]),
    "mvf_knn": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P]),
    "mvf_mapping_occlusion": (C.c_int, [_P, _I32, _I32, _P, _P, _I64, _P, _I64, _P, _I32, _P]),
    "mvf_gather_csr": (C.c_int, [_P, _I32, _P, _I64, _I64, _P, _P]),
    "mvf_segment_softmax_csr": (C.c_int, [_P, _I32, _P, _I64, _F, _I32, _P, _P]),
    "mvf_segment_csr_two_level": (C.c_int, [_P, _I64, _I32, _P, _I64, _I32, _P, _I64, _I32, _P, _I32, _P, _P]),
    "mvf_gather_pool": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _I64, _I32, _P, _P]),
    "mvf_gather_features": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _I64, _P, _P]),
    "mvf_sparse_interpolation": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _I64, _I32, _P, _P]),
    "mvf_mlp_forward": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _I32, _P, _I32, _I64, _P, _I32,
                                  _P, _I32, _P]),
    "mvf_mlp_forward_dot": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _I32, _P, _I32, _I64, _P, _I32,
                                      _P, _I32, _P, _F, _P, _P]),
    "mvf_mlp_forward_pool_workspace_bytes": (_SZ, [_I64, _I32]),
    "mvf_mlp_forward_pool": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _I32, _P, _I64, _P, _I32,
                                       _P, _I32, _P, _P, _I64, _P, _P, _SZ, _P]),
    "mvf_mlp_forward_segmax": (C.c_int, [_P, _I32, _I32, _P, _I32, _I32, _I64, _P, _I32, _P, _I32,
                                         _P, _I64, _P, _P, _P]),
    "mvf_attention_pool": (C.c_int, [_P, _I32, _P, _I32, _P, _P, _I32, _P, _I64, _I64, _I32,
                                     _I32, _I32, _I32, _F, _P, _P, _P, _I32, _P, _P, _P, _P, _P]),
    "mvf_set_mlp_precision": (C.c_int, [_I32]),
    "mvf_get_mlp_precision": (C.c_int, []),
    "mvf_csr_row_segments": (C.c_int, [_P, _I64, _P, _P]),
    "mvf_segment_pick": (C.c_int, [_P, _I32, _P, _I32, _P, _I64, _I32, _P, _P, _P]),
    "mvf_gather_features_backward": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _I64, _P, _P]),
    "mvf_sparse_interpolation_backward": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _I64, _I32, _P, _P]),
}

EXPORTS = tuple(_SIGNATURES)


def load():
    """Load libmvfusion.so once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(SO_PATH):
                raise RuntimeError(
                    f"mvfusion: {SO_PATH} is missing -- build it with "
                    f"`make -C multimodal-fusion_amd/csrc` (there is no CPU fallback).")
            lib = C.CDLL(SO_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


class MvfError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise MvfError(f"mvfusion error {rc}: {load().mvf_last_error().decode()}")


def require_device(t=None):
    """Device on which a call runs: the tensor's own HIP device, else the
    current one.  Raises when no HIP device is visible."""
    if t is not None and t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("mvfusion: no HIP device is visible (there is no CPU fallback).")
    return torch.device("cuda", torch.cuda.current_device())


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
