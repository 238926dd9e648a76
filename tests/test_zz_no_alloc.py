"""mvf_camera_projection_noalloc (the entry point that allocates nothing: its
compaction scratch is a device array inside the library) against
mvf_camera_projection, through the C ABI: two calls in a row, the second on a
smaller input, plus the size limit."""
import ctypes as C

import numpy as np
import pytest
import torch

from mvfusion import _lib as L
from mvfusion import synthetic as S
from mvfusion import visibility as V

pytestmark = pytest.mark.gpu


def _project(lib, xyz_d, view, params, noalloc):
    n = xyz_d.shape[0]
    device = xyz_d.device
    indices = torch.full((max(n, 1),), -7, dtype=torch.int64, device=device)
    dist = torch.zeros(max(n, 1), dtype=torch.float32, device=device)
    x = torch.zeros(max(n, 1), dtype=torch.float64, device=device)
    y = torch.zeros(max(n, 1), dtype=torch.float64, device=device)
    counters = torch.full((4,), -1, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        if noalloc:
            L.check(lib.mvf_camera_projection_noalloc(
                L.ptr(xyz_d), n, C.byref(view), C.byref(params), None, L.ptr(indices),
                L.ptr(dist), L.ptr(x), L.ptr(y), L.ptr(counters), L.stream_ptr(device)))
        else:
            ws = L.workspace(lib.mvf_camera_projection_workspace_bytes(n), device)
            L.check(lib.mvf_camera_projection(
                L.ptr(xyz_d), n, C.byref(view), C.byref(params), None, L.ptr(indices),
                L.ptr(dist), L.ptr(x), L.ptr(y), L.ptr(counters), L.ptr(ws), ws.numel(),
                L.stream_ptr(device)))
    torch.cuda.synchronize(device)
    m = int(counters[0].item())
    return m, indices[:m].cpu().numpy(), dist[:m].cpu().numpy(), x[:m].cpu().numpy(), y[:m].cpu().numpy()


def test_noalloc_projection_equals_workspace_projection():
    lib = L.load()
    device = torch.device("cuda", 0)
    sc = S.make_scene("c2", seed=11, n_points=200_000, n_views=2)
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = sc["fx"][0], sc["fy"][0], sc["mx"][0], sc["my"][0]
    params = V.make_params(img_size=sc["img_size"], r_max=20, r_min=0, camera="kitti360_perspective")
    xyz = torch.from_numpy(sc["xyz"]).to(device)
    # two calls in a row on the same GPU: the second is smaller and must not see
    # anything the first left in the shared scratch
    for view_id, n in ((0, 200_000), (1, 37_123)):
        view = V.make_view(torch.from_numpy(sc["pos"][view_id]).float(), torch.from_numpy(K),
                           torch.from_numpy(sc["extrinsic"][view_id]), "kitti360_perspective")
        part = xyz[:n].contiguous()
        want = _project(lib, part, view, params, noalloc=False)
        got = _project(lib, part, view, params, noalloc=True)
        assert want[0] > 1000
        assert got[0] == want[0]
        for g, w in zip(got[1:], want[1:]):
            assert g.dtype == w.dtype
            assert np.array_equal(g.view(np.uint8), w.view(np.uint8))       # bit for bit


def test_noalloc_projection_size_limit_and_empty():
    lib = L.load()
    device = torch.device("cuda", 0)
    limit = lib.mvf_camera_projection_noalloc_max_points()
    assert limit == 1 << 24
    params = V.make_params(img_size=(64, 64), camera="kitti360_perspective")
    view = V.make_view(torch.zeros(3), None, None, "kitti360_perspective")
    dummy = torch.zeros(16, dtype=torch.float64, device=device)
    counters = torch.full((4,), 5, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        # one point too many: rejected before anything is launched (no buffer is touched)
        rc = lib.mvf_camera_projection_noalloc(
            L.ptr(dummy), limit + 1, C.byref(view), C.byref(params), None, L.ptr(dummy),
            L.ptr(dummy), L.ptr(dummy), L.ptr(dummy), L.ptr(counters), L.stream_ptr(device))
        assert rc == -1 and b"exceeds" in lib.mvf_last_error()
        L.check(lib.mvf_camera_projection_noalloc(
            None, 0, C.byref(view), C.byref(params), None, None, None, None, None,
            L.ptr(counters), L.stream_ptr(device)))
    torch.cuda.synchronize(device)
    assert int(counters[0].item()) == 0
