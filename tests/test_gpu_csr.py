"""GPU parity of the CSR containers next to the hot path: lexicographic ops,
ImageMapping.from_dense, CSRBatch / ImageMappingBatch, against the golden
vectors captured from the reference (integer outputs bit-exact)."""
import numpy as np
import pytest
import torch

from mvfusion import mapping as M
from mvfusion.csr import CSRData

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
N = lambda t: t.detach().cpu().numpy()


def canonical_pixels(view_ptr, pixels):
    # the reference leaves the pixel order inside a (point, image) group to an
    # unstable argsort: compare groups as sorted sets
    out = pixels.copy()
    for a, b in zip(view_ptr[:-1], view_ptr[1:]):
        if b - a > 1:
            seg = pixels[a:b]
            out[a:b] = seg[np.lexsort((seg[:, 1], seg[:, 0]))]
    return out


def test_lexops_golden(golden):
    g = golden("lexops")
    a, b, c = T(g["a"]), T(g["b"]), T(g["c"])
    assert np.array_equal(N(M.lexargunique(a, b, c)), g["argunique"])
    assert np.array_equal(N(M.lexunique(a)), g["unique_a"])
    assert M.lexargunique(a[:0], b[:0]).shape[0] == 0
    mine = N(M.lexargsort(a, b))
    for arr in (g["a"], g["b"]):
        assert np.array_equal(arr[mine], arr[g["argsort"]])


def test_from_dense_golden(golden):
    g = golden("from_dense")
    m = M.ImageMapping.from_dense(
        T(g["point_ids"]), T(g["image_ids"]), T(g["pixels"]), T(g["features"]),
        num_points=int(g["num_points"]))
    assert np.array_equal(N(m.pointers), g["pointers"])
    assert np.array_equal(N(m.images), g["images"])
    assert np.array_equal(N(m.atomic_csr_indexing), g["view_pointers"])
    assert m.pixels.dtype == torch.int16
    assert np.array_equal(canonical_pixels(g["view_pointers"], N(m.pixels)),
                          canonical_pixels(g["view_pointers"], g["out_pixels"]))
    np.testing.assert_allclose(N(m.features), g["out_features"], rtol=1e-6, atol=1e-7)
    # the index tuple for x[B, C, H, W] (core/multimodal/image.py:1881-1895)
    ib, _, ih, iw = m.feature_map_indexing
    sizes = np.diff(g["view_pointers"])
    assert np.array_equal(N(ib), np.repeat(g["images"], sizes))
    assert np.array_equal(N(iw), N(m.pixels)[:, 0]) and np.array_equal(N(ih), N(m.pixels)[:, 1])


def test_csr_batch_golden(golden):
    g = golden("csr_batch")
    items = [M.ImageMapping.from_dense(
        T(g[f"pid{k}"]), T(g[f"iid{k}"]), T(g[f"pix{k}"]), T(g[f"feat{k}"]),
        num_points=int(g["num_points"])) for k in range(3)]
    batch = M.ImageMappingBatch.from_csr_list(items)
    assert np.array_equal(N(batch.pointers), g["pointers"])
    assert np.array_equal(N(batch.images), g["images"])
    assert np.array_equal(N(batch.atomic_csr_indexing), g["view_pointers"])
    assert np.array_equal(canonical_pixels(g["view_pointers"], N(batch.pixels)),
                          canonical_pixels(g["view_pointers"], g["pixels"]))
    np.testing.assert_allclose(N(batch.features), g["features"], rtol=1e-6, atol=1e-7)
    # round trip (core/multimodal/image.py:2360-2400)
    back = batch.to_csr_list()
    assert len(back) == 3
    for a, b in zip(items, back):
        assert torch.equal(a.pointers, b.pointers) and torch.equal(a.images, b.images)
        assert torch.equal(a.pixels, b.pixels)


def test_csr_docstring_case():
    # csr.py:26-41: dense indices -> pointers, empty groups inserted
    idx = torch.tensor([0, 0, 1, 3, 3, 3], device=DEV)
    val = torch.arange(6, device=DEV)
    c = CSRData(idx, val, dense=True)
    assert N(c.pointers).tolist() == [0, 2, 3, 6]
    c = c.insert_empty_groups(torch.tensor([0, 1, 3], device=DEV), num_groups=5)
    assert N(c.pointers).tolist() == [0, 2, 3, 3, 6, 6]


@pytest.mark.parametrize("mode", ["max", "mean", "min", "sum"])
def test_gather_atomic_pool_vs_indexing(golden, mode):
    """Fused feature-map gather + atomic pool == x[feature_map_indexing] followed by
    the CPU oracle's segment_csr (max / min bit-exact, sums to 1e-6)."""
    import mvf_oracle as orc
    from mvfusion.pooling import gather_atomic_pool
    g = golden("from_dense")
    m = M.ImageMapping.from_dense(
        T(g["point_ids"]), T(g["image_ids"]), T(g["pixels"]), T(g["features"]),
        num_points=int(g["num_points"]))
    B = int(m.images.max()) + 1
    Hh, Ww = int(m.pixels[:, 1].max()) + 1, int(m.pixels[:, 0].max()) + 1
    rng = np.random.default_rng(3)
    for C_ in (13, 70):
        x = rng.random((B, C_, Hh, Ww)).astype(np.float32)
        got = N(gather_atomic_pool(T(x), m, mode))
        ib, _, ih, iw = [N(t) if torch.is_tensor(t) else t for t in m.feature_map_indexing]
        dense = x[ib, :, ih, iw]
        want = orc.segment_csr(dense, N(m.atomic_csr_indexing), mode)
        if mode in ("max", "min"):
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mode", ["max", "min"])
def test_segment_csr_long_segments(mode):
    """max / min over long segments (a wave per segment instead of a thread per
    output value): bit-equal to the oracle's in-order fold for 1, 13, 64 and 96
    channels, with empty and one-row segments in between, values quantised so
    that ties are frequent, and signed zeros (a tie between +0.0 and -0.0 goes
    to the first row); then the two-level pool over such inner segments."""
    import mvf_oracle as orc
    from mvfusion.pooling import segment_csr, segment_csr_two_level
    rng = np.random.default_rng(17)
    nseg = 3000
    sizes = rng.integers(20, 200, nseg)
    sizes[rng.random(nseg) < 0.05] = 0
    sizes[rng.random(nseg) < 0.05] = 1
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    assert ptr[-1] >= 16 * nseg
    for C_ in (1, 13, 64, 96):
        x = np.round(rng.normal(size=(int(ptr[-1]), C_)) * 4).astype(np.float32) / 4
        x[rng.random(x.shape) < 0.3] = 0.0
        x[rng.random(x.shape) < 0.1] = -0.0
        got = N(segment_csr(T(x), T(ptr), reduce=mode))
        want = orc.segment_csr(x, ptr, mode)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), C_
    outer_sizes = rng.integers(0, 5, 900)
    optr = np.concatenate([[0], np.cumsum(outer_sizes)]).astype(np.int64)
    n_inner = int(optr[-1])
    x = rng.random((int(ptr[n_inner]), 13)).astype(np.float32) - 0.5
    got = segment_csr_two_level(T(x), T(ptr[:n_inner + 1]), T(optr), inner=mode, outer=("max", "mean"))
    mid = orc.segment_csr(x, ptr[:n_inner + 1], mode)
    assert np.array_equal(N(got[0]), orc.segment_csr(mid, optr, "max"))
    np.testing.assert_allclose(N(got[1]), orc.segment_csr(mid, optr, "mean"), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mode", ["max", "min"])
def test_segment_csr_long_segments_special_values(mode):
    """The wave-per-segment kernel folds with the hardware's max / min and repeats a segment
    with the exact first-row-wins procedure when a row holds a NaN, an infinity, or a zero wins:
    segments of infinities only, mixed infinities and numbers, zeros of both signs that win
    (all other values on the losing side), a mostly quiet cloud with a few such segments, and
    segment lengths around the batch of rows a wave has in flight -- bit for bit the in-order
    fold."""
    from mvfusion.pooling import segment_csr
    rng = np.random.default_rng(31)
    sizes = np.concatenate([rng.integers(17, 200, 600), [63, 64, 65, 127, 128, 129, 256, 1000]])
    nseg = sizes.shape[0]
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    seg_of = np.repeat(np.arange(nseg), sizes)
    lose = -1.0 if mode == "max" else 1.0
    for C_ in (1, 13, 32, 64):
        x = rng.normal(size=(int(ptr[-1]), C_)).astype(np.float32)
        kind = rng.integers(0, 8, nseg)                      # per segment; 4..7: left alone
        rows = lambda k: np.isin(seg_of, np.where(kind == k)[0])
        x[rows(0)] = np.where(rng.random((int(rows(0).sum()), C_)) < 0.5, np.inf, -np.inf)
        x[rows(1) & (rng.random(x.shape[0]) < 0.1)] = -np.inf if mode == "max" else np.inf
        x[rows(2) & (rng.random(x.shape[0]) < 0.1)] = np.inf
        z = rows(3)
        x[z] = lose * np.abs(x[z])                           # losers ...
        zero = rng.random(x.shape) < 0.2                     # ... and zeros of both signs that win
        zero[~z] = False
        x[zero] = np.where(rng.random(int(zero.sum())) < 0.5, 0.0, -0.0)
        want = np.empty((nseg, C_), np.float32)
        for s_ in range(nseg):
            acc = x[ptr[s_]].copy()
            for r in range(ptr[s_] + 1, ptr[s_ + 1]):
                take = x[r] > acc if mode == "max" else x[r] < acc
                acc = np.where(take, x[r], acc)
            want[s_] = acc
        got = N(segment_csr(T(x), T(ptr), reduce=mode))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), C_
        assert (np.signbit(want[kind == 3]) & (want[kind == 3] == 0)).any()      # -0.0 results are in


@pytest.mark.parametrize("mode", ["max", "min"])
def test_segment_csr_nan_rows_long_equals_short_kernel(mode):
    """Rows that hold a NaN: the in-order fold of the reference's segment reduction keeps a
    NaN only when it is the first row of a segment (every comparison with it is false), and
    passes it over anywhere else.  The wave-per-segment kernel (long segments) returns the
    same bits as that fold, computed here row by row in numpy and by the thread-per-value
    kernel (the same segments padded with empty ones, so that the average length is short)."""
    from mvfusion.pooling import segment_csr
    rng = np.random.default_rng(23)
    nseg = 800
    sizes = rng.integers(17, 150, nseg)
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    for C_ in (1, 13, 64):
        x = rng.normal(size=(int(ptr[-1]), C_)).astype(np.float32)
        x[rng.random(x.shape) < 0.05] = np.nan
        first = ptr[:-1]
        x[first[rng.random(nseg) < 0.3]] = np.nan             # whole first rows
        x[first[::7] + 1] = np.nan                            # and second rows
        want = np.empty((nseg, C_), np.float32)
        for s_ in range(nseg):
            acc = x[ptr[s_]].copy()
            for r in range(ptr[s_] + 1, ptr[s_ + 1]):
                take = x[r] > acc if mode == "max" else x[r] < acc
                acc = np.where(take, x[r], acc)
            want[s_] = acc
        got_long = N(segment_csr(T(x), T(ptr), reduce=mode))
        assert np.array_equal(got_long.view(np.uint32), want.view(np.uint32)), C_
        # the same segments among 20 empty ones each: the serial kernel
        sparse = np.concatenate([np.repeat(ptr[:-1], 20), ptr[-1:]]).astype(np.int64)
        got_short = N(segment_csr(T(x), T(sparse), reduce=mode))[19::20]
        assert np.array_equal(got_short.view(np.uint32), want.view(np.uint32)), C_


@pytest.mark.parametrize("inner", ["max", "mean", "min", "sum"])
def test_segment_csr_two_level_equals_two_passes(inner):
    """One-pass two-level pool == two segment_csr calls, bit for bit, and == the
    CPU oracle (max / min exact, sums to 1e-6); empty inner and outer segments,
    13 and 64 channels."""
    import mvf_oracle as orc
    from mvfusion.pooling import segment_csr, segment_csr_two_level
    rng = np.random.default_rng(7)
    n_outer = 5000
    outer_sizes = rng.integers(0, 6, n_outer)
    n_inner = int(outer_sizes.sum())
    inner_sizes = rng.integers(0, 4, n_inner)
    optr = np.concatenate([[0], np.cumsum(outer_sizes)]).astype(np.int64)
    iptr = np.concatenate([[0], np.cumsum(inner_sizes)]).astype(np.int64)
    for C_ in (13, 64):
        x = (rng.random((int(iptr[-1]), C_)) - 0.5).astype(np.float32)
        mid = segment_csr(T(x), T(iptr), reduce=inner)
        for outer in (("max", "mean"), ("min", "sum"), ("mean",)):
            got = segment_csr_two_level(T(x), T(iptr), T(optr), inner=inner, outer=outer)
            assert len(got) == len(outer)
            for g_, o_ in zip(got, outer):
                assert torch.equal(g_, segment_csr(mid, T(optr), reduce=o_))
                want = orc.segment_csr(orc.segment_csr(x, iptr, inner), optr, o_)
                if inner in ("max", "min") and o_ in ("max", "min"):
                    assert np.array_equal(N(g_), want)
                else:
                    np.testing.assert_allclose(N(g_), want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mode", ["pick", "merge"])
def test_select_points_golden(golden, mode):
    """ImageMapping.select_points (core/multimodal/image.py:2177-2287) against
    the reference's output: CSR arrays bit-exact, features to 1e-6."""
    g = golden("select_points")
    m = M.ImageMapping.from_dense(
        T(g["point_ids"]), T(g["image_ids"]), T(g["pixels"]), T(g["features"]),
        num_points=int(g["num_points"]))
    out = m.select_points(T(g[mode]), mode=mode)
    assert isinstance(out, M.ImageMapping)
    assert np.array_equal(N(out.pointers), g[mode + "_pointers"])
    assert np.array_equal(N(out.images), g[mode + "_images"])
    vp = g[mode + "_view_pointers"]
    assert np.array_equal(N(out.atomic_csr_indexing), vp)
    assert np.array_equal(canonical_pixels(vp, N(out.pixels)), canonical_pixels(vp, g[mode + "_pixels"]))
    np.testing.assert_allclose(N(out.features), g[mode + "_features"], rtol=1e-6, atol=1e-7)
    # degenerate inputs return an unchanged copy / empty groups (image.py:2208-2215)
    same = m.select_points(T(np.zeros(0, np.int64)), mode=mode)
    assert torch.equal(same.pointers, m.pointers)
    if mode == "merge":
        bad = m.select_points(T(g["merge"][:-1]), mode="merge")      # wrong length: unchanged
        assert torch.equal(bad.pointers, m.pointers)


# ----------------------------------------------------------- mapped features
def _mapping_from_fixture(g):
    return M.ImageMapping.from_csr_arrays(
        T(g["pointers"]).cuda(), T(g["images"]).cuda(), T(g["pixels"]).cuda(),
        T(g["features"]).cuda(), view_pointers=T(g["view_pointers"]).cuda())


def test_mapped_features_golden(golden):
    """get_mapped_features and its parts against the reference's outputs:
    gathers bit-exact, bilinear interpolation to 1e-6 relative."""
    from mvfusion import pooling as P
    from mvfusion.images import SameSettingImageData
    g = golden("mapped_features")
    m = _mapping_from_fixture(g)
    x1, x2 = T(g["x1"]).cuda(), T(g["x2"]).cuda()
    assert np.array_equal(P.gather_features(x1, m).cpu().numpy(), g["gather_scale1"])
    down = m.rescale_images(1 / 2)
    assert down.pixels.dtype == torch.int16
    assert np.array_equal(down.pixels.cpu().numpy(), g["down_pixels"])
    assert np.array_equal(down.values[1].pointers.cpu().numpy(), g["down_view_pointers"])
    assert np.array_equal(P.gather_features(x2, down).cpu().numpy(), g["gather_scale_half"])
    for center in (True, False):
        up = m.upscale_images(2, center=center)
        assert np.array_equal(up.pixels.cpu().numpy(), g[f"up_pixels_center{int(center)}"])
    # the source mapping is left as it was
    assert np.array_equal(m.pixels.cpu().numpy(), g["pixels"])
    coords, batch = T(g["coords"]).cuda(), T(g["batch"]).cuda()
    for mode in ("border", "zeros", "reflection"):
        got = P.sparse_interpolation(x2, coords, batch, padding_mode=mode).cpu().numpy()
        np.testing.assert_allclose(got, g["interp_" + mode], rtol=1e-6, atol=1e-7)
    with pytest.raises(NotImplementedError):
        P.sparse_interpolation(x2, coords, batch, padding_mode="nope")
    # the container method: maps at the mapping resolution, at half of it
    # (mappings rescaled) and at half of it with interpolation
    n_img = g["x1"].shape[0]
    Wm, Hm = (int(v) for v in g["mapping_size"])
    for x, downscale, interpolate, want in (
            (x1, 1, False, g["gather_scale1"]), (x2, 2, False, g["gather_scale_half"]),
            (x2, 2, True, g["interp_border"])):
        images = SameSettingImageData(pos=torch.zeros(n_img, 3), ref_size=(Wm, Hm),
                                      proj_upscale=1, downscale=downscale, x=x, mappings=m)
        got = images.get_mapped_features(interpolate=interpolate).cpu().numpy()
        if interpolate:
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
        else:
            assert np.array_equal(got, want)


def test_gather_features_and_interpolation_vs_oracle_large():
    """C=64 feature maps, 300k mapped pixels: the gather is a pure copy
    (bit-exact), the interpolation follows the oracle's float32 order."""
    import mvf_oracle as orc
    from mvfusion import pooling as P
    rng = np.random.default_rng(41)
    B, C, H, W, n_views = 6, 64, 188, 352, 60000
    sizes = rng.integers(1, 10, n_views)
    vptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(vptr[-1])
    images = rng.integers(0, B, n_views).astype(np.int64)
    pixels = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.int16)
    x = rng.random((B, C, H, W)).astype(np.float32)
    ptr = np.arange(n_views + 1, dtype=np.int64)
    m = M.ImageMapping.from_csr_arrays(
        T(ptr).cuda(), T(images).cuda(), T(pixels).cuda(), None, view_pointers=T(vptr).cuda())
    got = P.gather_features(T(x).cuda(), m).cpu().numpy()
    assert np.array_equal(got, orc.gather_features(x, images, vptr, pixels))
    batch = orc.feature_map_batch(images, vptr)
    coords = rng.random((n, 2)).astype(np.float32)
    coords[:50] = 0.0
    coords[50:100] = 1.0
    for mode in ("border", "zeros", "reflection"):
        got = P.sparse_interpolation(T(x).cuda(), T(coords).cuda(), T(batch).cuda(), mode)
        want = orc.sparse_interpolation(x, coords, batch, mode)
        assert np.array_equal(got.cpu().numpy(), want), mode
    with pytest.raises(IndexError):
        bad = M.ImageMapping.from_csr_arrays(
            T(ptr).cuda(), T(images + B).cuda(), T(pixels).cuda(), None,
            view_pointers=T(vptr).cuda())
        P.gather_features(T(x).cuda(), bad)


# ------------------------------------------------------------------ backward
@pytest.mark.parametrize("mode", ["sum", "mean", "max", "min"])
def test_bimodal_pool_backward(mode):
    """Gradient of BimodalCSRPool / segment_csr with respect to the pooled
    features against torch.segment_reduce's autograd on the CPU (empty,
    singleton and long segments; no ties in the random data), plus the
    first-occurrence rule for equal maxima."""
    from mvfusion.pooling import BimodalCSRPool, segment_csr
    rng = np.random.default_rng(51)
    sizes = rng.poisson(4, 3000)
    sizes[rng.random(3000) < 0.15] = 0
    sizes[7] = 900
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows, C = int(ptr[-1]), 13
    x = rng.normal(size=(rows, C)).astype(np.float32)
    w = rng.normal(size=(len(sizes), C)).astype(np.float32)

    xg = T(x).requires_grad_(True)
    pool = BimodalCSRPool(mode=mode)
    out = pool(None, xg, None, T(ptr))
    (out * T(w)).sum().backward()

    xc = torch.from_numpy(x).requires_grad_(True)
    ref = torch.segment_reduce(xc, mode, offsets=torch.from_numpy(ptr), axis=0, initial=0.0) \
        if mode in ("sum", "mean") else \
        torch.segment_reduce(xc, mode, offsets=torch.from_numpy(ptr), axis=0)
    ref = torch.where(torch.from_numpy(sizes > 0)[:, None], ref, torch.zeros_like(ref))
    (ref * torch.from_numpy(w)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-6, atol=1e-7)

    if mode in ("max", "min"):
        # equal values: the first row of the segment takes the whole gradient
        t = T(np.array([[1.0], [3.0], [3.0], [2.0], [2.0]], np.float32)).requires_grad_(True)
        sign = 1.0 if mode == "max" else -1.0
        o = segment_csr(t * sign, T(np.array([0, 3, 5], np.int64)), reduce=mode)
        o.sum().backward()
        want = [0, 1, 0, 1, 0] if mode == "max" else [0, 1, 0, 1, 0]
        assert (t.grad.view(-1).cpu() * sign).tolist() == [float(v) for v in want]
    # no graph is recorded when gradients are off or not needed
    with torch.no_grad():
        assert not segment_csr(xg, T(ptr), reduce=mode).requires_grad
    assert not segment_csr(T(x), T(ptr), reduce=mode).requires_grad


def test_images_select_points_merge_golden(golden):
    """SameSettingImageData.select_points(mode='merge') (the re-indexing after a
    strided 3D convolution) on device tensors against the reference."""
    from mvfusion.images import SameSettingImageData
    g = golden("image_selection")
    n_img, size = g["pos_img"].shape[0], int(g["size"])
    m = M.ImageMapping.from_csr_arrays(
        T(g["in_pointers"]), T(g["in_images"]), T(g["in_pixels"]), T(g["in_features"]),
        view_pointers=T(g["in_view_pointers"]))
    images = SameSettingImageData(pos=T(g["pos_img"]), ref_size=(size, size), proj_upscale=1,
                                  mappings=m)
    out = images.select_points(T(g["merge"]), mode="merge")
    assert out.num_views == n_img
    mo = out.mappings
    assert np.array_equal(mo.pointers.cpu().numpy(), g["merge_pointers"])
    assert np.array_equal(mo.images.cpu().numpy(), g["merge_images"])
    vp = g["merge_view_pointers"]
    assert np.array_equal(mo.values[1].pointers.cpu().numpy(), vp)
    assert np.array_equal(mo.pixels.cpu().numpy(), canonical_pixels(vp, g["merge_pixels"]))
    np.testing.assert_allclose(mo.features.cpu().numpy(), g["merge_features"], rtol=1e-6, atol=1e-7)
    # 'pick' on the device gives what the host-side test checks on the CPU
    pick = images.select_points(T(g["pick"]), mode="pick")
    assert np.array_equal(pick.mappings.pointers.cpu().numpy(), g["pick_pointers"])
    assert np.array_equal(pick.pos.cpu().numpy(), g["pick_pos"])


def test_cropping_golden(golden):
    """ImageMapping.crop / SameSettingImageData.update_cropping and the
    downscale bookkeeping when feature maps of another resolution are set."""
    from mvfusion.images import SameSettingImageData
    g = golden("cropping")
    W, H = (int(v) for v in g["size"])
    n_img = g["x"].shape[0]
    m = M.ImageMapping.from_csr_arrays(
        T(g["in_pointers"]), T(g["in_images"]), T(g["in_pixels"]), T(g["in_features"]),
        view_pointers=T(g["in_view_pointers"]))
    images = SameSettingImageData(pos=torch.zeros(n_img, 3, device=DEV), ref_size=(W, H),
                                  proj_upscale=1, downscale=1, mappings=m, x=T(g["x"]))
    images.update_cropping(tuple(int(v) for v in g["crop_size"]), T(g["crop_offsets"]))
    mo = images.mappings
    assert np.array_equal(mo.pointers.cpu().numpy(), g["crop_pointers"])
    assert np.array_equal(mo.images.cpu().numpy(), g["crop_images"])
    vp = g["crop_view_pointers"]
    assert np.array_equal(mo.values[1].pointers.cpu().numpy(), vp)
    assert mo.pixels.dtype == torch.int64
    assert np.array_equal(mo.pixels.cpu().numpy(), canonical_pixels(vp, g["crop_pixels"]))
    np.testing.assert_allclose(mo.features.cpu().numpy(), g["crop_features"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(images.x.cpu().numpy(), g["crop_x"])
    assert tuple(images.crop_size) == tuple(g["crop_state_size"])
    assert np.array_equal(images.crop_offsets.cpu().numpy(), g["crop_state_offsets"])
    assert tuple(images.img_size) == tuple(g["crop_img_size"])
    images.x = images.x[:, :, ::2, ::2].contiguous()
    assert float(images.downscale) == float(g["half_downscale"])
    assert tuple(images.img_size) == tuple(g["half_img_size"])
    # the mapped features follow the new resolution
    feats = images.get_mapped_features()
    assert feats.shape == (mo.pixels.shape[0], g["x"].shape[1])
    far = M.ImageMapping.from_csr_arrays(
        T(g["in_pointers"]), T(g["in_images"]), T(g["in_pixels"]), T(g["in_features"]),
        view_pointers=T(g["in_view_pointers"])).crop(
            (4, 4), torch.full((n_img, 2), 1000, dtype=torch.int64, device=DEV))
    assert np.array_equal(far.pointers.cpu().numpy(), g["far_pointers"])
    assert np.array_equal(far.pixels.cpu().numpy(), g["far_pixels"])


def test_crop_image_groups_golden(golden):
    """CropImageGroups: size families, centred offsets, cropped maps and rebuilt
    mappings of every family against the reference's transform."""
    from mvfusion.images import ImageData, SameSettingImageData
    from mvfusion.transforms import CropImageGroups
    g = golden("crop_groups")
    n_img, (W, H) = int(g["n_img"]), (int(v) for v in g["size"])

    def sample():
        m = M.ImageMapping.from_csr_arrays(
            T(g["in_pointers"]), T(g["in_images"]), T(g["in_pixels"]), T(g["in_features"]),
            view_pointers=T(g["in_view_pointers"]))
        x = torch.arange(n_img * 2 * H * W, dtype=torch.float32, device=DEV).view(n_img, 2, H, W)
        return SameSettingImageData(
            path=[f"v{i}" for i in range(n_img)], pos=T(g["pos_img"]), ref_size=(W, H),
            proj_upscale=1, downscale=1, mappings=m, x=x)

    class Data:
        pos = torch.zeros(3, 3)

    for tag, kw in (("pad0", dict(padding=0, min_size=32)), ("pad8", dict(padding=8, min_size=64))):
        _, out = CropImageGroups(**kw)(Data(), sample())
        assert isinstance(out, ImageData) and len(out) == int(g[tag + "_n"])
        for j, im in enumerate(out):
            t = f"{tag}_{j}"
            assert [str(p) for p in im.path] == [str(p) for p in g[t + "_path"]]
            assert tuple(im.crop_size) == tuple(g[t + "_crop_size"])
            assert np.array_equal(im.crop_offsets.cpu().numpy(), g[t + "_crop_offsets"])
            assert tuple(im.x.shape) == tuple(g[t + "_x_shape"])
            assert np.array_equal(im.x.double().sum(dim=(1, 2, 3)).cpu().numpy(), g[t + "_x_sum"])
            mo = im.mappings
            assert np.array_equal(mo.pointers.cpu().numpy(), g[t + "_pointers"])
            assert np.array_equal(mo.images.cpu().numpy(), g[t + "_images"])
            vp = g[t + "_view_pointers"]
            assert np.array_equal(mo.values[1].pointers.cpu().numpy(), vp)
            assert np.array_equal(mo.pixels.cpu().numpy(), canonical_pixels(vp, g[t + "_pixels"]))
            np.testing.assert_allclose(mo.features.cpu().numpy(), g[t + "_features"], rtol=1e-6, atol=1e-7)
    empty = sample()[torch.zeros(0, dtype=torch.long, device=DEV)]
    _, out = CropImageGroups()(Data(), empty)
    assert isinstance(out, ImageData) and len(out) == 1 and out[0].num_views == 0


def test_from_dense_and_merge_large_vs_oracle():
    """The device CSR assembly (one key per row, one radix sort) at a size where
    the sort runs many blocks: from_dense on 300 k dense rows with repeated rows,
    unseen points and unseen images, then select_points('merge') that folds the
    points 5 to 1 (duplicate pixels dropped, view features averaged over the
    merged views); both against the oracle."""
    import mvf_oracle as orc
    rng = np.random.default_rng(12)
    n_pts, n_img, rows, F = 40_000, 23, 300_000, 3
    point_ids = rng.integers(0, n_pts - 50, rows)            # the last points stay empty
    image_ids = rng.integers(0, n_img, rows)
    pixels = np.stack([rng.integers(0, 40, rows), rng.integers(0, 30, rows)], 1).astype(np.int16)
    feats = rng.random((rows, F), dtype=np.float32)
    want = orc.mapping_from_dense(point_ids, image_ids, pixels, feats, num_points=n_pts)
    m = M.ImageMapping.from_dense(T(point_ids).cuda(), T(image_ids).cuda(), T(pixels).cuda(),
                                  T(feats).cuda(), num_points=n_pts)
    assert m.is_cuda if hasattr(m, "is_cuda") else m.pointers.is_cuda
    assert np.array_equal(N(m.pointers), want["pointers"])
    assert np.array_equal(N(m.images), want["images"])
    assert np.array_equal(N(m.atomic_csr_indexing), want["view_pointers"])
    assert np.array_equal(N(m.pixels), want["pixels"])       # (x, y) order inside a view on both sides
    np.testing.assert_allclose(N(m.features), want["features"], rtol=1e-6, atol=1e-7)

    idx = np.arange(n_pts) // 5
    idx = idx[rng.permutation(n_pts)]                        # merge targets in any order
    assert np.array_equal(np.unique(idx), np.arange(idx.max() + 1))
    small = 6_000                                            # the oracle's merge loops over groups
    sub_rows = point_ids < small
    m_small = M.ImageMapping.from_dense(
        T(point_ids[sub_rows]).cuda(), T(image_ids[sub_rows]).cuda(), T(pixels[sub_rows]).cuda(),
        T(feats[sub_rows]).cuda(), num_points=small)
    ref_small = orc.mapping_from_dense(point_ids[sub_rows], image_ids[sub_rows], pixels[sub_rows],
                                       feats[sub_rows], num_points=small)
    idx_small = np.arange(small) // 5
    idx_small = idx_small[rng.permutation(small)]
    want_m = orc.select_points(ref_small, idx_small, mode="merge")
    got_m = m_small.select_points(T(idx_small).cuda(), mode="merge")
    assert np.array_equal(N(got_m.pointers), want_m["pointers"])
    assert np.array_equal(N(got_m.images), want_m["images"])
    assert np.array_equal(N(got_m.atomic_csr_indexing), want_m["view_pointers"])
    assert np.array_equal(N(got_m.pixels), want_m["pixels"])
    np.testing.assert_allclose(N(got_m.features), want_m["features"], rtol=1e-6, atol=1e-7)
    assert want_m["pixels"].shape[0] < ref_small["pixels"].shape[0]        # duplicates were dropped
    # the full-size merge agrees with the same merge composed from torch sorts
    got = m.select_points(T(idx).cuda(), mode="merge")
    assert got.num_groups == idx.max() + 1 and got.pixels.shape[0] < rows
    vp = N(got.atomic_csr_indexing)
    assert np.all(np.diff(vp) > 0) and vp[-1] == got.pixels.shape[0]


@pytest.mark.parametrize("n_img,dtype", [(37, np.int16), (1500, np.int64)])
def test_image_stats_large(n_img, dtype):
    """Pixel count and bounding box per image in one pass over the views:
    few images (partial results staged in LDS) and more images than the
    staging holds (direct atomics), int16 and int64 pixels with negative
    coordinates, empty views, images without pixels."""
    rng = np.random.default_rng(n_img)
    n_views = 200_000
    sizes = rng.integers(0, 5, n_views)
    vp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    images = rng.integers(0, n_img - 3, n_views).astype(np.int64)      # the last ids stay empty
    pix = rng.integers(-300, 3000, (int(vp[-1]), 2)).astype(dtype)
    m = M.ImageMapping.from_csr_arrays(
        T(np.arange(n_views + 1, dtype=np.int64)), T(images), T(pix), None, view_pointers=T(vp))
    got = m.image_stats(n_img).cpu().numpy()
    ids = np.repeat(images, sizes)
    want = np.zeros((n_img, 5), np.int64)
    np.add.at(want[:, 0], ids, 1)
    for col, fn, src in ((1, np.minimum, 0), (2, np.maximum, 0), (3, np.minimum, 1), (4, np.maximum, 1)):
        acc = np.full(n_img, np.iinfo(np.int64).max if fn is np.minimum else np.iinfo(np.int64).min)
        fn.at(acc, ids, pix[:, src].astype(np.int64))
        want[:, col] = np.where(want[:, 0] > 0, acc, 0)
    assert np.array_equal(got, want)
    boxes = m.bounding_boxes
    assert boxes[0].dtype == m.pixels.dtype and boxes[0].shape[0] == int(images.max()) + 1
    assert np.array_equal(boxes[1].cpu().numpy(), want[:int(images.max()) + 1, 2].astype(dtype))


def test_from_dense_negative_coordinates_and_feature_dtype():
    """Negative ids or pixel coordinates are refused (they are unsigned fields of the sort
    keys; the reference's composite keys do not order them either), and the features come
    back in the dtype they were given in."""
    from mvfusion.mapping import ImageMapping
    rng = np.random.default_rng(3)
    rows, n_pts = 4000, 300
    pid = rng.integers(0, n_pts, rows)
    iid = rng.integers(0, 4, rows)
    pix = rng.integers(0, 100, (rows, 2)).astype(np.int16)
    feat = rng.random((rows, 3))
    _, keep = np.unique(np.stack([pid, iid, pix[:, 0], pix[:, 1]], 1), axis=0, return_index=True)
    pid, iid, pix, feat = pid[keep], iid[keep], pix[keep], feat[keep]
    m64 = ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat), num_points=n_pts)
    m32 = ImageMapping.from_dense(T(pid), T(iid), T(pix), T(feat.astype(np.float32)), num_points=n_pts)
    assert m64.features.dtype == torch.float64 and m32.features.dtype == torch.float32
    assert torch.equal(m64.pointers, m32.pointers) and torch.equal(m64.pixels, m32.pixels)
    assert torch.equal(m64.features.float(), m32.features)
    for bad in ((pid - 5, iid, pix), (pid, iid - 1, pix), (pid, iid, pix - np.int16(3))):
        with pytest.raises(ValueError):
            ImageMapping.from_dense(T(bad[0]), T(bad[1]), T(bad[2]), T(feat), num_points=n_pts)
