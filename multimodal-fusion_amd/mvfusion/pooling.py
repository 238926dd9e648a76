"""Host-side mirror of ``torch_points3d/modules/multimodal/pooling.py``: same
class names, constructor arguments, ``forward(x_main, x_mod, x_map, csr_idx)``
signatures and ``state_dict`` keys, with the segment reductions done by the HIP
kernels of libmvfusion.so instead of ``torch_scatter``.

Reference lines (DeepViewAgg-Wildscenes/torch_points3d/modules/multimodal/
pooling.py): BimodalCSRPool :14-71, segment_softmax_csr :770-822, gather_csr
:825-853, segment_gather_csr :856-868, expand_group_feat :760-767,
group_sizes :749-757, HeuristicBimodalCSRPool :74-156, GroupBimodalCSRPool :159-319,
QKVBimodalCSRPool :322-551, MinMaxDiffSetFeat :554-601, DeepSetFeat :604-685,
MLPSetFeat :688-699, Gating :702-727.  ``segment_csr``, ``gather_csr`` and
``segment_softmax_csr`` are differentiable; the attentive pools use fused
inference kernels in ``.eval()`` and a differentiable composition
(``training.py``) in ``.train()``.
"""
import math
import sys
from abc import ABC

import torch
import torch.nn as nn

from . import _lib as L


def _prep(src, csr_idx):
    if not src.is_cuda:
        raise RuntimeError(
            "mvfusion pooling runs on a HIP device only: move the features and "
            "the CSR pointers to the GPU (there is no CPU fallback).")
    if src.dtype != torch.float32:
        raise TypeError(f"mvfusion pooling expects float32 features, got {src.dtype}")
    if csr_idx.dim() != 1:
        raise ValueError("CSR pointers must be a 1D tensor.")
    ptr = csr_idx.to(device=src.device, dtype=torch.int64).contiguous()
    return src.detach().contiguous(), ptr


def _wants_graph(module, *tensors):
    """Does a forward of `module` on `tensors` have to record an autograd graph?
    The fused inference kernels record none, so such a call takes the composed,
    differentiable path (training.py) whatever the module's mode: in eval mode
    BatchNorm then uses its running statistics there, as the reference does."""
    if not torch.is_grad_enabled():
        return False
    if any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in module.parameters())


class _SegmentCSR(torch.autograd.Function):
    """segment_csr with the gradient torch_scatter defines for it: sum / mean
    spread the segment's gradient over its rows, max / min route it to the first
    row that holds the reduced value."""

    @staticmethod
    def forward(ctx, src, indptr, reduce):
        res = segment_csr(src.detach(), indptr, reduce=reduce)
        ctx.reduce = reduce
        ctx.shape = src.shape
        keep = reduce in ('max', 'min')
        ctx.save_for_backward(indptr, *((src.detach(), res) if keep else ()))
        return res

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        ptr = saved[0].to(device=grad_out.device, dtype=torch.int64).contiguous()
        src, res = (saved[1].contiguous(), saved[2]) if len(saved) == 3 else (None, None)
        g = grad_out.detach().float().contiguous()
        grad_src = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        nseg = ptr.shape[0] - 1
        rows = ctx.shape[0]
        C_ = 1
        for d in ctx.shape[1:]:
            C_ *= int(d)
        if nseg > 0 and rows > 0 and C_ > 0:
            lib = L.load()
            with torch.cuda.device(g.device):
                L.check(lib.mvf_segment_csr_backward(
                    L.ptr(g), L.ptr(src), L.ptr(res), rows, C_, L.ptr(ptr), nseg,
                    L.REDUCE[ctx.reduce], L.ptr(grad_src), L.stream_ptr(g.device)))
        return grad_src, None, None


def segment_csr(src, indptr, out=None, reduce="sum"):
    """Drop-in for ``torch_scatter.segment_csr`` on dim 0 of a 1D/2D float32
    tensor with 1D pointers: empty segment -> 0, mean = sum / count.
    Differentiable with respect to ``src`` (when ``out`` is not given)."""
    if out is None and torch.is_grad_enabled() and src.requires_grad:
        return _SegmentCSR.apply(src, indptr, reduce)
    src_c, ptr = _prep(src, indptr)
    lib = L.load()
    nseg = ptr.shape[0] - 1
    rows = src_c.shape[0]
    C_ = 1
    for d in src_c.shape[1:]:
        C_ *= int(d)
    res = torch.empty((nseg,) + tuple(src_c.shape[1:]), dtype=torch.float32,
                      device=src_c.device) if out is None else out
    if nseg > 0 and C_ > 0:
        with torch.cuda.device(src_c.device):
            L.check(lib.mvf_segment_csr(
                L.ptr(src_c), rows, C_, L.ptr(ptr), nseg, L.REDUCE[reduce],
                L.ptr(res), L.stream_ptr(src_c.device)))
    return res


def segment_csr_two_level(src, inner_ptr, outer_ptr, inner="max", outer=("max",)):
    """``segment_csr(segment_csr(src, inner_ptr, inner), outer_ptr, o)`` for the one
    or two reductions ``o`` in ``outer``, in a single pass over ``src`` (the
    pixel -> view and view -> point pools of BimodalCSRPool, pooling.py:14-71,
    without the view-level intermediate).  Returns a tuple of [n_outer, C]
    tensors, bit-identical to the separate calls."""
    if isinstance(outer, str):
        outer = (outer,)
    if not 1 <= len(outer) <= 2:
        raise ValueError("one or two outer reductions")
    src_c, iptr = _prep(src, inner_ptr)
    _, optr = _prep(src, outer_ptr)
    if src_c.dim() != 2:
        raise NotImplementedError("segment_csr_two_level expects a 2D source")
    n_inner, n_outer = iptr.shape[0] - 1, optr.shape[0] - 1
    rows, C_ = src_c.shape
    outs = [torch.empty((n_outer, C_), dtype=torch.float32, device=src_c.device) for _ in outer]
    if n_outer > 0 and C_ > 0:
        lib = L.load()
        if inner in ("max", "min") and n_inner > 0 and rows >= 16 * n_inner:
            # long inner segments (the pixels of whole splats): a thread per output
            # value would walk them alone; the inner level goes through the
            # wave-per-segment reduction first (same values: max / min do not round)
            view = torch.empty((n_inner, C_), dtype=torch.float32, device=src_c.device)
            with torch.cuda.device(src_c.device):
                L.check(lib.mvf_segment_csr(L.ptr(src_c), rows, C_, L.ptr(iptr), n_inner,
                                            L.REDUCE[inner], L.ptr(view), L.stream_ptr(src_c.device)))
            src_c, rows = view, n_inner
            iptr = torch.arange(n_inner + 1, dtype=torch.int64, device=src_c.device)
        with torch.cuda.device(src_c.device):
            L.check(lib.mvf_segment_csr_two_level(
                L.ptr(src_c), rows, C_, L.ptr(iptr), n_inner, L.REDUCE[inner], L.ptr(optr), n_outer,
                L.REDUCE[outer[0]], L.ptr(outs[0]), L.REDUCE[outer[-1]],
                L.ptr(outs[1]) if len(outs) == 2 else None, L.stream_ptr(src_c.device)))
    return tuple(outs)


class _GatherCSR(torch.autograd.Function):
    """gather_csr; its gradient is the segment sum of the incoming gradient."""

    @staticmethod
    def forward(ctx, src, csr_idx):
        ctx.save_for_backward(csr_idx)
        return gather_csr(src.detach(), csr_idx)

    @staticmethod
    def backward(ctx, grad_out):
        (csr_idx,) = ctx.saved_tensors
        return segment_csr(grad_out.detach().float().contiguous(), csr_idx, reduce='sum'), None


class _SegmentSoftmaxCSR(torch.autograd.Function):
    """segment_softmax_csr with y * (g - sum(g * y)) / scale as gradient."""

    @staticmethod
    def forward(ctx, src, csr_idx, eps, scaling):
        y = segment_softmax_csr(src.detach(), csr_idx, eps=eps, scaling=scaling)
        ctx.scaling = bool(scaling)
        ctx.save_for_backward(csr_idx, y)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        csr_idx, y = ctx.saved_tensors
        g = grad_out.detach().float().contiguous()
        ptr = csr_idx.to(device=g.device, dtype=torch.int64).contiguous()
        grad = torch.zeros_like(y)
        if y.numel() > 0:
            lib = L.load()
            G = 1 if y.dim() == 1 else y.shape[1]
            with torch.cuda.device(g.device):
                L.check(lib.mvf_segment_softmax_csr_backward(
                    L.ptr(g), L.ptr(y), G, L.ptr(ptr), ptr.shape[0] - 1, int(ctx.scaling),
                    L.ptr(grad), L.stream_ptr(g.device)))
        return grad, None, None, None


def gather_csr(src, csr_idx):
    """Gather index-level src values into element-level values based on CSR
    indices (pooling.py:825-853).  Differentiable with respect to ``src``."""
    if torch.is_grad_enabled() and src.requires_grad:
        return _GatherCSR.apply(src, csr_idx)
    if not torch.is_floating_point(src):
        raise ValueError(
            '`gather_csr` can only be computed over tensors with '
            'floating point data types.')
    if src.dim() > 2:
        raise NotImplementedError(
            '`gather_csr` can only be computed over 1D or 2D source tensors.')
    src_c, ptr = _prep(src, csr_idx)
    lib = L.load()
    nseg = ptr.shape[0] - 1
    rows = int(ptr[-1].item()) if nseg > 0 else 0
    C_ = 1 if src_c.dim() == 1 else src_c.shape[1]
    out = torch.empty((rows,) + tuple(src_c.shape[1:]), dtype=torch.float32,
                      device=src_c.device)
    if rows > 0:
        with torch.cuda.device(src_c.device):
            L.check(lib.mvf_gather_csr(
                L.ptr(src_c), C_, L.ptr(ptr), nseg, rows, L.ptr(out),
                L.stream_ptr(src_c.device)))
    return out


def segment_gather_csr(src, csr_idx, reduce='sum'):
    """pooling.py:856-868."""
    return gather_csr(segment_csr(src, csr_idx, reduce=reduce), csr_idx)


def segment_softmax_csr(src, csr_idx, eps=1e-12, scaling=False):
    """Softmax inside CSR segments (pooling.py:770-822).  Differentiable with
    respect to ``src``."""
    if torch.is_grad_enabled() and src.requires_grad:
        return _SegmentSoftmaxCSR.apply(src, csr_idx, eps, scaling)
    if not torch.is_floating_point(src):
        raise ValueError(
            '`segment_csr_softmax` can only be computed over tensors with '
            'floating point data types.')
    if csr_idx.dim() != 1:
        raise ValueError(
            '`segment_csr_softmax` can only be computed over 1D CSR indices.')
    if src.dim() > 2:
        raise NotImplementedError(
            '`segment_csr_softmax` can only be computed over 1D or 2D source '
            'tensors.')
    src_c, ptr = _prep(src, csr_idx)
    lib = L.load()
    nseg = ptr.shape[0] - 1
    G = 1 if src_c.dim() == 1 else src_c.shape[1]
    out = torch.empty_like(src_c)
    if src_c.numel() > 0:
        with torch.cuda.device(src_c.device):
            L.check(lib.mvf_segment_softmax_csr(
                L.ptr(src_c), G, L.ptr(ptr), nseg, float(eps), int(bool(scaling)),
                L.ptr(out), L.stream_ptr(src_c.device)))
    return out


def _map_arrays(x, mappings):
    """Checked device arrays of a mapping for a [B, C, H, W] feature map:
    image of every view, view pointers, pixels, image of every pixel."""
    if not x.is_cuda:
        raise RuntimeError("mvfusion pooling runs on a HIP device only (there is no CPU fallback).")
    if x.dim() != 4 or x.dtype != torch.float32:
        raise TypeError("expected a float32 [B, C, H, W] feature map")
    dev = x.device
    images = mappings.images.to(dev, torch.int64).contiguous()
    vptr = mappings.atomic_csr_indexing.to(dev, torch.int64).contiguous()
    pixels = mappings.pixels.to(dev, torch.int16).contiguous()
    B, _, H, W = x.shape
    if images.shape[0] and (
            int(images.max()) >= B or int(pixels[:, 0].max()) >= W or int(pixels[:, 1].max()) >= H
            or int(pixels.min()) < 0 or int(images.min()) < 0):
        raise IndexError("mappings index outside the feature map")
    return images, vptr, pixels


def _gather_rows(x, batch, pixels):
    rows = pixels.shape[0]
    B, C_, H, W = x.shape
    out = torch.empty((rows, C_), dtype=torch.float32, device=x.device)
    if rows:
        with torch.cuda.device(x.device):
            L.check(L.load().mvf_gather_features(
                L.ptr(x), B, C_, H, W, L.ptr(batch), L.ptr(pixels), rows, L.ptr(out),
                L.stream_ptr(x.device)))
    return out


def _scatter_rows(grad_rows, shape, batch, pixels):
    """Gradient of _gather_rows: a zero map of `shape` plus the rows at their pixels."""
    B, C_, H, W = shape
    grad_x = torch.zeros(shape, dtype=torch.float32, device=grad_rows.device)
    rows = pixels.shape[0]
    if rows:
        g = grad_rows.detach().float().contiguous()
        with torch.cuda.device(g.device):
            L.check(L.load().mvf_gather_features_backward(
                L.ptr(g), B, C_, H, W, L.ptr(batch), L.ptr(pixels), rows, L.ptr(grad_x),
                L.stream_ptr(g.device)))
    return grad_x


class _GatherFeatures(torch.autograd.Function):
    """x[feature_map_indexing]; gradient = scatter-add of the rows into the maps."""

    @staticmethod
    def forward(ctx, x, batch, pixels):
        ctx.save_for_backward(batch, pixels)
        ctx.shape = x.shape
        return _gather_rows(x.detach().contiguous(), batch, pixels)

    @staticmethod
    def backward(ctx, grad_out):
        batch, pixels = ctx.saved_tensors
        return _scatter_rows(grad_out, ctx.shape, batch, pixels), None, None


class _GatherAtomicPool(torch.autograd.Function):
    """Fused gather + pixel -> view pool; the backward materialises the gathered
    rows once to route the gradient like segment_csr does, then scatters it."""

    @staticmethod
    def forward(ctx, x, images, vptr, pixels, mode):
        xd = x.detach().contiguous()
        rows = images.shape[0]
        B, C_, H, W = xd.shape
        out = torch.empty((rows, C_), dtype=torch.float32, device=xd.device)
        if rows:
            with torch.cuda.device(xd.device):
                L.check(L.load().mvf_gather_pool(
                    L.ptr(xd), B, C_, H, W, L.ptr(images), L.ptr(vptr), L.ptr(pixels), rows,
                    L.REDUCE[mode], L.ptr(out), L.stream_ptr(xd.device)))
        ctx.mode, ctx.shape = mode, x.shape
        ctx.save_for_backward(xd, images, vptr, pixels, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        xd, images, vptr, pixels, out = ctx.saved_tensors
        batch = images.repeat_interleave(vptr[1:] - vptr[:-1])
        src = _gather_rows(xd, batch, pixels)
        g = grad_out.detach().float().contiguous()
        grad_rows = torch.zeros_like(src)
        nseg, rows, C_ = vptr.shape[0] - 1, src.shape[0], src.shape[1]
        if nseg > 0 and rows > 0:
            with torch.cuda.device(g.device):
                L.check(L.load().mvf_segment_csr_backward(
                    L.ptr(g), L.ptr(src), L.ptr(out), rows, C_, L.ptr(vptr), nseg,
                    L.REDUCE[ctx.mode], L.ptr(grad_rows), L.stream_ptr(g.device)))
        return _scatter_rows(grad_rows, ctx.shape, batch, pixels), None, None, None, None


def gather_atomic_pool(x, mappings, mode='max'):
    """``BimodalCSRPool(mode)(None, x[mappings.feature_map_indexing], None,
    mappings.atomic_csr_indexing)`` in one kernel: x is the [B, C, H, W] batch of
    image feature maps, ``mappings`` an ImageMapping whose pixels are at the
    resolution of x (core/multimodal/image.py:1272-1297, 1881-1902).  Returns the
    view-level features [Vw, C]; differentiable with respect to x."""
    images, vptr, pixels = _map_arrays(x, mappings)
    return _GatherAtomicPool.apply(x, images, vptr, pixels, mode)


def gather_features(x, mappings):
    """``x[mappings.feature_map_indexing]`` for a [B, C, H, W] batch of feature
    maps (core/multimodal/image.py:1295, 1881-1895): one row of C features per
    mapped pixel; differentiable with respect to x."""
    _map_arrays(x, mappings)
    dev = x.device
    pixels = mappings.pixels.to(dev, torch.int16).contiguous()
    batch = mappings.feature_map_indexing[0].to(dev).contiguous()
    return _GatherFeatures.apply(x, batch, pixels)


_PADDING = {'border': 0, 'zeros': 1, 'reflection': 2}


class _SparseInterpolation(torch.autograd.Function):
    """Bilinear sampling; gradient = the same four weights scattered back."""

    @staticmethod
    def forward(ctx, x, coords, batch, mode):
        xd = x.detach().float().contiguous()
        n = coords.shape[0]
        B, C_, H, W = xd.shape
        out = torch.empty((n, C_), dtype=torch.float32, device=xd.device)
        if n:
            with torch.cuda.device(xd.device):
                L.check(L.load().mvf_sparse_interpolation(
                    L.ptr(xd), B, C_, H, W, L.ptr(coords), L.ptr(batch), n, mode, L.ptr(out),
                    L.stream_ptr(xd.device)))
        ctx.mode, ctx.shape, ctx.dtype = mode, x.shape, x.dtype
        ctx.save_for_backward(coords, batch)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, batch = ctx.saved_tensors
        B, C_, H, W = ctx.shape
        g = grad_out.detach().float().contiguous()
        grad_x = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        if coords.shape[0]:
            with torch.cuda.device(g.device):
                L.check(L.load().mvf_sparse_interpolation_backward(
                    L.ptr(g), B, C_, H, W, L.ptr(coords), L.ptr(batch), coords.shape[0], ctx.mode,
                    L.ptr(grad_x), L.stream_ptr(g.device)))
        return grad_x.to(ctx.dtype), None, None, None


def sparse_interpolation(features, coords, batch, padding_mode='border'):
    """Bilinear interpolation of a batch of feature maps [B, C, H, W] at the
    [N, 2] coordinates ``coords`` (row, column, both in [0, 1]) of the maps
    ``batch`` [N] (core/multimodal/image.py:105-170).  Returns [N, C];
    differentiable with respect to the feature maps."""
    assert len(features.shape) == 4
    assert coords.shape[0] == batch.shape[0]
    assert len(coords.shape) == 2 and coords.shape[1] == 2
    if padding_mode not in _PADDING:
        raise NotImplementedError(f"Unknown padding_mode='{padding_mode}'")
    if not features.is_cuda:
        raise RuntimeError("mvfusion pooling runs on a HIP device only (there is no CPU fallback).")
    dev = features.device
    coords = coords.detach().to(dev, torch.float32).contiguous()
    batch = batch.to(dev, torch.int64).contiguous()
    n = coords.shape[0]
    B = features.shape[0]
    if n:
        assert 0 <= coords.min()
        assert coords.max() <= 1
        if int(batch.max()) >= B or int(batch.min()) < 0:
            raise IndexError("batch index outside the feature maps")
    return _SparseInterpolation.apply(features, coords, batch, _PADDING[padding_mode])


def group_sizes(num_elements, num_groups):
    """pooling.py:749-757."""
    sizes = torch.full(
        (num_groups,), math.floor(num_elements / num_groups), dtype=torch.long)
    sizes += torch.arange(num_groups) < num_elements - sizes.sum()
    return sizes


def expand_group_feat(A, num_groups, num_channels):
    """pooling.py:760-767."""
    if num_groups == 1:
        A = A.view(-1, 1)
    elif num_groups < num_channels:
        if num_channels % num_groups == 0 and A.dim() == 2:
            # equal groups: the same values as repeat_interleave, as a broadcast (its gradient
            # is a plain sum over the group's channels instead of an index_add)
            A = A.unsqueeze(2).expand(-1, num_groups, num_channels // num_groups).reshape(
                A.shape[0], num_channels)
        else:
            sizes = group_sizes(num_channels, num_groups).to(A.device)
            A = A.repeat_interleave(sizes, dim=1)
    return A


class BimodalCSRPool(nn.Module, ABC):
    """max / mean / min / sum pooling of modality features over CSR groups
    (pooling.py:14-71)."""

    _POOLING_MODES = ['max', 'mean', 'min', 'sum']

    def __init__(self, mode='max', save_last=False, **kwargs):
        super(BimodalCSRPool, self).__init__()
        assert mode in self._POOLING_MODES, \
            f"Unsupported mode '{mode}'. Expected one of: {self._POOLING_MODES}"
        self._mode = mode
        self.save_last = save_last
        self._last_x_map = None
        self._last_x_mod = None
        self._last_idx = None
        self._last_view_num = None

    def forward(self, x_main, x_mod, x_map, csr_idx):
        """
        :param x_main: N x F_main
        :param x_mod: V x F_mod
        :param x_map: V x F_map
        :param csr_idx:
        :return: x_pool
        """
        x_pool = segment_csr(x_mod, csr_idx, reduce=self._mode)
        if self.save_last:
            self._last_x_map = x_map
            self._last_x_mod = x_mod
            self._last_idx = torch.arange(
                csr_idx.shape[0] - 1, device=x_mod.device).repeat_interleave(
                csr_idx[1:] - csr_idx[:-1])
            self._last_view_num = csr_idx[1:] - csr_idx[:-1]
        return x_pool


# --------------------------------------------------------------------------- #
# attentive pools (pooling.py:74-727)                                          #
# --------------------------------------------------------------------------- #
from . import dense as _dense                      # noqa: E402
from .dense import MLP                             # noqa: E402,F401  (state_dict-compatible container)

_local_modules = sys.modules[__name__]


def _device_ptr(x, csr_idx):
    if not x.is_cuda:
        raise RuntimeError(
            "mvfusion pooling runs on a HIP device only: move the features and "
            "the CSR pointers to the GPU (there is no CPU fallback).")
    if not csr_idx.is_cuda and csr_idx.numel() > 0 and (
            int(csr_idx[0]) != 0 or int(csr_idx[-1]) != x.shape[0]):
        # (pointers on the device are not read back: the kernels then only make sure that rows
        # outside every group cannot make them read out of bounds)
        raise ValueError(
            f"CSR pointers cover rows {int(csr_idx[0])}..{int(csr_idx[-1])} of {x.shape[0]} feature rows")
    return csr_idx.to(device=x.device, dtype=torch.int64).contiguous()


def _last_idx(csr_idx, device):
    return torch.arange(csr_idx.shape[0] - 1, device=device).repeat_interleave(
        csr_idx[1:] - csr_idx[:-1])


def _set_num(ptr):
    # Heuristic to normalize in [0,1] (pooling.py:590,675)
    return torch.sqrt(1 / (ptr[1:] - ptr[:-1] + 1e-3)).float()


def nearest_power_of_2(x, min_power=16):
    """pooling.py:730-746."""
    x = int(x)
    if x < min_power:
        return min_power
    previous_power = 2 ** ((x - 1).bit_length() - 1)
    next_power = 2 ** (x - 1).bit_length()
    if x - previous_power < next_power - x:
        return previous_power
    return next_power


class HeuristicBimodalCSRPool(nn.Module, ABC):
    """Selects the modality features of the view with the largest / smallest
    value of one handcrafted mapping feature (pooling.py:74-156).  Ties go to
    the first row of the group."""

    _MODES = ['max', 'min']
    _FEATURES = [
        'normalized_depth', 'linearity', 'planarity', 'scattering',
        'orientation_to_the_surface', 'normalized_pixel_height', 'density',
        'occlusion']

    def __init__(self, mode='max', feat=0, save_last=False, **kwargs):
        super(HeuristicBimodalCSRPool, self).__init__()
        assert mode in self._MODES, \
            f"Unsupported mode '{mode}'. Expected one of: {self._MODES}."
        self._mode = mode
        feat = self._FEATURES.index(feat) if isinstance(feat, str) else feat
        assert feat < len(self._FEATURES), \
            f"Feat={feat} is too large. Expected feat<{len(self._FEATURES)}."
        self._feat = feat
        self.save_last = save_last
        self._last_x_map = None
        self._last_x_mod = None
        self._last_idx = None
        self._last_view_num = None

    def forward(self, x_main, x_mod, x_map, csr_idx):
        ptr = _device_ptr(x_mod, csr_idx)
        x_pool = _dense.segment_pick(x_mod, x_map[:, self._feat], ptr, self._mode == 'max')
        if self.save_last:
            self._last_x_map = x_map
            self._last_x_mod = x_mod
            self._last_idx = _last_idx(csr_idx, x_mod.device)
            self._last_view_num = csr_idx[1:] - csr_idx[:-1]
        return x_pool

    def extra_repr(self) -> str:
        return f'mode={self._mode}, feat={self._FEATURES[self._feat]}, ' \
               f'save_last={self.save_last}'


class MinMaxDiffSetFeat(nn.Module, ABC):
    """Element-wise set features from difference-to-min / -max and set size
    (pooling.py:554-601)."""

    def __init__(self, d_in, d_out, use_min=True, use_max=True, use_num=False, **kwargs):
        super(MinMaxDiffSetFeat, self).__init__()
        self.d_in = d_in
        self.d_out = d_out
        self.use_min = use_min
        self.use_max = use_max
        self.use_num = use_num
        in_mlp = d_in * (1 + self.use_min + self.use_max) + self.use_num
        self.mlp = MLP([in_mlp, d_out, d_out], bias=False)

    def embed(self, x, ptr, row2seg, tail=(), dot=None):
        parts = [x]
        if self.use_min:
            parts.append(x - segment_gather_csr(x, ptr, reduce='min'))
        if self.use_max:
            parts.append(x - segment_gather_csr(x, ptr, reduce='max'))
        if self.use_num:
            num = _set_num(ptr)
            parts.append(num.repeat_interleave(ptr[1:] - ptr[:-1]).view(-1, 1))
        x_out = torch.cat(parts, dim=1) if len(parts) > 1 else x
        return _dense.run_layers(_dense.mlp_layers(self.mlp) + list(tail), x_out, dot=dot)

    def forward(self, x, csr_idx):
        ptr = _device_ptr(x, csr_idx)
        if self.training or _wants_graph(self, x):
            from . import training
            return training.min_max_diff_set_feat(self, x, ptr)
        return self.embed(x, ptr, None)

    def extra_repr(self) -> str:
        repr_attr = ['use_min', 'use_max', 'use_num']
        return "\n".join([f'{a}={getattr(self, a)}' for a in repr_attr])


class DeepSetFeat(nn.Module, ABC):
    """Element-wise set features from shared learned features
    (pooling.py:604-685)."""

    _POOLING_MODES = ['max', 'mean', 'min', 'sum']
    _FUSION_MODES = ['residual', 'concatenation', 'both']

    def __init__(self, d_in, d_out, pool='max', fusion='concatenation', use_num=False, **kwargs):
        super(DeepSetFeat, self).__init__()
        pool = pool.split('_')
        assert all([p in self._POOLING_MODES for p in pool]), \
            f"Unsupported pool='{pool}'. Expected elements of: " \
            f"{self._POOLING_MODES}"
        self.pool = pool
        if fusion not in self._FUSION_MODES:
            raise NotImplementedError(
                f"Unknown fusion='{fusion}'. Please choose among "
                f"supported modes: {self._FUSION_MODES}.")
        self.fusion = fusion
        self.d_in = d_in
        self.d_out = d_out
        self.use_num = use_num
        self.mlp_elt_1 = MLP([d_in, d_out, d_out], bias=False)
        in_set_mlp = d_out * len(self.pool) + self.use_num
        self.mlp_set = MLP([in_set_mlp, d_out, d_out], bias=False)
        in_last_mlp = d_out if fusion == 'residual' else d_out * 2
        self.mlp_elt_2 = MLP([in_last_mlp, d_out, d_out], bias=False)

    def embed(self, x, ptr, row2seg, tail=(), dot=None):
        """forward with optional extra layers (the pools' K / E_score Linear)
        fused behind mlp_elt_2, and optionally their dot product with a query
        row per element (run_layers' dot)."""
        if row2seg is None:
            row2seg = _dense.csr_row_segments(ptr, x.shape[0], outside=0)
        elt_1, x = _dense.aligned_input(_dense.mlp_layers(self.mlp_elt_1), x)
        if self.pool == ['max'] and _dense.pool_fusable(elt_1, x, 1):
            # the rows and their per-segment maximum out of one launch
            x1, x_max = _dense.layers_with_segment_max(elt_1, x, ptr, row2seg)
            sets = [x_max]
        else:
            x1 = _dense.run_layers(elt_1, x)
            sets = [segment_csr(x1, ptr, reduce=p) for p in self.pool]
        if self.use_num:
            sets.append(_set_num(ptr).view(-1, 1))
        x_set = torch.cat(sets, dim=-1) if len(sets) > 1 else sets[0]
        x_set = _dense.run_layers(_dense.mlp_layers(self.mlp_set), x_set)
        mode = {'concatenation': 0, 'residual': 1, 'both': 2}[self.fusion]
        # gather_csr(x_set) and the fusion happen while mlp_elt_2 loads its input
        return _dense.run_layers(
            _dense.mlp_layers(self.mlp_elt_2) + list(tail), x1, None, x_set, row2seg, mode, dot=dot)

    def forward(self, x, csr_idx):
        ptr = _device_ptr(x, csr_idx)
        if self.training or _wants_graph(self, x):
            from . import training
            return training.deep_set_feat(self, x, ptr)
        return self.embed(x, ptr, None)

    def extra_repr(self) -> str:
        repr_attr = ['pool', 'fusion', 'use_num']
        return "\n".join([f'{a}={getattr(self, a)}' for a in repr_attr])


class MLPSetFeat(nn.Module, ABC):
    """Element-wise set features with a simple MLP (pooling.py:688-699)."""

    def __init__(self, d_in, d_out, **kwargs):
        super(MLPSetFeat, self).__init__()
        self.d_in = d_in
        self.d_out = d_out
        self.mlp = MLP([d_in, d_out, d_out], bias=False)

    def embed(self, x, ptr, row2seg, tail=(), dot=None):
        return _dense.run_layers(_dense.mlp_layers(self.mlp) + list(tail), x, dot=dot)

    def forward(self, x, csr_idx):
        if self.training or _wants_graph(self, x):
            from . import training
            return training.mlp(self.mlp, x)
        return self.embed(x, None, None)


class Gating(nn.Module):
    """Rectified-tanh gating mechanism with learnable linear correction
    (pooling.py:702-727).  Inside the pools the gate is evaluated by the
    attention kernel; this forward serves direct calls.  As in the reference,
    ``activation`` is recorded but the forward always applies tanh(relu(x))
    (pooling.py:717-723)."""

    def __init__(self, num_groups, weight=True, bias=True, activation='tanh+'):
        super(Gating, self).__init__()
        self.num_groups = num_groups
        self.weight = nn.Parameter(torch.ones(1, num_groups)) if weight else None
        self.bias = nn.Parameter(torch.zeros(1, num_groups)) if bias else None
        if activation not in ('tanh+', 'sigmoid'):
            raise ValueError(f"Activation '{activation}' not supported for Gating")
        self._activation = activation

    def forward(self, x):
        if self.weight is not None:
            x = x * self.weight
        if self.bias is not None:
            x = x + self.bias
        x = torch.tanh(torch.relu(x))
        return x.view(-1, self.num_groups).squeeze(1)

    def extra_repr(self) -> str:
        return f'num_groups={self.num_groups}, ' \
            f'weight={self.weight is not None}, bias={self.bias is not None}'


_MAP_ENCODERS = {'DeepSetFeat': DeepSetFeat, 'MinMaxDiffSetFeat': MinMaxDiffSetFeat,
                 'MLPSetFeat': MLPSetFeat}


class _AttentivePool(nn.Module, ABC):
    """What GroupBimodalCSRPool and QKVBimodalCSRPool share."""

    def _chan2group(self, device):
        if self.num_groups == 1:
            return None
        cache = self.__dict__.setdefault('_c2g', {})
        key = str(device)
        if key not in cache:
            if self.num_groups < self.out_mod:
                sizes = group_sizes(self.out_mod, self.num_groups)
                c2g = torch.arange(self.num_groups).repeat_interleave(sizes)
            else:
                c2g = torch.arange(self.out_mod)
            cache[key] = c2g.to(device=device, dtype=torch.int32)
        return cache[key]

    def _forward_train(self, composed, x_main, x_mod, x_map, csr_idx):
        """Training mode, or any call that must record an autograd graph:
        differentiable composition (training.py) instead of the fused inference
        kernels."""
        ptr = _device_ptr(x_mod, csr_idx)
        x_pool, parts = composed(self, x_main, x_mod, x_map, ptr)
        if self.save_last:
            self._last_x_map = parts['x_map']
            self._last_x_mod = parts['x_mod']
            self._last_idx = _last_idx(csr_idx, x_mod.device)
            self._last_view_num = csr_idx[1:] - csr_idx[:-1]
            if 'K' in parts:
                self._last_K, self._last_Q = parts['K'], parts['Q']
            self._last_C = parts['C']
            self._last_A = parts['A']
            if self.G:
                self._last_G = parts['G']
        return x_pool


class GroupBimodalCSRPool(_AttentivePool):
    """Attention over the views of a point from mapping features only
    (pooling.py:159-319)."""

    def __init__(
            self, in_map=None, in_mod=None, out_mod=None, num_groups=1,
            use_mod=False, gating=True, group_scaling=True, save_last=False,
            nc_inner=32, map_encoder='DeepSetFeat', **kwargs):
        super(GroupBimodalCSRPool, self).__init__()
        self.nc_inner = nc_inner
        self.save_last = save_last
        self._last_x_map = None
        self._last_x_mod = None
        self._last_idx = None
        self._last_view_num = None
        self._last_C = None
        self._last_A = None
        self._last_G = None
        assert 1 <= num_groups <= in_mod, \
            f"Number of groups must be between 1 and in_mod={in_mod}."
        out_mod = in_mod if out_mod is None else out_mod
        self.in_mod = in_mod
        self.out_mod = out_mod
        self.use_mod = use_mod
        self.num_groups = num_groups
        self.group_scaling = group_scaling
        self.E_map = _MAP_ENCODERS[map_encoder](in_map, nc_inner, **kwargs)
        self.E_mod = MLP([in_mod, out_mod, out_mod], bias=False)
        if self.use_mod:
            in_mix = nc_inner + out_mod
            out_mix = nc_inner
            mid_mix = nearest_power_of_2((in_mix + out_mix) / 2, out_mix * 2)
            self.E_mix = MLP([in_mix, mid_mix, out_mix], bias=False)
        self.E_score = nn.Linear(nc_inner, num_groups, bias=True)
        self.G = Gating(num_groups, bias=True) if gating else None

    def forward(self, x_main, x_mod, x_map, csr_idx):
        """
        :param x_main: N x F_main
        :param x_mod: V x F_mod
        :param x_map: V x F_map
        :param csr_idx:
        :return: x_pool
        """
        if self.training or _wants_graph(self, x_main, x_mod, x_map):
            from . import training
            return self._forward_train(training.group_pool, x_main, x_mod, x_map, csr_idx)
        ptr = _device_ptr(x_mod, csr_idx)
        row2seg = _dense.csr_row_segments(ptr, x_mod.shape[0], outside=0)
        score = [_dense.linear_layer(self.E_score)]
        e_mod, x_mod = _dense.aligned_input(_dense.mlp_layers(self.E_mod), x_mod)
        fuse_pool = (not (self.save_last or self.use_mod)
                     and _dense.pool_fusable(e_mod, x_mod, self.num_groups))
        if not fuse_pool:
            x_mod = _dense.run_layers(e_mod, x_mod)
        if self.use_mod or self.save_last:
            x_map = self.E_map.embed(x_map, ptr, row2seg)
            if self.use_mod:
                comp = _dense.run_layers(_dense.mlp_layers(self.E_mix) + score, x_map, None, x_mod)
            else:
                comp = _dense.run_layers(score, x_map)
        else:
            comp = self.E_map.embed(x_map, ptr, row2seg, tail=score)
        x_pool, comp, att, gate = _dense.attention_pool(
            self.out_mod if fuse_pool else x_mod, ptr, self.num_groups,
            self._chan2group(x_mod.device), comp=comp,
            group_scaling=self.group_scaling,
            gate_w=None if self.G is None else self.G.weight,
            gate_b=None if self.G is None else self.G.bias, gating=self.G is not None)
        if fuse_pool:
            _dense.pooled_layers(e_mod, x_mod, att, row2seg, ptr, x_pool)
        if self.save_last:
            self._last_x_map = x_map
            self._last_x_mod = x_mod
            self._last_idx = _last_idx(csr_idx, x_mod.device)
            self._last_view_num = csr_idx[1:] - csr_idx[:-1]
            self._last_C = comp
            self._last_A = att
            if self.G:
                self._last_G = gate.view(-1, self.num_groups).squeeze(1)
        return x_pool

    def extra_repr(self) -> str:
        repr_attr = ['num_groups', 'use_mod', 'group_scaling', 'save_last']
        return "\n".join([f'{a}={getattr(self, a)}' for a in repr_attr])


class QKVBimodalCSRPool(_AttentivePool):
    """Key-query attention over the views of a point: keys from the mapping
    features, queries from the main modality (pooling.py:322-551)."""

    def __init__(
            self, in_main=None, in_map=None, in_mod=None, out_mod=None,
            num_groups=1, use_mod_q=False, use_mod_k=False, nc_qk=8,
            gating=True, dim_scaling=True, group_scaling=False, debug=False,
            save_last=False, nc_inner=32, map_encoder='DeepSetFeat', **kwargs):
        super(QKVBimodalCSRPool, self).__init__()
        self.nc_inner = nc_inner
        self.save_last = save_last
        self._last_x_map = None
        self._last_x_mod = None
        self._last_idx = None
        self._last_view_num = None
        self._last_Q = None
        self._last_K = None
        self._last_C = None
        self._last_A = None
        self._last_G = None
        self.debug = debug
        if debug:
            raise NotImplementedError(
                "QKVBimodalCSRPool(debug=True) replaces its inputs by random numbers "
                "(pooling.py:463-468); not reproduced")
        assert 1 <= num_groups <= in_mod, \
            f"Number of groups must be between 1 and in_mod={in_mod}."
        out_mod = in_mod if out_mod is None else out_mod
        self.in_mod = in_mod
        self.out_mod = out_mod
        self.nc_qk = nc_qk
        self.use_mod_q = use_mod_q
        self.use_mod_k = use_mod_k
        self.num_groups = num_groups
        self.dim_scaling = dim_scaling
        self.group_scaling = group_scaling
        self.E_main = MLP([in_main, nc_inner, nc_inner], bias=False)
        self.E_map = _MAP_ENCODERS[map_encoder](in_map, nc_inner, **kwargs)
        self.E_mod = MLP([in_mod, out_mod, out_mod], bias=False)
        if self.use_mod_q:
            in_mix = nc_inner + out_mod
            out_mix = nc_inner
            mid_mix = nearest_power_of_2((in_mix + out_mix) / 2, out_mix * 2)
            self.E_mix_Q = MLP([in_mix, mid_mix, out_mix], bias=False)
        self.Q = nn.Linear(nc_inner, nc_qk * num_groups, bias=True)
        if self.use_mod_k:
            in_mix = nc_inner + in_mod
            out_mix = nc_inner
            mid_mix = nearest_power_of_2((in_mix + out_mix) / 2, out_mix * 2)
            self.E_mix_K = MLP([in_mix, mid_mix, out_mix], bias=False)
        self.K = nn.Linear(nc_inner, nc_qk * num_groups, bias=True)
        self.G = Gating(num_groups, bias=True) if gating else None

    def forward(self, x_main, x_mod, x_map, csr_idx):
        """
        :param x_main: N x F_main
        :param x_mod: V x F_mod
        :param x_map: V x F_map
        :param csr_idx:
        :return: x_pool
        """
        if self.training or _wants_graph(self, x_main, x_mod, x_map):
            from . import training
            return self._forward_train(training.qkv_pool, x_main, x_mod, x_map, csr_idx)
        ptr = _device_ptr(x_mod, csr_idx)
        row2seg = _dense.csr_row_segments(ptr, x_mod.shape[0], outside=0)
        k_lin, q_lin = [_dense.linear_layer(self.K)], [_dense.linear_layer(self.Q)]

        # values : V x F_mod.  Unless the encoded values are asked for (save_last,
        # mixed keys / queries) they are produced last, inside the launch that sums
        # them per point (_dense.pooled_layers).
        e_mod, x_mod = _dense.aligned_input(_dense.mlp_layers(self.E_mod), x_mod)
        fuse_pool = (not (self.save_last or self.use_mod_q or self.use_mod_k)
                     and _dense.pool_fusable(e_mod, x_mod, self.num_groups))
        if not fuse_pool:
            x_mod = _dense.run_layers(e_mod, x_mod)

        # queries : N x (D x num_groups), or V x ... when the modality is mixed in
        if self.use_mod_q:
            x_main = _dense.run_layers(_dense.mlp_layers(self.E_main), x_main)
            queries = _dense.run_layers(
                _dense.mlp_layers(self.E_mix_Q) + q_lin, x_main, row2seg, x_mod)
        else:
            queries = _dense.run_layers(_dense.mlp_layers(self.E_main) + q_lin, x_main)

        # keys : V x (D x num_groups), the K Linear fused behind the map encoder.
        # With one group and plain keys / queries the compatibilities come out of
        # the same launch (the keys meet their query row in its epilogue).
        keys = comp = None
        if self.use_mod_k or self.save_last:
            x_map = self.E_map.embed(x_map, ptr, row2seg)
            if self.use_mod_k:
                keys = _dense.run_layers(_dense.mlp_layers(self.E_mix_K) + k_lin, x_map, None, x_mod)
            else:
                keys = _dense.run_layers(k_lin, x_map)
        elif self.num_groups == 1 and not self.use_mod_q and _dense.dot_fusable(k_lin, queries):
            div = math.sqrt(self.nc_qk) if self.dim_scaling else 0.0
            comp = self.E_map.embed(x_map, ptr, row2seg, tail=k_lin, dot=(queries, row2seg, div))
        else:
            keys = self.E_map.embed(x_map, ptr, row2seg, tail=k_lin)

        x_pool, comp, att, gate = _dense.attention_pool(
            self.out_mod if fuse_pool else x_mod, ptr, self.num_groups,
            self._chan2group(x_mod.device), keys=keys, comp=comp,
            queries=queries, row2seg=None if self.use_mod_q else row2seg, nc_qk=self.nc_qk,
            dim_scaling=self.dim_scaling, group_scaling=self.group_scaling,
            gate_w=None if self.G is None else self.G.weight,
            gate_b=None if self.G is None else self.G.bias, gating=self.G is not None)
        if fuse_pool:
            _dense.pooled_layers(e_mod, x_mod, att, row2seg, ptr, x_pool)

        if self.save_last:
            self._last_x_map = x_map
            self._last_x_mod = x_mod
            self._last_idx = _last_idx(csr_idx, x_mod.device)
            self._last_view_num = csr_idx[1:] - csr_idx[:-1]
            self._last_K = keys
            self._last_Q = queries if self.use_mod_q else gather_csr(queries, ptr)
            self._last_C = comp
            self._last_A = att
            if self.G:
                self._last_G = gate.view(-1, self.num_groups).squeeze(1)
        return x_pool

    def extra_repr(self) -> str:
        repr_attr = ['dim_scaling', 'group_scaling', 'save_last']
        return "\n".join([f'{a}={getattr(self, a)}' for a in repr_attr])
