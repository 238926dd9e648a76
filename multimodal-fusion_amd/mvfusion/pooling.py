"""Host-side mirror of ``torch_points3d/modules/multimodal/pooling.py``: same
class names, constructor arguments, ``forward(x_main, x_mod, x_map, csr_idx)``
signatures and ``state_dict`` keys, with the segment reductions done by the HIP
kernels of libmvfusion.so instead of ``torch_scatter``.

Reference lines (DeepViewAgg-Wildscenes/torch_points3d/modules/multimodal/
pooling.py): BimodalCSRPool :14-71, segment_softmax_csr :770-822, gather_csr
:825-853, segment_gather_csr :856-868, expand_group_feat :760-767,
group_sizes :749-757.  Forward (inference) only: no autograd graph is recorded
through the HIP kernels.
"""
import math
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


def segment_csr(src, indptr, out=None, reduce="sum"):
    """Drop-in for ``torch_scatter.segment_csr`` on dim 0 of a 1D/2D float32
    tensor with 1D pointers: empty segment -> 0, mean = sum / count."""
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


def gather_csr(src, csr_idx):
    """Gather index-level src values into element-level values based on CSR
    indices (pooling.py:825-853)."""
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
    """Softmax inside CSR segments (pooling.py:770-822)."""
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
