"""Differentiable forward of the attentive pools and their set encoders.

Calls that need no autograd graph (``.eval()`` under ``torch.no_grad()``, or
nothing requiring a gradient) run through the fused MFMA layer-chain and
attention kernels (``dense.py``), which implement BatchNorm with running
statistics and record no graph.  In ``.train()``, and in ``.eval()`` whenever an
input or a parameter requires a gradient, the pools are composed here from
differentiable pieces instead (BatchNorm follows the module's mode): the Linear
of a dense layer is this package's streaming MFMA product for its output and
input gradient (``mvf_linear_rows``; a library GEMM for shapes it does not
take) and ``mvf_xty`` for its weight and bias gradients (``linear``), its
BatchNorm1d + LeakyReLU tail one pair of this package's kernels forward and
one backward (``batch_norm_act``), the attention-weighted sum one kernel each
way (``weighted_segment_sum``), and the CSR operations are this package's
kernels, all wrapped in ``torch.autograd.Function``
(``pooling.segment_csr`` / ``gather_csr`` / ``segment_softmax_csr``).  The
computation follows modules/multimodal/pooling.py: ``DeepSetFeat.forward``
:658-681, ``MinMaxDiffSetFeat.forward`` :577-597, ``MLPSetFeat.forward``
:688-699, ``GroupBimodalCSRPool.forward`` :263-315 and
``QKVBimodalCSRPool.forward`` :454-547.
"""
import ctypes
import math

import torch

from . import _lib as L
from . import pooling as P


class _BatchNormAct(torch.autograd.Function):
    """BatchNorm1d (+ LeakyReLU) of float32 rows on the kernels
    mvf_batch_norm_act_forward / _backward: two passes over the rows forward,
    two backward; only the layer's input is kept for the backward."""

    @staticmethod
    def forward(ctx, z, weight, bias, rm, rv, eps, slope, batch_stats, factor, col_sums):
        lib = L.load()
        z = z.detach().contiguous()
        rows, C = z.shape
        dev = z.device
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty_like(z)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        nbytes = lib.mvf_batch_norm_act_workspace_bytes(C)
        ws = L.workspace(nbytes, dev)
        with torch.cuda.device(dev):
            L.check(lib.mvf_batch_norm_act_forward(
                L.ptr(z), rows, C, L.ptr(w), L.ptr(b), L.ptr(rm), L.ptr(rv), int(batch_stats),
                float(factor), float(eps), int(slope is not None),
                0.0 if slope is None else float(slope), L.ptr(y), L.ptr(mean), L.ptr(invstd),
                L.ptr(col_sums), 0 if col_sums is None else col_sums.shape[0],
                L.ptr(ws), nbytes, L.stream_ptr(dev)))
        ctx.save_for_backward(z, w, b, mean, invstd)
        ctx.slope, ctx.batch_stats = slope, bool(batch_stats)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        lib = L.load()
        z, w, b, mean, invstd = ctx.saved_tensors
        rows, C = z.shape
        dev = z.device
        grad_y = grad_y.contiguous()
        gz = torch.empty_like(z)
        gw = torch.empty(C, dtype=torch.float32, device=dev) if w is not None and ctx.needs_input_grad[1] else None
        gb = torch.empty(C, dtype=torch.float32, device=dev) if b is not None and ctx.needs_input_grad[2] else None
        nbytes = lib.mvf_batch_norm_act_workspace_bytes(C)
        ws = L.workspace(nbytes, dev)
        slope = ctx.slope
        with torch.cuda.device(dev):
            L.check(lib.mvf_batch_norm_act_backward(
                L.ptr(grad_y), L.ptr(z), rows, C, L.ptr(w), L.ptr(b), L.ptr(mean), L.ptr(invstd),
                int(ctx.batch_stats), int(slope is not None), 0.0 if slope is None else float(slope),
                L.ptr(gz), L.ptr(gw), L.ptr(gb), L.ptr(ws), nbytes, L.stream_ptr(dev)))
        return gz, gw, gb, None, None, None, None, None, None, None


def batch_norm_act(z, bn, act, col_sums=None):
    """``act(bn(z))`` for an ``nn.BatchNorm1d`` (train or eval mode, running
    statistics and ``num_batches_tracked`` updated as the module does) followed
    by a LeakyReLU, on this package's kernels when z is a float32 [rows, C]
    tensor on the HIP device; other inputs and activations take the modules.
    col_sums: partial column sums of z and z * z as ``linear(..., want_sums=True)``
    returns them (batch statistics without another pass over z)."""
    fused_act = isinstance(act, torch.nn.LeakyReLU)
    if not (z.is_cuda and z.dtype == torch.float32 and z.dim() == 2 and z.shape[0] > 0
            and isinstance(bn, torch.nn.BatchNorm1d)
            and all(t is None or (t.is_cuda and t.dtype == torch.float32)
                    for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))):
        return act(bn(z))
    batch_stats = bn.training or bn.running_mean is None
    factor = 0.0
    if batch_stats:
        if z.shape[0] < 2:
            raise ValueError(
                f"Expected more than 1 value per channel when training, got input size {tuple(z.shape)}")
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
            factor = (1.0 / float(bn.num_batches_tracked)) if bn.momentum is None else bn.momentum
    # running statistics: updated in train mode, read in eval mode, unused by an eval-mode
    # module that tracks none (it normalises with batch statistics)
    use_running = bn.running_mean is not None and (bn.training or not batch_stats)
    y = _BatchNormAct.apply(z, bn.weight, bn.bias,
                            bn.running_mean if use_running else None,
                            bn.running_var if use_running else None, bn.eps,
                            float(act.negative_slope) if fused_act else None, batch_stats, factor,
                            col_sums if batch_stats else None)
    return y if fused_act else act(y)


def xty(x, y=None):
    """x^T . y of two float32 [rows, *] tensors on the HIP device (mvf_xty);
    without y the column sums of x, [A, 1]."""
    lib = L.load()
    x = x.contiguous()
    dev = x.device
    if y is None:
        out = torch.empty((x.shape[1], 1), dtype=torch.float32, device=dev)
        nbytes = lib.mvf_xty_workspace_bytes()
        ws = L.workspace(nbytes, dev)
        with torch.cuda.device(dev):
            L.check(lib.mvf_xty(L.ptr(x), x.stride(0), x.shape[1], None, 0, 1, x.shape[0],
                                L.ptr(out), 1, L.ptr(ws), nbytes, L.stream_ptr(dev)))
        return out
    y = y.contiguous()
    out = torch.empty((x.shape[1], y.shape[1]), dtype=torch.float32, device=dev)
    nbytes = lib.mvf_xty_workspace_bytes()
    ws = L.workspace(nbytes, dev)
    with torch.cuda.device(dev):
        L.check(lib.mvf_xty(L.ptr(x), x.stride(0), x.shape[1], L.ptr(y), y.stride(0), y.shape[1],
                            x.shape[0], L.ptr(out), out.stride(0), L.ptr(ws), nbytes,
                            L.stream_ptr(dev)))
    return out


class _Weights:
    """What dense.LayerSpec reads of an nn.Linear."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self.out_features, self.in_features = weight.shape


def _product(x, weight, bias, want_sums=False):
    """(x W^T + b, partial column sums or None) over many rows: the streaming
    float32 MFMA kernel (mvf_linear_rows, which can leave the column sums of its
    output for BatchNorm's batch statistics) for at most 64 input and output
    columns, one launch of the layer-chain kernel with a single plain layer for
    wider ones (both stream the rows at a higher rate than the library GEMM
    does for these shapes), the library GEMM otherwise."""
    if max(weight.shape) > 256 or x.shape[0] < 4096:
        return torch.nn.functional.linear(x, weight, bias), None
    dense = P._dense
    w = weight.detach()
    specs, xp = dense.aligned_input(
        [dense.LayerSpec(_Weights(w if w.is_contiguous() else w.contiguous(), bias))], x.detach())
    lib = L.load()
    cout, cin = specs[0].cout, specs[0].cin
    if not (lib.mvf_linear_rows_supported(cin, cout) and xp.is_contiguous() and xp.data_ptr() % 16 == 0):
        return dense.run_layers(specs, xp), None
    wp, bp = specs[0].weight.float().contiguous(), None if bias is None else bias.detach().float().contiguous()
    dev = xp.device
    y = torch.empty((xp.shape[0], cout), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if not want_sums:
            L.check(lib.mvf_linear_rows(L.ptr(xp), cin, L.ptr(wp), L.ptr(bp), cout, xp.shape[0], L.ptr(y),
                                        L.stream_ptr(dev)))
            return y, None
        nbytes = lib.mvf_linear_rows_sums_bytes(cout)
        sums = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
        n = ctypes.c_int32(0)
        L.check(lib.mvf_linear_rows_sums(L.ptr(xp), cin, L.ptr(wp), L.ptr(bp), cout, xp.shape[0], L.ptr(y),
                                         L.ptr(sums), nbytes, ctypes.byref(n), L.stream_ptr(dev)))
    return y, sums[:n.value * 2 * cout].view(n.value, 2, cout)


class _Linear(torch.autograd.Function):
    """y = x W^T + b: the output and the input gradient (as many rows as the
    input) by ``_product``, the weight gradient by mvf_xty: a reduction over
    every row into a few thousand sums, where library GEMMs reach a few percent
    of the memory bandwidth."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_sums):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        y, sums = _product(x, weight, bias, want_sums)
        if sums is None:
            sums = torch.empty(0, dtype=torch.float64, device=x.device)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, grad_y, _grad_sums):
        x, weight = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        gx = _product(grad_y, weight.t(), None)[0] if ctx.needs_input_grad[0] else None
        gw = xty(grad_y, x) if ctx.needs_input_grad[1] else None
        gb = xty(grad_y).view(-1) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


def linear(lin, x, want_sums=False):
    """``lin(x)`` for an ``nn.Linear`` on float32 [rows, C] device tensors,
    with the weight gradient on this package's kernel; anything else takes the
    module.  want_sums: returns (output, partial column sums of the output and
    its square or None) for ``batch_norm_act``."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0
            and lin.weight.is_cuda and lin.weight.dtype == torch.float32):
        y, sums = _Linear.apply(x, lin.weight, lin.bias, bool(want_sums))
        return (y, sums if sums.numel() else None) if want_sums else y
    return (lin(x), None) if want_sums else lin(x)


def mlp(seq, x):
    """MLP of base_modules.py:38-48: per block Linear (streaming MFMA products,
    weight gradient by mvf_xty), then BatchNorm1d and the activation in one pair
    of kernels; in train mode the Linear leaves the column sums the batch
    statistics are made of."""
    for block in seq:
        bn = block[1].batch_norm
        if bn.training or bn.running_mean is None:
            z, sums = linear(block[0], x, want_sums=True)
        else:
            z, sums = linear(block[0], x), None
        x = batch_norm_act(z, bn, block[2], col_sums=sums)
    return x


def mlp_on_rows_and_sets(seq, x, x_set, ptr):
    """``mlp(seq, cat((x, gather_csr(x_set, ptr)), 1))`` without the concatenated rows: the
    first Linear is split by columns, [x | x_set[seg]] W^T = x Wa^T + (x_set Wb^T + b)[seg],
    so the set features go through their half once per segment instead of once per row
    (DeepSetFeat's 'concatenation' fusion, pooling.py:669-681)."""
    lin, bn, act = seq[0][0], seq[0][1].batch_norm, seq[0][2]
    d = x.shape[1]
    za = _Linear.apply(x, lin.weight[:, :d], None, False)[0]
    zb = _Linear.apply(x_set, lin.weight[:, d:], lin.bias, False)[0]
    z = za + P.gather_csr(zb, ptr)
    y = batch_norm_act(z, bn, act)
    return mlp(seq[1:], y) if len(seq) > 1 else y


def deep_set_feat(m, x, ptr):
    x = mlp(m.mlp_elt_1, x)
    sets = [P.segment_csr(x, ptr, reduce=p) for p in m.pool]
    if m.use_num:
        sets.append(P._set_num(ptr).view(-1, 1))
    x_set = torch.cat(sets, dim=-1) if len(sets) > 1 else sets[0]
    x_set = mlp(m.mlp_set, x_set)
    if m.fusion == 'concatenation' and x.is_cuda and x.dtype == torch.float32:
        return mlp_on_rows_and_sets(m.mlp_elt_2, x, x_set, ptr)
    x_set = P.gather_csr(x_set, ptr)
    if m.fusion == 'residual':
        x_out = x + x_set
    elif m.fusion == 'concatenation':
        x_out = torch.cat((x, x_set), dim=-1)
    else:
        x_out = torch.cat((x, x + x_set), dim=-1)
    return mlp(m.mlp_elt_2, x_out)


def min_max_diff_set_feat(m, x, ptr):
    parts = [x]
    if m.use_min:
        parts.append(x - P.segment_gather_csr(x, ptr, reduce='min'))
    if m.use_max:
        parts.append(x - P.segment_gather_csr(x, ptr, reduce='max'))
    if m.use_num:
        parts.append(P._set_num(ptr).repeat_interleave(ptr[1:] - ptr[:-1]).view(-1, 1))
    return mlp(m.mlp, torch.cat(parts, dim=1) if len(parts) > 1 else x)


def map_encoder(m, x, ptr):
    if isinstance(m, P.DeepSetFeat):
        return deep_set_feat(m, x, ptr)
    if isinstance(m, P.MinMaxDiffSetFeat):
        return min_max_diff_set_feat(m, x, ptr)
    return mlp(m.mlp, x)


class _WeightedSegmentSum(torch.autograd.Function):
    """segment_csr(x * expand_group_feat(att), ptr, 'sum') in one pass over x,
    and both gradients in one pass (mvf_weighted_segment_sum / _backward)."""

    @staticmethod
    def forward(ctx, x, att, ptr, num_groups):
        lib = L.load()
        x, att = x.detach().contiguous(), att.detach().contiguous()
        dev = x.device
        nseg = ptr.shape[0] - 1
        out = torch.empty((nseg, x.shape[1]), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.mvf_weighted_segment_sum(L.ptr(x), L.ptr(att), x.shape[1], num_groups,
                                                 L.ptr(ptr), nseg, L.ptr(out), L.stream_ptr(dev)))
        ctx.save_for_backward(x, att, ptr)
        ctx.num_groups = num_groups
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = L.load()
        x, att, ptr = ctx.saved_tensors
        dev = x.device
        rows, C = x.shape
        grad_out = grad_out.contiguous()
        row2seg = P._dense.csr_row_segments(ptr, rows, outside=-1)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gatt = torch.empty_like(att) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(dev):
            L.check(lib.mvf_weighted_segment_sum_backward(
                L.ptr(grad_out), L.ptr(x), L.ptr(att), C, ctx.num_groups, L.ptr(row2seg), rows,
                ptr.shape[0] - 1, L.ptr(gx), L.ptr(gatt), L.stream_ptr(dev)))
        return gx, gatt, None, None


def weighted_segment_sum(x, att, ptr, num_groups):
    """Sum over the rows of every segment of ``x * expand_group_feat(att)``
    (pooling.py:289-296, 519-526): one kernel forward and one backward when the
    channels split into equal groups of a power-of-two multiple of 4 channels
    (rows outside every segment get zero gradients); the composition of
    ``expand_group_feat`` and ``segment_csr`` otherwise."""
    C = x.shape[1]
    cv = C // 4
    ok = (x.is_cuda and x.dtype == torch.float32 and att.dtype == torch.float32 and x.dim() == 2
          and x.shape[0] > 0 and C % 4 == 0 and C % num_groups == 0 and (C // num_groups) % 4 == 0
          and cv <= 64 and cv & (cv - 1) == 0
          and (cv // num_groups) & (cv // num_groups - 1) == 0
          and att.numel() == x.shape[0] * num_groups and ptr.dtype == torch.int64)
    if not ok:
        return P.segment_csr(x * P.expand_group_feat(att, num_groups, C), ptr, reduce='sum')
    return _WeightedSegmentSum.apply(x, att, ptr, num_groups)


def _group_shape_ok(C, num_groups):
    cv = C // 4
    return (C % 4 == 0 and C % num_groups == 0 and (C // num_groups) % 4 == 0 and 1 <= cv <= 64
            and cv & (cv - 1) == 0 and (cv // num_groups) & (cv // num_groups - 1) == 0)


class _SegmentDot(torch.autograd.Function):
    """comp[r, g] = sum over the channels c of group g of keys[r, c] * queries[segment of r, c]:
    the compatibilities of the views with the query of their point
    (pooling.py:497-512: ``(K * gather_csr(Q)).view(-1, G, D).sum(2)``) without the
    gathered queries or the products in memory.  Forward and both gradients are the
    kernels of the attention-weighted sum with the roles of the tensors exchanged."""

    @staticmethod
    def forward(ctx, keys, queries, ptr, num_groups):
        lib = L.load()
        keys, queries = keys.detach().contiguous(), queries.detach().contiguous()
        dev = keys.device
        rows, C = keys.shape
        nseg = ptr.shape[0] - 1
        row2seg = P._dense.csr_row_segments(ptr, rows, outside=-1)
        comp = torch.empty((rows, num_groups), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            # (the weights argument is only read for the row gradient, which is not asked for)
            L.check(lib.mvf_weighted_segment_sum_backward(
                L.ptr(queries), L.ptr(keys), L.ptr(keys), C, num_groups, L.ptr(row2seg), rows, nseg,
                None, L.ptr(comp), L.stream_ptr(dev)))
        ctx.save_for_backward(keys, queries, ptr, row2seg)
        ctx.num_groups = num_groups
        return comp

    @staticmethod
    def backward(ctx, grad_comp):
        lib = L.load()
        keys, queries, ptr, row2seg = ctx.saved_tensors
        dev = keys.device
        rows, C = keys.shape
        nseg = ptr.shape[0] - 1
        grad_comp = grad_comp.contiguous()
        gk = gq = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                gk = torch.empty_like(keys)
                L.check(lib.mvf_weighted_segment_sum_backward(
                    L.ptr(queries), L.ptr(keys), L.ptr(grad_comp), C, ctx.num_groups, L.ptr(row2seg),
                    rows, nseg, L.ptr(gk), None, L.stream_ptr(dev)))
            if ctx.needs_input_grad[1]:
                gq = torch.empty_like(queries)
                L.check(lib.mvf_weighted_segment_sum(L.ptr(keys), L.ptr(grad_comp), C, ctx.num_groups,
                                                     L.ptr(ptr), nseg, L.ptr(gq), L.stream_ptr(dev)))
        return gk, gq, None, None


def segment_dot(keys, queries, ptr, num_groups):
    """[rows, G] group-wise dot products of every key row with the query row of its segment
    (queries: one row per segment); the composition with ``gather_csr`` for shapes the
    kernels do not take."""
    C = keys.shape[1]
    if (keys.is_cuda and keys.dtype == torch.float32 and queries.dtype == torch.float32
            and keys.dim() == 2 and keys.shape[0] > 0 and queries.shape == (ptr.shape[0] - 1, C)
            and ptr.dtype == torch.int64 and _group_shape_ok(C, num_groups)):
        return _SegmentDot.apply(keys, queries, ptr, num_groups)
    q = P.gather_csr(queries, ptr)
    return (keys.view(-1, num_groups, C // num_groups) * q.view(-1, num_groups, C // num_groups)).sum(dim=2)


def attend(m, x_mod, comp, ptr):
    """Softmax of the compatibilities over the views of a point, attention-
    weighted sum of the values, optional gate from the best compatibility."""
    att = P.segment_softmax_csr(comp, ptr, scaling=m.group_scaling)
    x_pool = weighted_segment_sum(x_mod, att, ptr, m.num_groups)
    gate = None
    if m.G is not None:
        gate = m.G(P.segment_csr(comp, ptr, reduce='max'))
        x_pool = x_pool * P.expand_group_feat(gate, m.num_groups, m.out_mod)
    return x_pool, att, gate


def group_pool(m, x_main, x_mod, x_map, ptr):
    x_map = map_encoder(m.E_map, x_map, ptr)
    x_mod = mlp(m.E_mod, x_mod)
    x_mix = mlp(m.E_mix, torch.cat([x_map, x_mod], dim=1)) if m.use_mod else x_map
    comp = linear(m.E_score, x_mix)
    x_pool, att, gate = attend(m, x_mod, comp, ptr)
    return x_pool, dict(x_map=x_map, x_mod=x_mod, C=comp, A=att, G=gate)


def qkv_pool(m, x_main, x_mod, x_map, ptr):
    x_main = mlp(m.E_main, x_main)
    x_map = map_encoder(m.E_map, x_map, ptr)
    x_mod = mlp(m.E_mod, x_mod)
    if m.use_mod_k:
        keys = linear(m.K, mlp(m.E_mix_K, torch.cat([x_map, x_mod], dim=1)))
    else:
        keys = linear(m.K, x_map)
    if not m.use_mod_q and not m.save_last:
        # the queries stay one row per point: every view meets the row of its point in the kernel
        queries = linear(m.Q, x_main)
        if m.dim_scaling:
            queries = queries / math.sqrt(m.nc_qk)
        comp = segment_dot(keys, queries, ptr, m.num_groups)
    else:
        if m.use_mod_q:
            queries = linear(m.Q, mlp(m.E_mix_Q, torch.cat([P.gather_csr(x_main, ptr), x_mod], dim=1)))
        else:
            queries = P.gather_csr(linear(m.Q, x_main), ptr)
        comp = (keys.view(-1, m.num_groups, m.nc_qk) * queries.view(-1, m.num_groups, m.nc_qk)).sum(dim=2)
        if m.dim_scaling:
            comp = comp / math.sqrt(m.nc_qk)
    x_pool, att, gate = attend(m, x_mod, comp, ptr)
    return x_pool, dict(x_map=x_map, x_mod=x_mod, K=keys, Q=queries, C=comp, A=att, G=gate)
