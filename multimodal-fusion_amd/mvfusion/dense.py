"""Dense building blocks of the attentive pools: parameter containers with the
reference's ``state_dict`` layout (``MLP`` / ``FastBatchNorm1d``,
core/common_modules/base_modules.py:38-48,131-156) and the wrappers of the
MFMA layer-chain, attention and row-segment kernels of libmvfusion.so.

Inference only: the kernels implement eval-mode BatchNorm (running statistics)
and record no autograd graph.
"""
import copy
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L

_MAX_C = 64          # widest layer output (and non-first input) of one launch
_MAX_LAYERS = 3


class FastBatchNorm1d(nn.Module):
    """Parameter container of base_modules.py:131-156 (``batch_norm`` child)."""

    def __init__(self, num_features, momentum=0.1, **kwargs):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(num_features, momentum=momentum, **kwargs)

    def forward(self, x):
        raise RuntimeError(
            "mvfusion.FastBatchNorm1d only holds parameters; the enclosing MLP runs "
            "through mvfusion.dense.run_layers")


def MLP(channels, activation=None, bn_momentum=0.1, bias=True):
    """Same module tree (and state_dict keys) as base_modules.py:38-48."""
    activation = nn.LeakyReLU(0.2, inplace=True) if activation is None else activation
    return nn.Sequential(*[
        nn.Sequential(
            nn.Linear(channels[i - 1], channels[i], bias=bias),
            FastBatchNorm1d(channels[i], momentum=bn_momentum),
            activation)
        for i in range(1, len(channels))])


class LayerSpec:
    """One Linear (+ eval BatchNorm) (+ LeakyReLU) layer as the kernel sees it."""

    def __init__(self, linear, bn=None, slope=None):
        self.weight = linear.weight.detach()
        self.bias = None if linear.bias is None else linear.bias.detach()
        self.bn = bn
        self.slope = slope
        self.cin, self.cout = int(linear.in_features), int(linear.out_features)

    def tensors(self, lo=0, hi=None):
        """(weight, bias, bn_weight, bn_bias, bn_mean, bn_var) of output
        columns [lo, hi), contiguous float32."""
        hi = self.cout if hi is None else hi
        f = lambda t: None if t is None else t.detach()[lo:hi].float().contiguous()
        bn = self.bn
        return (f(self.weight), f(self.bias),
                f(bn.weight) if bn is not None else None,
                f(bn.bias) if bn is not None else None,
                f(bn.running_mean) if bn is not None else None,
                f(bn.running_var) if bn is not None else None)


def mlp_layers(seq):
    """LayerSpecs of an ``MLP(...)`` module tree."""
    specs = []
    for block in seq:
        linear, fbn, act = block[0], block[1], block[2]
        if not isinstance(act, nn.LeakyReLU):
            raise NotImplementedError(f"mvfusion MLP kernels implement LeakyReLU, got {type(act)}")
        specs.append(LayerSpec(linear, fbn.batch_norm, float(act.negative_slope)))
    return specs


def linear_layer(linear):
    return LayerSpec(linear)


def _check_eval(spec):
    if spec.bn is not None and spec.bn.training:
        raise RuntimeError(
            "mvfusion pools run BatchNorm with running statistics only: call .eval() "
            "(train-mode batch statistics and backward are not implemented)")


def _launch(src, specs, cols, out, ldo_tensor, dot=None, pool=None):
    """One mvf_mlp_forward launch.  src = (xa, ga, xb, gb, mode); specs = list
    of (LayerSpec, lo, hi); out = tensor view whose first element is the
    output's first column, ldo_tensor = row stride in floats.  dot = (queries,
    row2seg, divisor): mvf_mlp_forward_dot, `out` is the [rows] dot output.
    pool = (att, row2seg, ptr): mvf_mlp_forward_pool, `out` is the prepared
    [nseg, cout] pool output."""
    lib = L.load()
    xa, ga, xb, gb, mode = src
    keep = []
    arr = (L.MvfMlpLayer * len(specs))()
    for i, (sp, lo, hi) in enumerate(specs):
        _check_eval(sp)
        w, b, bw, bb, bm, bv = sp.tensors(lo, hi)
        keep += [w, b, bw, bb, bm, bv]
        a = arr[i]
        a.weight, a.bias = w.data_ptr(), (b.data_ptr() if b is not None else None)
        a.bn_weight = bw.data_ptr() if bw is not None else None
        a.bn_bias = bb.data_ptr() if bb is not None else None
        a.bn_mean = bm.data_ptr() if bm is not None else None
        a.bn_var = bv.data_ptr() if bv is not None else None
        a.bn_eps = float(sp.bn.eps) if sp.bn is not None else 0.0
        a.slope = float(sp.slope) if sp.slope is not None else 0.0
        a.cin, a.cout, a.act = sp.cin, hi - lo, int(sp.slope is not None)
    rows = xa.shape[0] if ga is None else ga.shape[0]
    dev = xa.device
    with torch.cuda.device(dev):
        if pool is not None:
            att, seg, ptr = pool
            # per-tile partial sums of the segments that cross a 32-row tile
            ws = L.workspace(lib.mvf_mlp_forward_pool_workspace_bytes(rows, specs[-1][2] - specs[-1][1]), dev)
            L.check(lib.mvf_mlp_forward_pool(
                L.ptr(xa), xa.stride(0), xa.shape[1], L.ptr(ga),
                L.ptr(xb), 0 if xb is None else xb.stride(0), 0 if xb is None else xb.shape[1],
                L.ptr(gb), rows, arr, len(specs), L.ptr(att), att.shape[1], L.ptr(seg),
                L.ptr(ptr), ptr.shape[0] - 1, L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr(dev)))
        elif dot is None:
            L.check(lib.mvf_mlp_forward(
                L.ptr(xa), xa.stride(0), xa.shape[1], L.ptr(ga),
                L.ptr(xb), 0 if xb is None else xb.stride(0), 0 if xb is None else xb.shape[1],
                L.ptr(gb), mode, rows, arr, len(specs), L.ptr(out), ldo_tensor,
                L.stream_ptr(dev)))
        else:
            q, seg, div = dot
            L.check(lib.mvf_mlp_forward_dot(
                L.ptr(xa), xa.stride(0), xa.shape[1], L.ptr(ga),
                L.ptr(xb), 0 if xb is None else xb.stride(0), 0 if xb is None else xb.shape[1],
                L.ptr(gb), mode, rows, arr, len(specs), L.ptr(q), q.stride(0), L.ptr(seg),
                float(div), L.ptr(out), L.stream_ptr(dev)))
    del keep


def aligned_input(specs, x):
    """(specs, x) with the same product specs(x), on input rows the 16-byte
    kernels accept.  Rows of at most 64 float32 columns whose width is no
    multiple of 8 (or whose storage is not 16-byte aligned) are copied into
    zero-padded rows, and the first layer's weights get zero columns to match
    (the added products are exact zeros): the aligned kernels, and their pool
    epilogues, run several times faster than the general one."""
    if (not specs or not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float32
            or x.dim() != 2 or x.shape[0] == 0 or x.shape[1] > _MAX_C
            or x.shape[1] != specs[0].cin):
        return specs, x
    c = x.shape[1]
    if c % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
        return specs, x
    cp = (c + 7) // 8 * 8
    xp = x.new_zeros((x.shape[0], cp))
    xp[:, :c] = x.detach()
    first = copy.copy(specs[0])
    first.weight = torch.nn.functional.pad(specs[0].weight.detach().float(), (0, cp - c))
    first.cin = cp
    return [first] + list(specs[1:]), xp


def dot_fusable(specs, queries):
    """Can run_layers(..., dot=...) fold the last layer into its dot epilogue?"""
    return (len(specs) >= 1 and specs[-1].cout <= 32 and specs[-1].cout % 4 == 0
            and specs[-1].cin <= _MAX_C and queries.shape[1] == specs[-1].cout
            and queries.stride(1) == 1 and queries.stride(0) % 4 == 0 and queries.data_ptr() % 16 == 0)


def run_layers(specs, xa, gather_a=None, xb=None, gather_b=None, mode=0, dot=None):
    """y = specs[-1](... specs[0](input)) with input rows
    [xa[ga] | xb[gb]] (mode 0), xa + xb (mode 1) or [xa | xa + xb] (mode 2).
    Layers are packed into as few kernel launches as the width limits allow;
    a layer wider than 64 outputs runs alone, 64 output columns per launch.

    dot = (queries [nq, cout], row2seg int32 [rows], divisor): y itself is not
    returned; the result is [rows, 1] with sum_c y[r, c] * queries[row2seg[r], c]
    (/ divisor when it is not 0), computed in the epilogue of the last launch
    (needs dot_fusable)."""
    def prep(t):
        if t is None:
            return None
        if not t.is_cuda:
            raise RuntimeError("mvfusion dense kernels run on a HIP device only "
                               "(there is no CPU fallback)")
        t = t.detach()
        if t.dtype != torch.float32:
            raise TypeError(f"mvfusion dense kernels expect float32, got {t.dtype}")
        if t.dim() != 2:
            raise ValueError("mvfusion dense kernels expect 2D [rows, channels] tensors")
        return t if t.stride(1) == 1 else t.contiguous()

    xa, xb = prep(xa), prep(xb)
    if gather_a is None and xb is None and mode == 0:
        specs, xa = aligned_input(specs, xa)
    for g in (gather_a, gather_b):
        if g is not None and (g.dtype != torch.int32 or not g.is_cuda):
            raise TypeError("row gathers must be int32 tensors on the HIP device")
    rows = xa.shape[0] if gather_a is None else gather_a.shape[0]
    src = (xa, gather_a, xb, gather_b, mode)
    dev = xa.device
    i = 0
    x = None
    while i < len(specs):
        sp = specs[i]
        if sp.cout > _MAX_C:
            # a wide layer alone, 64 output columns per launch
            out = torch.empty((rows, sp.cout), dtype=torch.float32, device=dev)
            for lo in range(0, sp.cout, _MAX_C):
                hi = min(sp.cout, lo + _MAX_C)
                _launch(src, [(sp, lo, hi)], None, out[:, lo:], sp.cout)
            i += 1
        else:
            group = [(sp, 0, sp.cout)]
            j = i + 1
            while (j < len(specs) and len(group) < _MAX_LAYERS and specs[j].cout <= _MAX_C
                   and specs[j].cin <= _MAX_C):
                group.append((specs[j], 0, specs[j].cout))
                j += 1
            cout = group[-1][0].cout
            if dot is not None and j == len(specs):
                q, seg, div = dot
                q = prep(q)
                if seg.dtype != torch.int32 or not seg.is_cuda or seg.shape[0] != rows:
                    raise TypeError("row2seg must be an int32 device tensor with one entry per row")
                out = torch.empty((rows, 1), dtype=torch.float32, device=dev)
                _launch(src, group, None, out, 1, dot=(q, seg, div))
            else:
                out = torch.empty((rows, cout), dtype=torch.float32, device=dev)
                _launch(src, group, None, out, cout)
            i = j
        x = out
        src = (x, None, None, None, 0)
    return x


def pool_fusable(specs, x, num_groups):
    """Can pooled_layers() run these layers with the attention-weighted segment
    sum as their epilogue?  Two layers of one output width <= 64 (a multiple of
    4) on an aligned float32 input; at most 8 attention groups of equally many channels,
    a multiple of 4."""
    return (len(specs) == 2 and specs[0].cout == specs[1].cout and num_groups <= 8
            and specs[1].cout % num_groups == 0 and (specs[1].cout // num_groups) % 4 == 0
            and specs[1].cout <= _MAX_C and specs[1].cout % 4 == 0
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
            and x.shape[1] % 8 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and x.shape[0] > 0)


def pooled_layers(specs, x, att, row2seg, ptr, out):
    """out[s] = sum_{r in s} specs(x)[r] * att[r]: the layer chain with the pool
    epilogue (needs pool_fusable; `att` [rows, G] and `out` as returned by
    attention_pool(<number of channels>, ...): that `att` carries the gate).
    ptr [nseg + 1] int64: the CSR pointers row2seg was made from."""
    _launch((x.detach(), None, None, None, 0), [(sp, 0, sp.cout) for sp in specs], None, out,
            specs[-1].cout, pool=(att, row2seg, ptr))
    return out


def layers_with_segment_max(specs, x, ptr, row2seg):
    """(y, segment max of y) with y = specs(x): the layer chain stores its rows and
    folds their per-segment maximum in the same launch (needs pool_fusable(specs,
    x, 1)); equals (run_layers(specs, x), segment_csr(y, ptr, 'max')) bit for bit."""
    lib = L.load()
    x = x.detach()
    rows, cout = x.shape[0], specs[-1].cout
    nseg = ptr.shape[0] - 1
    keep = []
    arr = (L.MvfMlpLayer * len(specs))()
    for i, sp in enumerate(specs):
        _check_eval(sp)
        w, b, bw, bb, bm, bv = sp.tensors()
        keep += [w, b, bw, bb, bm, bv]
        a = arr[i]
        a.weight, a.bias = w.data_ptr(), (b.data_ptr() if b is not None else None)
        a.bn_weight = bw.data_ptr() if bw is not None else None
        a.bn_bias = bb.data_ptr() if bb is not None else None
        a.bn_mean = bm.data_ptr() if bm is not None else None
        a.bn_var = bv.data_ptr() if bv is not None else None
        a.bn_eps = float(sp.bn.eps) if sp.bn is not None else 0.0
        a.slope = float(sp.slope) if sp.slope is not None else 0.0
        a.cin, a.cout, a.act = sp.cin, sp.cout, int(sp.slope is not None)
    out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    seg = torch.empty((nseg, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.mvf_mlp_forward_segmax(
            L.ptr(x), x.stride(0), x.shape[1], None, 0, 0, rows, arr, len(specs),
            L.ptr(out), cout, L.ptr(ptr), nseg, L.ptr(row2seg), L.ptr(seg), L.stream_ptr(x.device)))
    del keep
    return out, seg


def csr_row_segments(ptr, rows, outside=None):
    """int32 [rows]: segment index of every row of a CSR pointer array.  Rows
    that belong to no segment (before ptr[0], from ptr[-1] on) are left
    unwritten, or set to `outside` when it is given."""
    lib = L.load()
    if outside is None:
        out = torch.empty(rows, dtype=torch.int32, device=ptr.device)
    else:
        out = torch.full((rows,), int(outside), dtype=torch.int32, device=ptr.device)
    if rows > 0:
        with torch.cuda.device(ptr.device):
            L.check(lib.mvf_csr_row_segments(
                L.ptr(ptr), ptr.shape[0] - 1, L.ptr(out), L.stream_ptr(ptr.device)))
    return out


def attention_pool(x_mod, ptr, num_groups, chan2group, keys=None, queries=None,
                   row2seg=None, comp=None, nc_qk=1, dim_scaling=False,
                   group_scaling=False, gate_w=None, gate_b=None, gating=False, eps=1e-12):
    """Returns (x_pool, comp, att, gate).  Either ``keys``/``queries`` or
    ``comp`` (the compatibilities) is given; ``row2seg`` maps key rows to query
    rows (None: one query row per key row).  ``x_mod`` may be the number of
    value channels instead of the values: the scores are computed and x_pool
    comes back prepared for pooled_layers() (``att`` then holds att * gate)."""
    lib = L.load()
    if isinstance(x_mod, int):
        src = keys if keys is not None else comp
        dev, rows, Cm, x_mod = src.device, src.shape[0], x_mod, None
    else:
        dev = x_mod.device
        rows, Cm = x_mod.shape
        x_mod = x_mod.detach().contiguous()
    nseg = ptr.shape[0] - 1
    G = int(num_groups)
    if keys is None:
        comp = comp.detach().float().contiguous().view(rows, G)
    else:
        comp = torch.empty((rows, G), dtype=torch.float32, device=dev)
    att = torch.empty((rows, G), dtype=torch.float32, device=dev)
    gate = torch.empty((nseg, G), dtype=torch.float32, device=dev) if gating else None
    out = torch.empty((nseg, Cm), dtype=torch.float32, device=dev)
    f = lambda t: None if t is None else t.detach().float().contiguous().view(-1)
    gw, gb = f(gate_w), f(gate_b)
    if nseg > 0:
        with torch.cuda.device(dev):
            L.check(lib.mvf_attention_pool(
                L.ptr(keys), 0 if keys is None else keys.stride(0),
                L.ptr(queries), 0 if queries is None else queries.stride(0), L.ptr(row2seg),
                L.ptr(x_mod), Cm, L.ptr(ptr), nseg, rows, G, int(nc_qk),
                int(bool(dim_scaling)), int(bool(group_scaling)), float(eps),
                L.ptr(chan2group), L.ptr(gw), L.ptr(gb), int(bool(gating)),
                L.ptr(comp), L.ptr(att), L.ptr(gate), L.ptr(out), L.stream_ptr(dev)))
    return out, comp, att, gate


def _segment_pick(x_mod, feat, ptr, pick_max, want_rows=False):
    lib = L.load()
    dev = x_mod.device
    x_mod = x_mod.detach().float().contiguous()
    feat = feat.detach().float()
    nseg = ptr.shape[0] - 1
    out = torch.empty((nseg, x_mod.shape[1]), dtype=torch.float32, device=dev)
    picked = torch.empty(nseg, dtype=torch.int64, device=dev) if want_rows else None
    if nseg > 0:
        with torch.cuda.device(dev):
            L.check(lib.mvf_segment_pick(
                L.ptr(x_mod), x_mod.shape[1], L.ptr(feat), feat.stride(0), L.ptr(ptr), nseg,
                int(bool(pick_max)), L.ptr(out), L.ptr(picked), L.stream_ptr(dev)))
    return out, picked


class _SegmentPick(torch.autograd.Function):
    """Row selection per segment; the gradient goes to the selected rows."""

    @staticmethod
    def forward(ctx, x_mod, feat, ptr, pick_max):
        out, picked = _segment_pick(x_mod, feat, ptr, pick_max, want_rows=True)
        ctx.save_for_backward(picked)
        ctx.shape = x_mod.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        picked, = ctx.saved_tensors
        grad = torch.zeros(ctx.shape, dtype=grad_out.dtype, device=grad_out.device)
        ok = picked >= 0
        grad[picked[ok]] = grad_out[ok]          # one row per segment: no duplicates
        return grad, None, None, None


def segment_pick(x_mod, feat, ptr, pick_max):
    """x_mod rows of the largest / smallest ``feat`` per segment; differentiable
    with respect to x_mod."""
    if torch.is_grad_enabled() and x_mod.requires_grad:
        return _SegmentPick.apply(x_mod, feat, ptr, pick_max)
    return _segment_pick(x_mod, feat, ptr, pick_max)[0]
