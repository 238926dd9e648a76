"""The attentive pools of mvf_oracle.py evaluated in float64 numpy.

TEST INFRASTRUCTURE ONLY, and not the reference's arithmetic: the reference, the
oracle and the HIP kernels all compute in float32, each with its own summation
order.  This file is the yardstick that tells a badly conditioned case (every
float32 evaluation, the oracle's included, is far from the float64 value) from
a kernel that is less accurate than float32 allows.  Same formulas and
citations as mvf_oracle.qkv_pool / group_pool (DeepSetFeat map encoder).
"""
import math

import numpy as np

F = np.float64
PARTS = {}      # intermediate values of the last call: C (scores), A (attention), G (gate), x_mod


def _reduceat(x, ptr, ufunc):
    """Per-segment reduction of the rows of x; empty segments give 0."""
    out = np.zeros((ptr.shape[0] - 1, x.shape[1]), F)
    full = ptr[1:] > ptr[:-1]
    if full.any():
        out[full] = ufunc.reduceat(x, ptr[:-1][full], axis=0)
    return out


def segment_csr(x, ptr, reduce):
    x = np.asarray(x, F).reshape(x.shape[0], -1)
    if reduce == "max":
        return _reduceat(x, ptr, np.maximum)
    if reduce == "min":
        return _reduceat(x, ptr, np.minimum)
    s = _reduceat(x, ptr, np.add)
    if reduce == "mean":
        s = s / np.maximum(ptr[1:] - ptr[:-1], 1).astype(F)[:, None]
    return s


def gather_csr(x, ptr):
    return np.repeat(x, ptr[1:] - ptr[:-1], axis=0)


def segment_softmax_csr(x, ptr, eps=1e-12, scaling=False):
    c = x - gather_csr(segment_csr(x, ptr, "max"), ptr)
    if scaling:
        c = c / gather_csr(np.sqrt((ptr[1:] - ptr[:-1]).astype(F)), ptr)[:, None]
    e = np.exp(c)
    return e / (gather_csr(segment_csr(e, ptr, "sum"), ptr) + eps)


def linear(x, weight, bias=None):
    y = np.asarray(x, F) @ np.asarray(weight, F).T
    return y if bias is None else y + np.asarray(bias, F)


def mlp(x, sd, prefix, n_layers=2, eps=1e-5, slope=0.2):
    for i in range(n_layers):
        bn = f"{prefix}.{i}.1.batch_norm."
        y = linear(x, sd[f"{prefix}.{i}.0.weight"], sd.get(f"{prefix}.{i}.0.bias"))
        # constants as the reference's modules hold them (Python floats), as in module.double()
        inv = 1.0 / np.sqrt(np.asarray(sd[bn + "running_var"], F) + eps)
        y = (y - np.asarray(sd[bn + "running_mean"], F)) * inv * np.asarray(sd[bn + "weight"], F) \
            + np.asarray(sd[bn + "bias"], F)
        x = np.where(y < 0, y * slope, y)
    return x


def deep_set_feat(x, ptr, sd, prefix, pool, fusion, use_num):
    x = mlp(x, sd, prefix + ".mlp_elt_1")
    x_set = np.concatenate([segment_csr(x, ptr, p) for p in pool], axis=-1)
    if use_num:
        num = np.sqrt(1.0 / ((ptr[1:] - ptr[:-1]).astype(F) + 1e-3))
        x_set = np.concatenate([x_set, num.reshape(-1, 1)], axis=1)
    x_set = gather_csr(mlp(x_set, sd, prefix + ".mlp_set"), ptr)
    if fusion == "residual":
        x_out = x + x_set
    elif fusion == "concatenation":
        x_out = np.concatenate([x, x_set], axis=-1)
    else:
        x_out = np.concatenate([x, x + x_set], axis=-1)
    return mlp(x_out, sd, prefix + ".mlp_elt_2")


def _expand(A, num_groups, num_channels):
    if num_groups == 1:
        return A.reshape(-1, 1)
    if num_groups < num_channels:
        sizes = np.full(num_groups, num_channels // num_groups, np.int64)
        sizes += np.arange(num_groups) < num_channels - sizes.sum()
        return np.repeat(A, sizes, axis=1)
    return A


def _attend(xv, comp, ptr, sd, num_groups, out_mod, gating, group_scaling):
    att = segment_softmax_csr(comp, ptr, scaling=group_scaling)
    x_pool = segment_csr(xv * _expand(att, num_groups, out_mod), ptr, "sum")
    gate = None
    if gating:
        g = segment_csr(comp, ptr, "max") * np.asarray(sd["G.weight"], F) + np.asarray(sd["G.bias"], F)
        gate = np.tanh(np.maximum(g, 0.0))
        x_pool = x_pool * _expand(gate, num_groups, out_mod)
    PARTS.clear()
    PARTS.update(x_pool=x_pool, C=comp, A=att, G=gate, x_mod=xv)
    return x_pool


def qkv_pool(x_main, x_mod, x_map, ptr, sd, num_groups=1, nc_qk=8, gating=True, dim_scaling=True,
             group_scaling=False, use_num=False, pool=("max",), fusion="concatenation",
             use_mod_q=False, use_mod_k=False):
    ptr = np.asarray(ptr, np.int64)
    out_mod = sd["E_mod.1.0.weight"].shape[0]
    xm = mlp(np.asarray(x_main, F), sd, "E_main")
    xp = deep_set_feat(np.asarray(x_map, F), ptr, sd, "E_map", pool, fusion, use_num)
    xv = mlp(np.asarray(x_mod, F), sd, "E_mod")
    k_in = mlp(np.concatenate([xp, xv], axis=1), sd, "E_mix_K") if use_mod_k else xp
    keys = linear(k_in, sd["K.weight"], sd["K.bias"])
    if use_mod_q:
        queries = linear(mlp(np.concatenate([gather_csr(xm, ptr), xv], axis=1), sd, "E_mix_Q"),
                         sd["Q.weight"], sd["Q.bias"])
    else:
        queries = gather_csr(linear(xm, sd["Q.weight"], sd["Q.bias"]), ptr)
    V = keys.shape[0]
    comp = (keys.reshape(V, num_groups, nc_qk) * queries.reshape(V, num_groups, nc_qk)).sum(axis=2)
    if dim_scaling:
        comp = comp / math.sqrt(nc_qk)
    return _attend(xv, comp, ptr, sd, num_groups, out_mod, gating, group_scaling)


def group_pool(x_mod, x_map, ptr, sd, num_groups=1, gating=True, group_scaling=True, use_num=False,
               pool=("max",), fusion="concatenation", use_mod=False):
    ptr = np.asarray(ptr, np.int64)
    out_mod = sd["E_mod.1.0.weight"].shape[0]
    xp = deep_set_feat(np.asarray(x_map, F), ptr, sd, "E_map", pool, fusion, use_num)
    xv = mlp(np.asarray(x_mod, F), sd, "E_mod")
    if use_mod:
        xp = mlp(np.concatenate([xp, xv], axis=1), sd, "E_mix")
    comp = linear(xp, sd["E_score.weight"], sd["E_score.bias"])
    return _attend(xv, comp, ptr, sd, num_groups, out_mod, gating, group_scaling)
