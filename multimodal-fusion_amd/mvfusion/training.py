"""Differentiable forward of the attentive pools and their set encoders.

Calls that need no autograd graph (``.eval()`` under ``torch.no_grad()``, or
nothing requiring a gradient) run through the fused MFMA layer-chain and
attention kernels (``dense.py``), which implement BatchNorm with running
statistics and record no graph.  In ``.train()``, and in ``.eval()`` whenever an
input or a parameter requires a gradient, the pools are composed here from
differentiable pieces instead (BatchNorm follows the module's mode): the dense layers (Linear, BatchNorm1d with batch
statistics, LeakyReLU) are torch modules on the HIP device, the CSR operations
are this package's kernels wrapped in ``torch.autograd.Function``
(``pooling.segment_csr`` / ``gather_csr`` / ``segment_softmax_csr``).  The
computation follows modules/multimodal/pooling.py: ``DeepSetFeat.forward``
:658-681, ``MinMaxDiffSetFeat.forward`` :577-597, ``MLPSetFeat.forward``
:688-699, ``GroupBimodalCSRPool.forward`` :263-315 and
``QKVBimodalCSRPool.forward`` :454-547.
"""
import math

import torch

from . import pooling as P


def mlp(seq, x):
    """MLP of base_modules.py:38-48: per block Linear, BatchNorm1d, activation."""
    for block in seq:
        x = block[2](block[1].batch_norm(block[0](x)))
    return x


def deep_set_feat(m, x, ptr):
    x = mlp(m.mlp_elt_1, x)
    sets = [P.segment_csr(x, ptr, reduce=p) for p in m.pool]
    if m.use_num:
        sets.append(P._set_num(ptr).view(-1, 1))
    x_set = torch.cat(sets, dim=-1) if len(sets) > 1 else sets[0]
    x_set = P.gather_csr(mlp(m.mlp_set, x_set), ptr)
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


def attend(m, x_mod, comp, ptr):
    """Softmax of the compatibilities over the views of a point, attention-
    weighted sum of the values, optional gate from the best compatibility."""
    att = P.segment_softmax_csr(comp, ptr, scaling=m.group_scaling)
    x_pool = P.segment_csr(x_mod * P.expand_group_feat(att, m.num_groups, m.out_mod), ptr,
                           reduce='sum')
    gate = None
    if m.G is not None:
        gate = m.G(P.segment_csr(comp, ptr, reduce='max'))
        x_pool = x_pool * P.expand_group_feat(gate, m.num_groups, m.out_mod)
    return x_pool, att, gate


def group_pool(m, x_main, x_mod, x_map, ptr):
    x_map = map_encoder(m.E_map, x_map, ptr)
    x_mod = mlp(m.E_mod, x_mod)
    x_mix = mlp(m.E_mix, torch.cat([x_map, x_mod], dim=1)) if m.use_mod else x_map
    comp = m.E_score(x_mix)
    x_pool, att, gate = attend(m, x_mod, comp, ptr)
    return x_pool, dict(x_map=x_map, x_mod=x_mod, C=comp, A=att, G=gate)


def qkv_pool(m, x_main, x_mod, x_map, ptr):
    x_main = mlp(m.E_main, x_main)
    x_map = map_encoder(m.E_map, x_map, ptr)
    x_mod = mlp(m.E_mod, x_mod)
    if m.use_mod_k:
        keys = m.K(mlp(m.E_mix_K, torch.cat([x_map, x_mod], dim=1)))
    else:
        keys = m.K(x_map)
    if m.use_mod_q:
        queries = m.Q(mlp(m.E_mix_Q, torch.cat([P.gather_csr(x_main, ptr), x_mod], dim=1)))
    else:
        queries = P.gather_csr(m.Q(x_main), ptr)
    comp = (keys.view(-1, m.num_groups, m.nc_qk) * queries.view(-1, m.num_groups, m.nc_qk)).sum(dim=2)
    if m.dim_scaling:
        comp = comp / math.sqrt(m.nc_qk)
    x_pool, att, gate = attend(m, x_mod, comp, ptr)
    return x_pool, dict(x_map=x_map, x_mod=x_mod, K=keys, Q=queries, C=comp, A=att, G=gate)
