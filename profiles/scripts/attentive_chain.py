"""Cumulative distance from float64, layer group by layer group, of the map-encoder / score chain of
a GroupBimodalCSRPool case of fuzz_attentive.py: HIP layer chains next to the float32 oracle.
usage (GPU box, repo root): python profiles/scripts/attentive_chain.py <seed>"""
import sys

sys.path.insert(0, "profiles/scripts")
import numpy as np
import torch

import fuzz_attentive as FZ
import mvf_exact64 as x64
import mvf_oracle as orc
from mvfusion import dense as _dense, pooling
from mvfusion.pooling import segment_csr

seed = int(sys.argv[1])
cap = {}
_group = orc.group_pool
orc.group_pool = lambda *a, **k: (cap.update(a=a, k=k), _group(*a, **k))[1]
mods = []
_to = torch.nn.Module.to
torch.nn.Module.to = lambda self, *a, **k: (mods.append(self) if isinstance(self, pooling.GroupBimodalCSRPool) else None,
                                            _to(self, *a, **k))[1]
try:
    print(FZ.one_case(seed)[:300])
except AssertionError as e:
    print("MISS", str(e)[:300])
torch.nn.Module.to = _to
(x_mod, x_map, ptr, sd), k = cap["a"], cap["k"]
mod = mods[-1]
pool, fusion, use_num, use_mod = k["pool"], k["fusion"], k["use_num"], k["use_mod"]
T = FZ.T
tp = T(ptr)


def show(name, hip, o32, ex):
    hip = hip.cpu().numpy().astype(np.float64)
    sc = np.abs(ex).max()
    print(f"  {name:22s} HIP {np.abs(hip - ex).max() / sc:.1e} (rms {np.sqrt(((hip - ex) ** 2).mean()) / sc:.1e})   "
          f"oracle {np.abs(o32 - ex).max() / sc:.1e} (rms {np.sqrt(((o32 - ex) ** 2).mean()) / sc:.1e})   scale {sc:.2e}")


with torch.no_grad():
    E = mod.E_map
    # stage 1: mlp_elt_1
    h1 = _dense.run_layers(_dense.mlp_layers(E.mlp_elt_1), T(x_map))
    o1 = orc.mlp(x_map, sd, "E_map.mlp_elt_1")
    e1 = x64.mlp(np.asarray(x_map, np.float64), sd, "E_map.mlp_elt_1")
    show("E_map.mlp_elt_1", h1, o1, e1)
    # stage 2: set pools + mlp_set
    hs = [segment_csr(h1, tp, reduce=p) for p in pool]
    os_ = [orc.segment_csr(o1, ptr, p) for p in pool]
    es = [x64.segment_csr(e1, ptr, p) for p in pool]
    if use_num:
        cnt = (ptr[1:] - ptr[:-1])
        num32 = np.sqrt(np.float32(1) / (cnt.astype(np.float32) + np.float32(1e-3))).astype(np.float32)
        hs.append(T(num32).view(-1, 1)); os_.append(num32.reshape(-1, 1))
        es.append(np.sqrt(1.0 / (cnt.astype(np.float64) + 1e-3)).reshape(-1, 1))
    h2 = _dense.run_layers(_dense.mlp_layers(E.mlp_set), torch.cat(hs, 1).contiguous())
    o2 = orc.mlp(np.concatenate(os_, 1), sd, "E_map.mlp_set")
    e2 = x64.mlp(np.concatenate(es, 1), sd, "E_map.mlp_set")
    show("pools + mlp_set", h2, o2, e2)
    # stage 3: the whole encoder
    row2seg = _dense.csr_row_segments(tp, x_map.shape[0], outside=0)
    h3 = E.embed(T(x_map), tp, row2seg)
    o3 = orc.deep_set_feat(x_map, ptr, sd, "E_map", pool, fusion, use_num)
    e3 = x64.deep_set_feat(np.asarray(x_map, np.float64), ptr, sd, "E_map", pool, fusion, use_num)
    show("E_map (whole)", h3, o3, e3)
    hv = _dense.run_layers(_dense.mlp_layers(mod.E_mod), T(x_mod))
    ov = orc.mlp(x_mod, sd, "E_mod")
    ev = x64.mlp(np.asarray(x_mod, np.float64), sd, "E_mod")
    show("E_mod", hv, ov, ev)
    score = [_dense.linear_layer(mod.E_score)]
    if use_mod:
        hm = _dense.run_layers(_dense.mlp_layers(mod.E_mix), h3, None, hv)
        om = orc.mlp(np.concatenate([o3, ov], 1), sd, "E_mix")
        em = x64.mlp(np.concatenate([e3, ev], 1), sd, "E_mix")
        show("E_mix", hm, om, em)
        hc = _dense.run_layers(_dense.mlp_layers(mod.E_mix) + score, h3, None, hv)
    else:
        hm, om, em = h3, o3, e3
        hc = _dense.run_layers(score, h3)
    oc = orc.linear(om, sd["E_score.weight"], sd["E_score.bias"])
    ec = x64.linear(em, sd["E_score.weight"], sd["E_score.bias"])
    show("scores", hc, oc, ec)
    # the same layers fed the float64 inputs rounded to float32: the error one stage adds
    hc1 = _dense.run_layers(score, T(em.astype(np.float32)))
    oc1 = orc.linear(em.astype(np.float32), sd["E_score.weight"], sd["E_score.bias"])
    show("scores from exact in", hc1, oc1, ec)
print("E_score weight", np.abs(sd["E_score.weight"]).max(), "bias", sd["E_score.bias"], "rows", x_map.shape[0])
