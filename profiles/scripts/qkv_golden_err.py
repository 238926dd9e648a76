"""Error levels of the attentive pool on the reference fixtures, per arithmetic mode, against
the fixture (the reference's float32 result) and against a float64 evaluation of the module."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
from mvfusion import _lib, pooling  # noqa: E402

lib = _lib.load()
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def levels(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = np.abs(want).max()
    err = np.abs(got - want)
    big = np.abs(want) > 0.1 * scale
    return float((err[big] / np.abs(want[big])).max()), float(err.max() / scale), int(big.sum())


for name in ("qkv_default", "qkv_groups4", "qkv_nogate"):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False))
    mod = pooling.QKVBimodalCSRPool(
        in_main=g["x_main"].shape[1], in_map=g["x_map"].shape[1], in_mod=g["x_mod"].shape[1],
        out_mod=g["x_mod"].shape[1], num_groups=int(g["num_groups"]), nc_qk=int(g["nc_qk"]),
        gating=bool(g["gating"]), dim_scaling=bool(g["dim_scaling"]),
        group_scaling=bool(g["group_scaling"]), use_num=bool(g["use_num"]), save_last=False,
        map_encoder="DeepSetFeat")
    mod.load_state_dict({k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")})
    mod = mod.eval()
    # float64 evaluation with torch modules on the host (the composed path in double)
    import copy
    from mvfusion import training
    m64 = copy.deepcopy(mod).double()
    args = (T(g["x_main"]), T(g["x_mod"]), T(g["x_map"]), T(g["ptr"]))
    mod = mod.to(DEV)
    for mode in (1, 0):
        lib.mvf_set_mlp_precision(mode)
        with torch.no_grad():
            out = mod(*args).cpu().numpy()
        print(name, "mode", mode, "vs fixture (rel on big, abs/scale, n big):", levels(out, g["x_pool"]), flush=True)
    lib.mvf_set_mlp_precision(1)
