"""Where a case of fuzz_attentive.py loses accuracy: distance from the float64 evaluation of the
scores, attention weights, gate and pooled output, for the fused kernels (with and without the
saved intermediates) and the oracle.
usage (GPU box, repo root): python profiles/scripts/attentive_stages.py <seed>"""
import sys

sys.path.insert(0, "profiles/scripts")
import numpy as np
import torch

import fuzz_attentive as FZ
import mvf_exact64 as x64
import mvf_oracle as orc
from mvfusion import pooling

seed = int(sys.argv[1])
captured = {}
_group, _qkv = orc.group_pool, orc.qkv_pool


def spy(fn, name):
    def run(*a, **k):
        captured["args"], captured["kw"], captured["fn"] = a, k, name
        return fn(*a, **k)
    return run


orc.group_pool, orc.qkv_pool = spy(_group, "group_pool"), spy(_qkv, "qkv_pool")
mods = []
_to = torch.nn.Module.to


def keep(self, *a, **k):
    if isinstance(self, (pooling.QKVBimodalCSRPool, pooling.GroupBimodalCSRPool)):
        mods.append(self)
    return _to(self, *a, **k)


torch.nn.Module.to = keep
try:
    print(FZ.one_case(seed))
except AssertionError as e:
    print("MISS", str(e)[:600])
torch.nn.Module.to = _to
a, k = captured["args"], captured["kw"]
want = (_group if captured["fn"] == "group_pool" else _qkv)(*a, **k, return_parts=True)
getattr(x64, captured["fn"])(*a, **k)
exact = dict(x64.PARTS)
mod = mods[-1]
mod.save_last = True
is_qkv = captured["fn"] == "qkv_pool"
T = FZ.T
with torch.no_grad():
    if is_qkv:
        got = mod(T(a[0]), T(a[1]), T(a[2]), T(a[3]))
    else:
        got = mod(None, T(a[0]), T(a[1]), T(a[2]))


def far(x, ref, frac=0.1):
    x = np.asarray(x.detach().cpu().numpy() if torch.is_tensor(x) else x, np.float64).reshape(ref.shape)
    big = np.abs(ref) > frac * np.abs(ref).max()
    return (f"rel(big) {float((np.abs(x - ref)[big] / np.abs(ref[big])).max()):.1e}  "
            f"abs/scale {float(np.abs(x - ref).max() / np.abs(ref).max()):.1e}")


print("stage: fused (saved intermediates) | oracle  -- distance from float64")
for key, attr in (("C", "_last_C"), ("A", "_last_A"), ("G", "_last_G"), ("x_pool", None)):
    if exact[key] is None:
        continue
    f = got if attr is None else getattr(mod, attr)
    print(f"  {key:7s} fused {far(f, exact[key])} | oracle {far(want[key], exact[key])}")
g = exact["G"]
if g is not None:
    print("gate values: min %.3e median %.3e max %.3e" % (np.abs(g).min(), np.median(np.abs(g)), np.abs(g).max()))
# the pool itself: the oracle's arithmetic on the fused path's own intermediates
xm = mod._last_x_mod.cpu().numpy()
A = mod._last_A.cpu().numpy().reshape(xm.shape[0], -1)
ptr_np = np.asarray(a[3] if is_qkv else a[2], np.int64)
G_ = mod.num_groups
S = orc.segment_csr(xm * orc.expand_group_feat(A, G_, xm.shape[1]), ptr_np, "sum")
if exact["G"] is not None:
    S = S * orc.expand_group_feat(mod._last_G.cpu().numpy().reshape(S.shape[0], -1), G_, xm.shape[1])
print("  pool of the fused intermediates in the oracle's arithmetic:", far(S, exact["x_pool"]),
      "| fused pool against it:", far(got, S.astype(np.float64)))
print("  x_mod fused", far(mod._last_x_mod, exact["x_mod"]), "| oracle", far(want["x_mod"], exact["x_mod"]) if "x_mod" in want else "")
