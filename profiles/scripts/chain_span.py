"""Start / finish times of the workgroups of one layer-chain launch (needs a library built with
-DMVF_CHAIN_STAMP, MVFUSION_SO=<path>): the key chain of the c3 workload (64 -> 32 -> 32 -> 16 with
the dot epilogue on 2 M rows).
usage (GPU box, repo root): MVFUSION_SO=... python profiles/scripts/chain_span.py"""
import ctypes as C
import sys

sys.path.insert(0, "multimodal-fusion_amd")
import numpy as np
import torch

from mvfusion import _lib, dense, pooling

dev = "cuda:0"
lib = _lib.load()
rows = 2_002_000
torch.manual_seed(0)
x = torch.rand((rows, 64), device=dev)
mlp = pooling.MLP([64, 32, 32], bias=False).to(dev).eval()
K = torch.nn.Linear(32, 16).to(dev)
specs = dense.mlp_layers(mlp) + [dense.linear_layer(K)]
q = torch.rand((rows // 4 + 1, 16), device=dev)
seg = (torch.arange(rows, device=dev) // 4).int()
for _ in range(3):
    out = dense.run_layers(specs, x, dot=(q, seg, 4.0))
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
out = dense.run_layers(specs, x, dot=(q, seg, 4.0))
ev[1].record()
torch.cuda.synchronize()
n = 2048
buf = (C.c_ulonglong * (2 * n))()
fn = C.CDLL(_lib.SO_PATH).mvf_debug_chain_span
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, n) == 0
a = np.array(buf, dtype=np.uint64).reshape(n, 2).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
print(f"launch {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us; {len(a)} workgroups; start us: median {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}; "
      f"finish us: min {en.min():.1f} median {np.median(en):.1f} p90 {np.percentile(en, 90):.1f} max {en.max():.1f}; "
      f"duration us: median {np.median(en - st):.1f} max {(en - st).max():.1f}")
ids = np.arange(n)[: len(a)]
dur = en - st
print("mean duration by blockIdx % 8 (XCD):", np.round([dur[ids % 8 == k].mean() for k in range(8)], 1))
print("min / max inside XCD 0:", dur[ids % 8 == 0].min().round(1), dur[ids % 8 == 0].max().round(1))
print("mean duration by blockIdx // 256 (dispatch round):", np.round([dur[ids // 256 == k].mean() for k in range(len(a) // 256)], 1))
order = np.argsort(dur)
print("fastest ids", ids[order[:12]], "slowest ids", ids[order[-12:]])
