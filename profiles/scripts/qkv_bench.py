"""QKVBimodalCSRPool forward on the C3-shaped workload (500k points, ~2M
point-view rows, 64-channel modality features, 16-channel keys/queries)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
from mvfusion import pooling  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
N = 500_000
counts = np.minimum(rng.poisson(4, N), 32)
ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(dev)
Vw = int(ptr[-1])
g = torch.Generator(device=dev).manual_seed(0)
x_main = torch.rand((N, 32), device=dev, generator=g)
x_mod = torch.rand((Vw, 64), device=dev, generator=g)
x_map = torch.rand((Vw, 8), device=dev, generator=g)
torch.manual_seed(0)
mod = pooling.QKVBimodalCSRPool(in_main=32, in_map=8, in_mod=64, out_mod=64, nc_qk=16).to(dev).eval()
for _ in range(3):
    out = mod(x_main, x_mod, x_map, ptr)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    out = mod(x_main, x_mod, x_map, ptr)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
D, G, C = 16, 1, 64
flops = 2 * (N * (32 * 32 + 32 * 32 + 32 * D * G + 32 * 32 + 32 * 32)
             + Vw * (8 * 32 + 32 * 32 + 64 * 32 + 32 * 32 + 64 * C + C * C + 32 * D * G + D * G + C))
print(f"rows {Vw}  {ms:.3f} ms/forward  {flops / ms / 1e9:.1f} TFLOP/s of 157.3 fp32 MFMA peak "
      f"({100 * flops / ms / 1e9 / 157.3:.1f} %)  {Vw / ms / 1e3:.1f} M rows/s")
