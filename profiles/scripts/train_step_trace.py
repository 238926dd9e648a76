"""A few training steps of QKVBimodalCSRPool at C3 size for rocprofv3 --kernel-trace --stats."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multimodal-fusion_amd"))
from mvfusion import pooling

dev = "cuda:0"
rng = np.random.default_rng(3)
N = 500000
counts = np.minimum(rng.poisson(4, N), 32)
ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(dev)
Vw = int(ptr[-1])
x_main = torch.rand(N, 32, device=dev, requires_grad=True)
x_mod = torch.rand(Vw, 64, device=dev, requires_grad=True)
x_map = torch.rand(Vw, 8, device=dev, requires_grad=True)
torch.manual_seed(0)
group = len(sys.argv) > 1 and sys.argv[1] == "group"
if group:
    mod = pooling.GroupBimodalCSRPool(in_map=8, in_mod=64, num_groups=4, gating=True).to(dev).train()
else:
    mod = pooling.QKVBimodalCSRPool(in_main=32, in_map=8, in_mod=64, out_mod=64).to(dev).train()
for _ in range(5):
    mod.zero_grad(set_to_none=True)
    x_main.grad = x_mod.grad = x_map.grad = None
    mod(None if group else x_main, x_mod, x_map, ptr).sum().backward()
torch.cuda.synchronize()
