"""GroupBimodalCSRPool forward at C3 size, 20 calls, for a kernel trace: python group_pool_trace.py <groups>."""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multimodal-fusion_amd"))
from mvfusion import pooling
dev = "cuda:0"
rng = np.random.default_rng(3)
N = 500000
counts = np.minimum(rng.poisson(4, N), 32)
ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(dev)
Vw = int(ptr[-1])
x_mod = torch.rand(Vw, 64, device=dev)
x_map = torch.rand(Vw, 8, device=dev)
torch.set_grad_enabled(False)
G = int(sys.argv[1])
torch.manual_seed(0)
mod = pooling.GroupBimodalCSRPool(in_map=8, in_mod=64, num_groups=G, gating=True).to(dev).eval()
for _ in range(20):
    out = mod(None, x_mod, x_map, ptr)
torch.cuda.synchronize()
