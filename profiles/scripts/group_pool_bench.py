"""GroupBimodalCSRPool (4 groups, the reference's flagship setting) at C3 size: forward time with
the pool epilogue (save_last=False) and with the unfused launches (save_last=True)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "multimodal-fusion_amd")
from mvfusion import pooling

dev = "cuda:0"
rng = np.random.default_rng(3)
N = 500000
counts = np.minimum(rng.poisson(4, N), 32)
ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(dev)
Vw = int(ptr[-1])
x_mod = torch.rand(Vw, 64, device=dev)
x_map = torch.rand(Vw, 8, device=dev)
torch.set_grad_enabled(False)       # forward only: the fused inference kernels
for G in (1, 4):
    torch.manual_seed(0)
    mod = pooling.GroupBimodalCSRPool(in_map=8, in_mod=64, num_groups=G, gating=True).to(dev).eval()
    for save_last in (False, True):
        mod.save_last = save_last
        for _ in range(5):
            out = mod(None, x_mod, x_map, ptr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            out = mod(None, x_mod, x_map, ptr)
        e1.record()
        torch.cuda.synchronize()
        print(f"groups {G} save_last {save_last}: {e0.elapsed_time(e1) / 50:.3f} ms per forward ({Vw} rows)")
