"""One training step (forward + backward) of QKVBimodalCSRPool at C3 size, with the
BatchNorm1d + LeakyReLU tails and the weight gradients of the dense layers on this
library's kernels (training.batch_norm_act, training.linear) and on the torch modules."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multimodal-fusion_amd"))
from mvfusion import pooling, training

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
# argument "group": GroupBimodalCSRPool with 4 groups (the flagship model's setting) instead
if len(sys.argv) > 1 and sys.argv[1] == "group":
    mod = pooling.GroupBimodalCSRPool(in_map=8, in_mod=64, num_groups=4, gating=True).to(dev).train()
else:
    mod = pooling.QKVBimodalCSRPool(in_main=32, in_map=8, in_mod=64, out_mod=64).to(dev).train()
own = (training.batch_norm_act, training.linear)
plain = (lambda z, bn, act, col_sums=None: act(bn(z)),
         lambda lin, x, want_sums=False: (lin(x), None) if want_sums else lin(x))


def step():
    mod.zero_grad(set_to_none=True)
    x_main.grad = x_mod.grad = x_map.grad = None
    out = mod(x_main if isinstance(mod, pooling.QKVBimodalCSRPool) else None, x_mod, x_map, ptr)
    out.sum().backward()


for name, fns in (("kernels", own), ("torch modules", plain), ("kernels", own)):
    training.batch_norm_act, training.linear = fns
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.2f} ms per training step ({Vw} rows), "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB")
    torch.cuda.reset_peak_memory_stats()
