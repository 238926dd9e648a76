"""Output and input-gradient products of a Linear over 2 M rows: the library GEMM (torch) against
the single-layer MFMA chain launch (dense.run_layers with one plain Linear)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multimodal-fusion_amd"))
from mvfusion import dense

torch.set_grad_enabled(False)
dev = "cuda:0"
rows = 2_002_000


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


for cin, cout in ((64, 64), (8, 32), (32, 32), (64, 32), (32, 16), (16, 32), (40, 32)):
    x = torch.randn(rows, cin, device=dev)
    lin = torch.nn.Linear(cin, cout, bias=False).to(dev)
    spec = [dense.linear_layer(lin)]
    t_lib = timed(lambda: torch.nn.functional.linear(x, lin.weight))
    t_own = timed(lambda: dense.run_layers(spec, x))
    err = float((dense.run_layers(spec, x) - torch.nn.functional.linear(x, lin.weight)).abs().max())
    print(f"{cin:3d} -> {cout:3d}: library {t_lib:6.0f} us, chain kernel {t_own:6.0f} us (max diff {err:.1e})")
