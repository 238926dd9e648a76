import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multimodal-fusion_amd"))
from mvfusion import pooling, dense
torch.set_grad_enabled(False)
dev = "cuda:0"
rows = 2_000_000
for cin, widths in ((6, (64, 64)), (6, (32, 32)), (3, (8, 8))):
    x = torch.rand(rows, cin, device=dev)
    mlp = pooling.MLP([cin, *widths], bias=False).to(dev).eval()
    specs = dense.mlp_layers(mlp)
    def general():
        out = torch.empty(rows, widths[-1], device=dev)
        dense._launch((x, None, None, None, 0), [(s, 0, s.cout) for s in specs], None, out, widths[-1])
        return out
    for name, fn in (("padded", lambda: dense.run_layers(specs, x)), ("general", general)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(cin, widths, name, f"{e0.elapsed_time(e1)/20*1000:.0f} us")
