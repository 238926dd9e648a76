"""A/B timing of the MLP chain kernel with loads / stores switched off (needs a
library built with -DMVF_MLP_DEBUG; results are wrong while a switch is on)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
from mvfusion import _lib as L, dense, pooling  # noqa: E402

dev = torch.device("cuda", 0)
lib = L.load()
torch.manual_seed(0)
for name, chans, rows in [("E_mod 64-64-64", [64, 64, 64], 2_000_000),
                          ("elt_1 8-32-32", [8, 32, 32], 2_000_000),
                          ("elt_2+K 64-32-32-16", [64, 32, 32], 2_000_000)]:
    mlp = pooling.MLP(chans, bias=False).to(dev).eval()
    specs = dense.mlp_layers(mlp)
    if "K" in name:
        specs = specs + [dense.linear_layer(torch.nn.Linear(32, 16).to(dev))]
    x = torch.rand((rows, chans[0]), device=dev)
    for mask, label in [(0, "full"), (1, "no loads"), (2, "no stores"), (3, "no loads, no stores"), (7, "+ no LDS weight reads"), (15, "+ no activation")]:
        lib.mvf_debug_mlp_switch(mask)
        for _ in range(2):
            dense.run_layers(specs, x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            dense.run_layers(specs, x)
        ev[1].record()
        torch.cuda.synchronize()
        print(f"{name:22s} {label:22s} {ev[0].elapsed_time(ev[1]) / 5 * 1e3:8.1f} us")
    lib.mvf_debug_mlp_switch(0)
