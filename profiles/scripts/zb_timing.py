"""Phase timing of the tile z-buffer kernel (needs a library built with
-DMVF_ZB_TIMING).  Prints the share of wave time per phase."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
import bench  # noqa: E402
from mvfusion import _lib as L  # noqa: E402

sc, plan = bench.stage_sample("c2", seed=2, device=torch.device("cuda", 0))
plan.run()
lib = L.load()
out = (C.c_ulonglong * 8)()
lib.mvf_debug_zb_timing(out, 1)
plan.run()
torch.cuda.synchronize()
lib.mvf_debug_zb_timing(out, 1)
names = ["init", "prefetch issue", "stage", "raster", "drain", "barrier A", "readout", "barrier B + ticket"]
tot = sum(out)
for n, v in zip(names, out):
    print(f"{n:22s} {v:14d} {100.0 * v / tot:6.1f} %")
print("total wave-clocks", tot)
