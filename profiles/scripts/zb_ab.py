"""A/B timing of the tile z-buffer kernel with phases switched off (needs a
library built with -DMVF_ZB_DEBUG; results are wrong while a switch is on)."""
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
ms = (C.c_float * 16)()
names = (C.c_char_p * 16)()
for mask, label in [(0, "full"), (1, "no raster"), (2, "no winner stores"), (4, "no readout"),
                    (5, "no raster, no readout"), (0, "full")]:
    lib.mvf_debug_zb_switch(mask)
    lib.mvf_profile_enable(1)
    acc = {}
    for _ in range(10):
        plan.launch()
        torch.cuda.synchronize()
        n = lib.mvf_profile_read(ms, names, 16)
        for i in range(n):
            acc[names[i].decode()] = acc.get(names[i].decode(), 0.0) + ms[i] / 10
    lib.mvf_profile_enable(0)
    print(f"{label:24s} tile_zbuffer {acc['tile_zbuffer'] * 1e3:7.1f} us")
