"""Where the tile z-buffer kernel spends its time on the C2 sample.
MVFUSION_SO=.../libmvfusion_zbdebug.so  -> kernel time with phases switched off (results are wrong then)
MVFUSION_SO=.../libmvfusion_zbtiming.so -> share of wave time per phase (clock64 around the phases)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
import bench  # noqa: E402
from mvfusion import _lib as L  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
sc, plan = bench.stage_sample(cfg, seed=2, device=torch.device("cuda", 0))
plan.run()
lib = L.load()
ms = (C.c_float * 16)()
names = (C.c_char_p * 16)()


def stage_times(reps=10):
    lib.mvf_profile_enable(1)
    acc = {}
    for _ in range(reps):
        plan.launch()
        torch.cuda.synchronize()
        n = lib.mvf_profile_read(ms, names, 16)
        for i in range(n):
            acc[names[i].decode()] = acc.get(names[i].decode(), 0.0) + ms[i] / reps
    lib.mvf_profile_enable(0)
    return acc


st = stage_times(20)
print("stages us:", {k: round(v * 1e3, 1) for k, v in st.items()})
if hasattr(lib, "mvf_debug_zb_switch"):
    for mask, label in [(0, "full"), (1, "no raster"), (2, "no winner stores"), (4, "no readout"),
                        (5, "no raster, no readout"), (8, "static tiles"), (13, "static, no raster, no readout"), (0, "full")]:
        lib.mvf_debug_zb_switch(mask)
        print(f"{label:24s} tile_zbuffer {stage_times()['tile_zbuffer'] * 1e3:7.1f} us")
if hasattr(lib, "mvf_debug_zb_timing"):
    out = (C.c_ulonglong * 8)()
    lib.mvf_debug_zb_timing(out, 1)
    plan.run()
    torch.cuda.synchronize()
    lib.mvf_debug_zb_timing(out, 1)
    labels = ["init", "prefetch issue", "stage", "raster", "drain", "barrier A", "readout", "barrier B + ticket"]
    tot = sum(out)
    for n, v in zip(labels, out):
        print(f"{n:22s} {v:14d} {100.0 * v / tot:6.1f} %")
    print("total wave-clocks", tot, " tile_zbuffer", stage_times()["tile_zbuffer"] * 1e3, "us")
