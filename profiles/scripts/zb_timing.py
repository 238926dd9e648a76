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
if hasattr(lib, "mvf_debug_zb_timing"):
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
else:
    plan.run()
    torch.cuda.synchronize()
import numpy as np
G = 1024
span = (C.c_ulonglong * (6 * G))()
lib.mvf_debug_zb_span(span, G)
raw = np.array(span[:], dtype=np.uint64).reshape(G, 6)
dry_t = (raw[:, 3] & np.uint64((1 << 56) - 1)).astype(np.float64)
dry_n = (raw[:, 3] >> np.uint64(56)).astype(np.float64)
a = raw[:, :3].astype(np.float64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en, nt = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2]
print("workgroups", len(a), "tiles", int(nt.sum()), "tiles/WG min/mean/max", nt.min(), nt.mean(), nt.max())
print("start us  pct 0/50/100:", np.percentile(st, [0, 50, 100]))
print("finish us pct 0/10/50/90/99/100:", np.percentile(en, [0, 10, 50, 90, 99, 100]))
print("mean busy us", (en - st).mean(), " last finish", en.max())
h, edges = np.histogram(en, bins=12)
print("finish histogram:", list(zip(np.round(edges[:-1], 1), h)))
idx = np.arange(G)[: len(en)]
print("per queue (q: min/max finish, tiles):")
for q in range(32):
    m = idx % 32 == q
    print(f"  q{q:2d} xcd{q % 8} {en[m].min():6.1f} {en[m].max():6.1f} tiles {int(nt[m].sum())}", end=";")
    if q % 4 == 3:
        print()
print("finish by blockIdx // 128:", [round(float(en[(idx // 128) == k].mean()), 1) for k in range(8)])

dry = (dry_t - t0) / 100.0
print("dry seen us pct 0/50/100:", np.percentile(dry, [0, 50, 100]), " tiles after dry (mean):", (nt - dry_n).mean())
print("finish - dry pct 0/50/100:", np.percentile(en - dry, [0, 50, 100]))

ld, ln = raw[:, 4].astype(np.float64) / 100.0, raw[:, 5].astype(np.float64)
print("last tile: duration us pct 0/50/90/100", np.percentile(ld, [0, 50, 90, 100]), " entries pct 0/50/90/100", np.percentile(ln, [0, 50, 90, 100]))
o = np.argsort(-en)[:8]
print("latest finishers: blockIdx, finish, last dur, last entries:", [(int(i), float(en[i]), float(ld[i]), int(ln[i])) for i in o])

ntile = int(nt.sum())
tb = (C.c_uint * (2 * ntile))()
lib.mvf_debug_zb_tiles(tb, ntile)
tt = np.array(tb[:], dtype=np.float64).reshape(ntile, 2)
print("list position -> mean duration us / mean entries / max duration, per 1/16 of the list:")
for k in range(16):
    sl = tt[k * ntile // 16:(k + 1) * ntile // 16]
    print(f"  {k:2d}: {sl[:, 0].mean() / 100:7.2f} {sl[:, 1].mean():8.1f} {sl[:, 0].max() / 100:7.2f}")
print("sum of durations / 1024 =", tt[:, 0].sum() / 100 / 1024)

lg = (C.c_uint * (1024 * 16 * 3))()
lib.mvf_debug_zb_log(lg)
lg = np.array(lg[:], dtype=np.int64).reshape(1024, 16, 3)
for w in list(o[:3]) + [0, 1, 500]:
    n = int(nt[w])
    print("WG", int(w), [(int(lg[w, j, 0]), round(lg[w, j, 1] / 100.0, 1), int(lg[w, j, 2])) for j in range(min(n, 16))])
print("workgroups starting later than 1 us:", int((st > 1.0).sum()), " their blockIdx:", list(idx[st > 1.0][:20]))
