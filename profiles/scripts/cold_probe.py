"""Where the time of MapImages goes on a sample that was never seen before (no cached point
order, nothing staged) in a process that has already mapped other samples."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
import bench as B  # noqa: E402
from mvfusion.transforms import MapImages, kdtree_order  # noqa: E402
from mvfusion.images import SameSettingImageData  # noqa: E402

dev = torch.device("cuda", 0)
sc, data, images = B.host_sample("c2", seed=2)
pos = data.pos
for th in (1, 2, 4, 8, 16, 0):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        kdtree_order(pos, True, threads=th)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"kdtree_order threads={th}: {np.median(ts):.2f} ms (min {min(ts):.2f})", flush=True)
t0 = time.perf_counter()
from sklearn.neighbors import KDTree
KDTree(np.asarray(pos[:, :2]), leaf_size=50)
print(f"sklearn: {(time.perf_counter() - t0) * 1e3:.1f} ms")

t = MapImages(method="SplattingVisibility", cylinder=True, output_device=dev, **B.VIS)


def fresh(on_device):
    d = B.Data()
    d.pos = pos.clone().to(dev) if on_device else pos.clone()
    d.mapping_index = data.mapping_index.clone()
    im = SameSettingImageData(pos=images.pos.clone(), extrinsic=images.extrinsic.clone(), fx=images.fx.clone(),
                              fy=images.fy.clone(), mx=images.mx.clone(), my=images.my.clone(),
                              ref_size=sc["img_size"], proj_upscale=1)
    return d, im


for on_device in (True, False):
    for k in range(6):
        d, im = fresh(on_device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t(d, im)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        print(f"fresh sample (cloud on {'device' if on_device else 'host'}) call {k}: {ms:.2f} ms",
              {k_: round(v, 3) for k_, v in t.timings.items()}, flush=True)
