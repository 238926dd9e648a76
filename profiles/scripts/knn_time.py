"""Timing of the exact k-NN kernel (mvf_knn) on C1- and C2-sized clouds."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
from mvfusion import synthetic as S
from mvfusion.transforms import knn

for cfg, k in (("c1", 20), ("c2", 20), ("c2", 50)):
    xyz = torch.from_numpy(S.make_scene(cfg, seed=1)["xyz"]).cuda()
    knn(xyz, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        knn(xyz, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    n = xyz.shape[0]
    print(f"{cfg}: n={n} k={k}: {dt * 1e3:.1f} ms, {n * n / dt / 1e12:.2f} T pair evaluations/s", flush=True)
