"""What slows the host KD-tree order down inside MapImages: fresh memory, idle cores, GPU work in between."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
import bench as B
from mvfusion.transforms import kdtree_order
sc, data, images = B.host_sample("c2", seed=2)
pos = data.pos
def t(f, n=6):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); out.append((time.perf_counter() - t0) * 1e3)
    return " ".join(f"{v:.2f}" for v in out)
print("same tensor, back to back:      ", t(lambda: kdtree_order(pos, True)))
print("fresh clone each call:          ", t(lambda: kdtree_order(pos.clone(), True)))
def idle():
    time.sleep(0.02); kdtree_order(pos, True)
print("20 ms sleep before (incl.):     ", t(idle))
x = torch.zeros(1 << 20, device="cuda")
def gpu_then():
    for _ in range(20): x.add_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); kdtree_order(pos, True); return (time.perf_counter() - t0) * 1e3
print("after GPU work + synchronize:   ", " ".join(f"{gpu_then():.2f}" for _ in range(6)))
d = pos.cuda()
def dev():
    t0 = time.perf_counter(); h = d.cpu(); t1 = time.perf_counter(); kdtree_order(h, True); t2 = time.perf_counter()
    return f"{(t1 - t0) * 1e3:.2f}+{(t2 - t1) * 1e3:.2f}"
print("device cloud: .cpu() + order:   ", " ".join(dev() for _ in range(6)))
pin = torch.empty(pos.shape, dtype=torch.float32).pin_memory()
def dev_pin():
    t0 = time.perf_counter(); pin.copy_(d, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter(); kdtree_order(pin, True); t2 = time.perf_counter()
    return f"{(t1 - t0) * 1e3:.2f}+{(t2 - t1) * 1e3:.2f}"
print("device cloud: pinned + order:   ", " ".join(dev_pin() for _ in range(6)))
