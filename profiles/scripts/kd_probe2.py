import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
import bench as B
from mvfusion import _lib as L
from mvfusion.transforms import kdtree_order
sc, data, images = B.host_sample("c2", seed=2)
pos = data.pos
lib = L.load()
def raw(p, order):
    t0 = time.perf_counter()
    lib.mvf_kdtree_order(p.ctypes.data, p.shape[0], 2, 3, 50, order.ctypes.data, 0)
    return (time.perf_counter() - t0) * 1e3
order = np.empty(pos.shape[0], np.int64)
p0 = np.asarray(pos)
print("raw call, same arrays:        ", " ".join(f"{raw(p0, order):.2f}" for _ in range(6)))
print("raw call, fresh input copy:   ", " ".join(f"{raw(p0.copy(), order):.2f}" for _ in range(6)))
print("raw call, fresh output:       ", " ".join(f"{raw(p0, np.empty(pos.shape[0], np.int64)):.2f}" for _ in range(6)))
print("raw call, torch clone input:  ", " ".join(f"{raw(np.asarray(pos.clone()), order):.2f}" for _ in range(6)))
big = [np.zeros(1 << 20) for _ in range(4)]
def churn():
    x = torch.empty(3_000_000).zero_(); del x
print("raw call after alloc churn:   ", end="")
for _ in range(6):
    churn(); print(f" {raw(p0, order):.2f}", end="")
print()
dev = torch.device("cuda", 0)
x = torch.zeros(10, device=dev); torch.cuda.synchronize()
print("after HIP init, same arrays:  ", " ".join(f"{raw(p0, order):.2f}" for _ in range(6)))
print("after HIP init, fresh input:  ", " ".join(f"{raw(p0.copy(), order):.2f}" for _ in range(6)))
print("after HIP init, torch clone:  ", " ".join(f"{raw(np.asarray(pos.clone()), order):.2f}" for _ in range(6)))
t = torch.empty(pos.shape, dtype=torch.float32).pin_memory()
def via_pinned():
    src = pos.clone()
    t0 = time.perf_counter(); t.copy_(src); r = raw(np.asarray(t), order); return (time.perf_counter() - t0) * 1e3
print("clone -> pinned copy -> call: ", " ".join(f"{via_pinned():.2f}" for _ in range(6)))
