"""ImageMapping.select_points(mode='merge') and from_dense at the C3 mapping size (500 k points,
~2 M view rows, one pixel each): the device assembly (mvf_group_mean + mvf_mapping_from_dense)
against the same operations composed from torch sorts / repeat_interleave."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
from mvfusion import mapping as M  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
N, n_img = 500_000, 32
counts = np.minimum(rng.poisson(4, N), 32)
ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
Vw = int(ptr[-1])
point_ids = np.repeat(np.arange(N), counts)
image_ids = np.concatenate([rng.choice(n_img, c, replace=False) for c in counts]) if False else \
    (np.arange(Vw) - ptr[point_ids] + rng.integers(0, n_img, N)[point_ids]) % n_img
pixels = np.stack([rng.integers(0, 512, Vw), rng.integers(0, 384, Vw)], 1).astype(np.int16)
feats = rng.random((Vw, 8), dtype=np.float32)
T = lambda a: torch.from_numpy(a).to(dev)
args = (T(point_ids), T(image_ids), T(pixels), T(feats))
idx = T(np.arange(N) // 8)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


m = M.ImageMapping.from_dense(*args, num_points=N)
t_fd = timed(lambda: M.ImageMapping.from_dense(*args, num_points=N))
t_merge = timed(lambda: m.select_points(idx, mode="merge"))
fits = M._device_assembly_fits
M._device_assembly_fits = lambda *a: False          # the composition from torch sorts
t_fd_torch = timed(lambda: M.ImageMapping.from_dense(*args, num_points=N))
t_merge_torch = timed(lambda: m.select_points(idx, mode="merge"))
M._device_assembly_fits = fits
print(f"rows {Vw}: from_dense {t_fd:.2f} ms (torch composition {t_fd_torch:.2f} ms); "
      f"select_points merge 8:1 {t_merge:.2f} ms (torch composition {t_merge_torch:.2f} ms)")
