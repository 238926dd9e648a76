"""cProfile of the steady-state MapImages.__call__ on one C2 sample (host objects in, mappings
out): where the host time of a call goes.  Run from the repo root on the GPU box."""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, "multimodal-fusion_amd")
sys.path.insert(0, ".")
import torch

import bench
from mvfusion.transforms import MapImages

sc, data, images = bench.host_sample("c2", 2)
t = MapImages(method="SplattingVisibility", cylinder=True, **bench.VIS)
for _ in range(5):
    t(data, images[torch.arange(images.num_views)])
torch.cuda.synchronize()
fresh = [images[torch.arange(images.num_views)] for _ in range(300)]
t0 = time.perf_counter()
for im in fresh[:100]:
    t(data, im)
torch.cuda.synchronize()
print("steady ms per call (fresh images objects)", (time.perf_counter() - t0) * 10)
t0 = time.perf_counter()
for _ in range(100):
    images.mappings = None
    t(data, images)
torch.cuda.synchronize()
print("steady ms per call (same images object)", (time.perf_counter() - t0) * 10)
for label, calls in (("same images object", [images] * 200), ("fresh images objects", fresh[100:300])):
    pr = cProfile.Profile()
    pr.enable()
    for im in calls:
        im.mappings = None
        t(data, im)
    pr.disable()
    print("=====", label)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
