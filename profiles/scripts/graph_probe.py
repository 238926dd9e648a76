"""Step time of the fused MapImages pass + pools launched kernel by kernel against the same
launch sequence captured once into a HIP graph and replayed (one graph per stream).
usage: python profiles/scripts/graph_probe.py [c1|c2|c4|c5] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
import bench as B  # noqa: E402
from mvfusion.pooling import segment_csr_two_level  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
# "nocopy": the captured graph holds kernels (and the call's memset) only, no copy of the sizes
NOCOPY = len(sys.argv) > 3 and sys.argv[3] == "nocopy"
dev = torch.device("cuda", 0)
wl = B.WORKLOADS[cfg]
C_feat = wl["channels"]
sc, plan = B.stage_sample(wl["scene"], 2, dev, depth=cfg == "c5", n_views=16 if cfg == "c5" else None)
plan.run()
plan.run()                       # capacities settled
gen = torch.Generator(device=dev).manual_seed(0)
x_pix = torch.rand((plan.out_cap, C_feat), device=dev, generator=gen)
aptr = torch.arange(plan.out_cap + 1, dtype=torch.int64, device=dev)


def pools(pointers):
    return segment_csr_two_level(x_pix, aptr, pointers, inner="max", outer=("max", "mean"))


for lanes in (1, 3):
    plans = [plan] + [plan.twin(k) for k in range(1, lanes)]
    streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
    for p, s in zip(plans, streams):
        with torch.cuda.stream(s):
            p.run()
    torch.cuda.synchronize()

    # kernel by kernel
    def loop_plain(k):
        pend = [None] * lanes
        for j in range(min(lanes, k)):
            with torch.cuda.stream(streams[j]):
                pend[j] = plans[j].run_async()
        out = None
        for j in range(k):
            ln = j % lanes
            with torch.cuda.stream(streams[ln]):
                nxt = plans[ln].run_async() if j + lanes < k else None
                res = pend[ln].result()
                out = pools(res[0])
                pend[ln] = nxt
        return out

    loop_plain(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop_plain(steps)
    torch.cuda.synchronize()
    t_plain = (time.perf_counter() - t0) / steps * 1e3

    # captured: mapping pass + copy of the sizes + pools, one graph per stream
    graphs, events, hosts, outs = [], [], [], []
    for p, s in zip(plans, streams):
        sl = p._slot(0)
        p._workspace()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            p.launch(0)
            if not NOCOPY:
                sl[2].copy_(sl[1], non_blocking=True)
            outs.append(pools(sl[4]))
        graphs.append(g)
        events.append(torch.cuda.Event())
        hosts.append(sl[2])

    def loop_graph(k):
        for j in range(min(lanes, k)):
            with torch.cuda.stream(streams[j]):
                graphs[j].replay()
                events[j].record(streams[j])
        for j in range(k):
            ln = j % lanes
            events[ln].synchronize()
            assert NOCOPY or hosts[ln][2].item() == 0          # no overflow
            if j + lanes < k:
                with torch.cuda.stream(streams[ln]):
                    graphs[ln].replay()
                    events[ln].record(streams[ln])

    loop_graph(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop_graph(steps)
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / steps * 1e3
    # same results?
    with torch.cuda.stream(streams[0]):
        res = plans[0].run()
        want = pools(res[0])
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(outs[0], want))
    print(f"{cfg} lanes={lanes}: kernel by kernel {t_plain:.4f} ms/step, graph replay {t_graph:.4f} ms/step, "
          f"same pooled features: {same}", flush=True)
    del graphs
