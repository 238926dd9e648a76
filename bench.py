"""Benchmark of the multi-view image->point fusion hot path on MI355X.

One "step" = one pass of the hot path over one sample resident in HBM:
fused MapImages (projection + splat z-buffer visibility + CSR assembly for all
views) followed by the atomic (pixel->view, max) and view (view->point, max and
mean) CSR pools.  Metric: points x views mapped+fused per second (whole job, all
ranks).  With N > 1 ranks every rank maps and pools its own sample (weak
scaling) and the pooled features + labels are gathered once over RCCL at the
end of the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))

from mvfusion import _lib as L                      # noqa: E402
from mvfusion import synthetic as S                 # noqa: E402
from mvfusion.images import SameSettingImageData    # noqa: E402
from mvfusion.pooling import segment_csr_two_level  # noqa: E402
from mvfusion.transforms import MapImages, kdtree_order  # noqa: E402
from mvfusion import visibility as visibility_module     # noqa: E402
from mvfusion.distributed import gather_pooled           # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec peak
MFMA_F32_PEAK = 157.3e12   # FLOP/s, dense fp32-input MFMA (v_mfma_f32_32x32x2_f32)
VIS = dict(voxel=0.05, r_max=20, r_min=0, k_swell=1.5, d_swell=1e6, exact=True,
           camera="kitti360_perspective")
WORKLOADS = {
    "c1": dict(channels=64, label="KITTI-360-shaped cylinder, 100k pts x 8 views 1408x376"),
    "c2": dict(channels=13, label="WildScenes-shaped scan, 300k pts x 16 views 2016x1512"),
    "c3": dict(channels=64, label="QKVBimodalCSRPool, 500k pts x <=32 views (Poisson 4), "
                                  "64-ch modality, 32-ch main, 8-ch mapping features, 16-ch keys"),
}


class Data:
    pass


def stage_sample(cfg, seed, device):
    """Build one synthetic sample on the host and stage it in HBM."""
    sc = S.make_scene(cfg, seed=seed)
    N = sc["xyz"].shape[0]
    T = torch.from_numpy
    data = Data()
    data.pos = T(sc["xyz"])
    data.mapping_index = torch.arange(N)
    data.mvf_point_order = kdtree_order(data.pos, True)
    images = SameSettingImageData(
        pos=T(sc["pos"]), extrinsic=T(sc["extrinsic"]), fx=T(sc["fx"]), fy=T(sc["fy"]),
        mx=T(sc["mx"]), my=T(sc["my"]), ref_size=sc["img_size"], proj_upscale=1)
    t = MapImages(method="SplattingVisibility", cylinder=True, **VIS)
    model = visibility_module.SplattingVisibility(img_size=images.proj_size, **VIS)
    plan = t.prepare_fused(data, images, model, device)
    return sc, plan


def algorithmic_bytes(N, V, W, Hc, M, F, C):
    """Closed-form byte model: xyz read once per view, one write + one read of
    an 8-byte z-buffer cell per pixel, the emitted mappings, the output CSR,
    and each pool reading its input once and writing its output once."""
    Vw = P = M
    b = {}
    b["projection"] = V * 12 * N
    b["zbuffer"] = V * 16 * W * Hc
    b["emit"] = M * (8 + 4 + 4 + 4 * F)
    b["csr"] = 8 * (N + 1) + 8 * Vw + 8 * (Vw + 1) + 4 * P + 4 * F * Vw
    b["atomic_pool"] = 4 * C * P + 4 * C * Vw + 8 * (Vw + 1)
    b["view_pool"] = 2 * (4 * C * Vw + 4 * C * N + 8 * (N + 1))     # max and mean
    b["map"] = b["projection"] + b["zbuffer"] + b["emit"] + b["csr"]
    b["total"] = b["map"] + b["atomic_pool"] + b["view_pool"]
    return b


# stage of mvf_map_images_exact -> share of the byte model it is charged with
STAGE_BYTES = {
    "bin_count": "projection", "tile_zbuffer": "zbuffer", "csr_fill": "emit+csr",
}


def cpu_baseline(cfg, C_feat, target_s=12.0):
    """The CPU oracle (single-threaded C + numpy port of the reference's CPU
    algorithm) on the same workload, timed on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mvf_oracle as orc
    sc = S.make_scene(cfg, seed=2)
    N, V = sc["xyz"].shape[0], sc["pos"].shape[0]
    order = kdtree_order(torch.from_numpy(sc["xyz"]), True).numpy()
    rng = np.random.default_rng(0)
    times = []
    t_all = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        m = orc.map_images(sc["xyz"], np.arange(N), sc, order=order, ref_size=sc["img_size"],
                           cylinder=True, **VIS)
        t_map = time.perf_counter() - t0
        Pn = m["pixels"].shape[0]
        x_pix = rng.random((Pn, C_feat), dtype=np.float32)      # input staging, not timed
        t2 = time.perf_counter()
        xv = orc.segment_csr(x_pix, m["view_pointers"], "max")
        orc.segment_csr(xv, m["pointers"], "max")
        orc.segment_csr(xv, m["pointers"], "mean")
        times.append(t_map + time.perf_counter() - t2)
        if time.perf_counter() - t_all > target_s or len(times) >= 5:
            break
    med = float(np.median(times))
    out = dict(value=N * V / med, unit="points*views/s", cores=1, kind="port",
               sample=f"{len(times)} full passes of the same workload "
                      f"({N} pts x {V} views, map + pools), median {med:.3f} s/pass")
    # the same pass with the views spread over host threads (views are
    # independent; the merge into the nested CSR and the pools stay serial)
    workers = min(V, int(os.environ.get("OMP_NUM_THREADS", "0")) or os.cpu_count() or 1)
    if workers > 1:
        par = []
        for _ in range(3):
            t0 = time.perf_counter()
            m = orc.map_images(sc["xyz"], np.arange(N), sc, order=order, ref_size=sc["img_size"],
                               cylinder=True, workers=workers, **VIS)
            xv = orc.segment_csr(x_pix, m["view_pointers"], "max")
            orc.segment_csr(xv, m["pointers"], "max")
            orc.segment_csr(xv, m["pointers"], "mean")
            par.append(time.perf_counter() - t0)
        out["threaded"] = dict(value=N * V / float(np.median(par)), cores=workers,
                               sample=f"views on {workers} threads, median {np.median(par):.3f} s/pass")
    out["host"] = dict(cpus=os.cpu_count(), model=cpu_model())
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def qkv_flops(N, Vw, F_main, F_map, C, D, G, inner=32):
    """Multiply-adds x 2 of QKVBimodalCSRPool.forward (pooling.py:454-547) with a
    DeepSetFeat map encoder: E_main, Q on N rows; mlp_elt_1, mlp_elt_2, E_mod, K,
    the compatibilities and the weighted sum on Vw rows; mlp_set on N rows."""
    per_point = F_main * inner + inner * inner + inner * D * G + inner * inner + inner * inner
    per_view = (F_map * inner + inner * inner + 2 * inner * inner + inner * inner
                + C * C + C * C + inner * D * G + D * G + C)
    return 2 * (N * per_point + Vw * per_view)


def run_qkv(args, rank, world, device, dist):
    """C3: the attentive pool on synthetic CSR groups; one step = one forward."""
    from mvfusion import pooling
    N, F_main, F_map, Cm, D, G = 500_000, 32, 8, 64, 16, 1
    rng = np.random.default_rng(3 + rank)
    counts = np.minimum(rng.poisson(4, N), 32)
    ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(device)
    Vw = int(ptr[-1])
    gen = torch.Generator(device=device).manual_seed(rank)
    x_main = torch.rand((N, F_main), device=device, generator=gen)
    x_mod = torch.rand((Vw, Cm), device=device, generator=gen)
    x_map = torch.rand((Vw, F_map), device=device, generator=gen)
    labels = torch.randint(0, 19, (N,), device=device, generator=gen)
    torch.manual_seed(0)
    mod = pooling.QKVBimodalCSRPool(in_main=F_main, in_map=F_map, in_mod=Cm, out_mod=Cm,
                                    nc_qk=D, num_groups=G).to(device).eval()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # consecutive forwards are independent: they alternate between streams so
    # that the matrix-core-bound launches of one overlap the HBM-bound ones of
    # the other
    n_lanes = max(1, args.lanes)
    streams = [torch.cuda.Stream(device) for _ in range(n_lanes)]

    def forwards(k):
        out = None
        cur = torch.cuda.current_stream(device)
        for s in streams:
            s.wait_stream(cur)
        for j in range(k):
            with torch.cuda.stream(streams[j % n_lanes]):
                out = mod(x_main, x_mod, x_map, ptr)
        for s in streams:
            cur.wait_stream(s)
        return out

    out = forwards(args.warmup)
    if dist is not None and out is not None:
        # untimed: the first collective of each kind sets up its RCCL channels
        gather_pooled(out, labels)
    barrier()
    t0 = time.perf_counter()
    out = forwards(args.steps)
    if dist is not None:
        gather_pooled(out, labels)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    # launch duration of the whole forward with HIP events on the launch stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.steps):
        mod(x_main, x_mod, x_map, ptr)
    ev[1].record()
    torch.cuda.synchronize()
    fwd_ms = ev[0].elapsed_time(ev[1]) / args.steps
    flops = qkv_flops(N, Vw, F_main, F_map, Cm, D, G)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get("c3", {}).get("forward")
    bytes_ = 4 * (F_main * N + (Cm + F_map) * Vw + Cm * N) + 8 * (N + 1)
    line = {
        "metric": "points*views mapped+fused/sec",
        "value": world * Vw * args.steps / elapsed, "unit": "points*views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOADS["c3"]["label"], "name": "c3", "points_per_gpu": N,
                   "point_view_rows": Vw, "streams_per_gpu": n_lanes,
                   "parallelism": f"sample-sharded x{world}, final RCCL all-gather"},
        "roofline": {"bound": "mfma", "kernel": "QKVBimodalCSRPool.forward (all launches)",
                     "achieved": flops / (fwd_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12,
                     "unit": "TFLOP/s", "frac": flops / (fwd_ms * 1e-3) / MFMA_F32_PEAK,
                     "traffic": traffic, "flops_per_launch": flops, "kernel_ms": fwd_ms,
                     "overlapped": {"ms_per_forward": elapsed / args.steps * 1e3,
                                    "achieved": flops / (elapsed / args.steps) / 1e12,
                                    "frac": flops / (elapsed / args.steps) / MFMA_F32_PEAK},
                     "hbm": {"algorithmic_bytes": bytes_,
                             "frac": bytes_ / (fwd_ms * 1e-3) / HBM_PEAK}},
    }
    if not args.no_cpu_baseline:
        # the oracle's numpy restatement of the same forward on the first 50k points
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mvf_oracle as orc
        n_cpu = 50_000
        rows_cpu = int(ptr[n_cpu])
        sd = {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}
        inputs = (x_main[:n_cpu].cpu().numpy(), x_mod[:rows_cpu].cpu().numpy(),
                  x_map[:rows_cpu].cpu().numpy(), ptr[:n_cpu + 1].cpu().numpy())
        want = orc.qkv_pool(*inputs, sd, num_groups=G, nc_qk=D)
        ts = []
        for _ in range(3):
            c0 = time.perf_counter()
            orc.qkv_pool(*inputs, sd, num_groups=G, nc_qk=D)
            ts.append(time.perf_counter() - c0)
        err = float(np.abs(out[:n_cpu].cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-30))
        line["cpu_baseline"] = {
            "value": rows_cpu / sorted(ts)[1], "unit": "points*views/s", "cores": 1, "kind": "port",
            "sample": f"oracle QKV forward on the first {n_cpu} points ({rows_cpu} rows) of the same "
                      f"workload, median of 3: {sorted(ts)[1]:.3f} s; GPU output matches it to {err:.1e} (rel. to max)"}
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=3,
                    help="streams that consecutive steps alternate between (1 = no overlap)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    cfg = args.config
    if cfg == "c3":
        run_qkv(args, rank, world, device, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    C_feat = WORKLOADS[cfg]["channels"]
    sc, plan = stage_sample(cfg, seed=2 + rank, device=device)
    N, V = plan.N, plan.V
    W, H = sc["img_size"]
    gen = torch.Generator(device=device).manual_seed(rank)
    labels = torch.randint(0, 19, (N,), device=device, generator=gen)
    # pixel features for every possible mapping row, resident in HBM
    plan.run()
    cap = plan.out_cap
    x_pix = torch.rand((cap, C_feat), device=device, generator=gen)
    atomic_ptr = torch.arange(cap + 1, dtype=torch.int64, device=device)
    lib = L.load()

    def pools(res):
        pointers, img_ids, pixels, feats, view_seen = res
        P = img_ids.shape[0]
        # pixel -> view max, then view -> point max and mean, in one pass over x_pix
        x_max, x_mean = segment_csr_two_level(x_pix[:P], atomic_ptr[:P + 1], pointers,
                                              inner="max", outer=("max", "mean"))
        return x_max, x_mean, P

    # Consecutive samples are independent: steps alternate between two streams
    # (each with its own plan: outputs and workspace), so that the
    # latency-bound kernels of one step overlap with the other step's.  On each
    # stream the mapping pass of the next step is enqueued before the sizes of
    # the current one are waited for, so the device does not idle on the host
    # round trip.
    n_lanes = max(1, args.lanes)
    plans = [plan] + [plan.twin(k) for k in range(1, n_lanes)]
    streams = [torch.cuda.Stream(device) for _ in range(n_lanes)]
    # set-up: every lane allocates its outputs and workspace once
    for p_, s_ in zip(plans, streams):
        with torch.cuda.stream(s_):
            p_.run()
    torch.cuda.synchronize()

    def run_steps(k):
        """k full steps (mapping pass, size read-back, three pools each)."""
        out = None
        pending = [None] * n_lanes
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(device))
        for j in range(min(n_lanes, k)):
            with torch.cuda.stream(streams[j]):
                pending[j] = plans[j].run_async()
        for j in range(k):
            ln = j % n_lanes
            with torch.cuda.stream(streams[ln]):
                nxt = plans[ln].run_async() if j + n_lanes < k else None
                out = pools(pending[ln].result())
                pending[ln] = nxt
        for s in streams:
            torch.cuda.current_stream(device).wait_stream(s)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = run_steps(args.warmup) if args.warmup > 0 else None
    if dist is not None and out is not None:
        # untimed: the first collective of each kind sets up its RCCL channels
        gather_pooled(torch.cat(out[:2], dim=1), labels)
    barrier()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    if dist is not None:
        # the path's only exchange: final gather of pooled features and labels
        feats_all, labels_all, _ = gather_pooled(torch.cat(out[:2], dim=1), labels)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel timing with HIP events on the launch stream (rank 0) ----
    roofline = None
    stages = {}
    if rank == 0:
        lib.mvf_profile_enable(1)
        ms = (C.c_float * 16)()
        names = (C.c_char_p * 16)()
        pool_ms = {"pools": 0.0}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(args.steps):
            pointers, img_ids, pixels, feats, view_seen = plan.run()
            n = lib.mvf_profile_read(ms, names, 16)
            for i in range(n):
                k = names[i].decode()
                stages[k] = stages.get(k, 0.0) + ms[i]
            P = img_ids.shape[0]
            ev[0].record()
            segment_csr_two_level(x_pix[:P], atomic_ptr[:P + 1], pointers, inner="max",
                                  outer=("max", "mean"))
            ev[1].record()
            torch.cuda.synchronize()
            pool_ms["pools"] += ev[0].elapsed_time(ev[1])
        lib.mvf_profile_enable(0)
        stages.update(pool_ms)
        stages = {k: v / args.steps for k, v in stages.items()}          # ms per step
        M = plan.last_counts["mappings"]
        F = plan.F
        Hc = H
        B = algorithmic_bytes(N, V, W, Hc, M, F, C_feat)
        charged = {"bin_count": B["projection"], "tile_zbuffer": B["zbuffer"],
                   "csr_fill": B["emit"] + B["csr"], "pools": B["atomic_pool"] + B["view_pool"]}
        charged = {k: v for k, v in charged.items() if k in stages}
        dom = max(charged, key=lambda k: stages.get(k, 0.0))
        ach = charged[dom] / (stages[dom] * 1e-3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(cfg, {}).get(dom)
        roofline = dict(
            bound="hbm", kernel=dom, achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
            frac=ach / HBM_PEAK, traffic=traffic,
            algorithmic_bytes_per_launch=charged[dom], kernel_ms=stages[dom],
            pipeline=dict(
                algorithmic_bytes_per_step=B["total"],
                ms_per_step=elapsed / args.steps * 1e3,
                achieved=B["total"] / (elapsed / args.steps) / 1e9,
                frac=B["total"] / (elapsed / args.steps) / HBM_PEAK),
            stage_ms={k: round(v, 4) for k, v in stages.items()})

    if rank == 0:
        line = {
            "metric": "points*views mapped+fused/sec",
            "value": world * N * V * args.steps / elapsed,
            "unit": "points*views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[cfg]["label"]
                       + f", exact splat visibility + max atomic pool + max/mean view pools (C={C_feat}, pools fused in one pass)",
                       "name": cfg, "points_per_gpu": N, "views": V, "image": [W, H],
                       "mappings_per_sample": plan.last_counts["mappings"],
                       "splat_tile_pairs": plan.last_counts["pairs"],
                       "streams_per_gpu": n_lanes,
                       "parallelism": f"sample-sharded x{world}, final RCCL all-gather"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, C_feat)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
