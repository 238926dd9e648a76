"""Benchmark of the multi-view image->point fusion hot path on MI355X.

One "step" = one pass of the hot path over one sample (or one rank's batch of
samples) resident in HBM: MapImages for all views (projection, visibility,
CSR assembly) followed by the atomic (pixel->view, max) and view (view->point,
max and mean) CSR pools.  Metric: points x views mapped+fused per second (whole
job, all ranks).  With N > 1 ranks every rank maps and pools its own sample
(weak scaling) and the pooled features + labels are gathered once over RCCL at
the end of the timed region.

Workloads (--config):
  c2  WildScenes-shaped scan, 300k pts x 16 views 2016x1512, exact splat visibility   (headline)
  c1  KITTI-360-shaped cylinder, 100k pts x 8 views 1408x376, exact splat visibility
  c3  QKVBimodalCSRPool forward, 500k pts x <= 32 views
  c4  a rank's share of the 64-cylinder batch: 8 x c1, mapped on 3 streams, one ImageMappingBatch, pooled at once
  c5  a rank's share of the dense scene: 2M pts x 16 (--views) views of 4 MP, depth-map visibility

The default run (c2) also runs short versions of the other workloads and a
wall-clock measurement of MapImages.__call__ from host tensors; they are nested
in the one JSON line under "other_workloads" and "end_to_end".

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
from mvfusion.distributed import gather_pooled, rank_env, timed_steps   # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec peak
MFMA_F32_PEAK = 157.3e12   # FLOP/s, dense fp32-input MFMA (v_mfma_f32_32x32x2_f32)
# LDS atomic-minimum issue cost: 6.58 cycles per 64-bit wave-instruction per CU at the
# 2.4 GHz the cycle count was taken at (profiles/microbench/lds_ops_bench.hip), 256 CUs
LDS_ATOMIC_PEAK = 256 * 2.4e9 / 6.58          # wave-instructions / s
VIS = dict(voxel=0.05, r_max=20, r_min=0, k_swell=1.5, d_swell=1e6, exact=True,
           camera="kitti360_perspective")
WORKLOADS = {
    "c1": dict(scene="c1", channels=64, label="KITTI-360-shaped cylinder, 100k pts x 8 views 1408x376"),
    "c2": dict(scene="c2", channels=13, label="WildScenes-shaped scan, 300k pts x 16 views 2016x1512"),
    "c3": dict(channels=64, label="QKVBimodalCSRPool, 500k pts x <=32 views (Poisson 4), "
                                  "64-ch modality, 32-ch main, 8-ch mapping features, 16-ch keys"),
    "c4": dict(scene="c1", channels=64, samples=8,
               label="one rank's share of the 64-cylinder batch: 8 KITTI-360-shaped cylinders "
                     "(100k pts x 8 views 1408x376 each), one ImageMappingBatch"),
    "c5": dict(scene="c5", channels=64, label="one rank's share of the dense scene: 2M pts x 4 MP views, "
                                              "depth-map visibility"),
}


class Data:
    pass


def host_sample(scene, seed, n_views=None, camera="pinhole"):
    """Synthetic sample as the host tensors a data loader would hand over.
    camera='equirect': the same cloud and camera positions seen through
    S3DIS-style 2048 x 1024 equirectangular panoramas."""
    kw = {} if n_views is None else dict(n_views=n_views)
    # points in the order a voxel-grid sampler leaves them (the first pre-transform of the
    # reference's pipelines); MVF_BENCH_POINT_ORDER=random shuffles them instead
    kw["order"] = os.environ.get("MVF_BENCH_POINT_ORDER", "voxel")
    sc = S.make_scene(scene, seed=seed, **kw)
    N = sc["xyz"].shape[0]
    T = torch.from_numpy
    data = Data()
    data.pos = T(sc["xyz"])
    data.mapping_index = torch.arange(N)
    if camera == "equirect":
        V = sc["pos"].shape[0]
        yaw = np.linspace(0, 2 * np.pi, V, endpoint=False)
        sc["opk"] = np.stack([0.03 * np.cos(yaw), 0.05 * np.sin(yaw), yaw], 1).astype(np.float32)
        sc["img_size"] = (2048, 1024)
        images = SameSettingImageData(pos=T(sc["pos"]), opk=T(sc["opk"]), ref_size=sc["img_size"],
                                      proj_upscale=1)
        return sc, data, images
    images = SameSettingImageData(
        pos=T(sc["pos"]), extrinsic=T(sc["extrinsic"]), fx=T(sc["fx"]), fy=T(sc["fy"]),
        mx=T(sc["mx"]), my=T(sc["my"]), ref_size=sc["img_size"], proj_upscale=1)
    return sc, data, images


def stage_sample(scene, seed, device, depth=False, n_views=None, exact=True, camera="pinhole"):
    """Build one synthetic sample on the host and stage it in HBM."""
    sc, data, images = host_sample(scene, seed, n_views, camera)
    if depth:
        V = sc["pos"].shape[0]
        kw = {k: VIS[k] for k in ("voxel", "k_swell", "d_swell", "camera", "r_max", "r_min")}
        if camera == "equirect":
            kw["camera"] = "s3dis_equirectangular"
            images.depth_maps = visibility_module.render_depth_maps(
                data.pos.to(device), images.pos, None, None, sc["img_size"], img_opk=images.opk, **kw)
        else:
            K = np.tile(np.eye(4, dtype=np.float32), (V, 1, 1))
            K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = sc["fx"], sc["fy"], sc["mx"], sc["my"]
            images.depth_maps = visibility_module.render_depth_maps(
                data.pos.to(device), images.pos, torch.from_numpy(K), images.extrinsic, sc["img_size"], **kw)
        vis = {k: kw[k] for k in ("r_max", "r_min", "camera")}
        t = MapImages(method="DepthBasedVisibility", cylinder=True, depth_threshold=0.05, **vis)
        model = visibility_module.DepthBasedVisibility(img_size=images.proj_size, depth_threshold=0.05, **vis)
    else:
        data.mvf_point_order = kdtree_order(data.pos, True)
        vis = dict(VIS, exact=bool(exact))
        t = MapImages(method="SplattingVisibility", cylinder=True, **vis)
        model = visibility_module.SplattingVisibility(img_size=images.proj_size, **vis)
    plan = t.prepare_fused(data, images, model, device)
    return sc, plan


def survey_model_bytes(N, V, W, Hc, M, F, C, depth=False, m_fov=0, Vw=None):
    """The unfused byte model of the survey: xyz read once per view, one write +
    one read of an 8-byte z-buffer cell per pixel (splat visibility) or one
    4-byte depth gather per in-view point (depth-map visibility), the emitted
    mappings, the output CSR, and each pool reading its input once and writing
    its output once.  Kept for reference: this design keeps the z-buffer in LDS
    and fuses the pools, so it moves less (design_bytes)."""
    P = M                          # mapped pixels; Vw view-level rows (one pixel each unless given)
    Vw = M if Vw is None else Vw
    b = {}
    b["projection"] = V * 12 * N
    b["zbuffer"] = 4 * m_fov if depth else V * 16 * W * Hc
    b["emit"] = M * (8 + 4 + 4 + 4 * F)
    b["csr"] = 8 * (N + 1) + 8 * Vw + 8 * (Vw + 1) + 4 * P + 4 * F * Vw
    b["atomic_pool"] = 4 * C * P + 4 * C * Vw + 8 * (Vw + 1)
    b["view_pool"] = 2 * (4 * C * Vw + 4 * C * N + 8 * (N + 1))     # max and mean
    b["map"] = b["projection"] + b["zbuffer"] + b["emit"] + b["csr"]
    b["total"] = b["map"] + b["atomic_pool"] + b["view_pool"]
    return b


def design_bytes(N, V, M, P, F, C, m_fov, pairs, depth=False, splat=False, map_px=0, entries=0, chunks=0):
    """HBM bytes each stage of THIS design has to move per sample (inputs read once,
    outputs written once; the z-buffer of a tile lives in LDS and costs none):
    N points, V views, M view-level mapping rows, P mapped pixels, m_fov in-view
    (point, view) pairs, `pairs` (splat, tile) entries, F mapping features, C
    pixel-feature channels.  16-byte records: key + box of a pair / an entry.
    Plain splat visibility also has: map_px pixels of the views' id maps, `entries`
    verdict bytes (8 box pixels each) and `chunks` work items of the pixel pass."""
    seen_row = (V + 15) // 16 * 16
    b = {}
    if depth:
        # one pass per point over its views: cloud, ids, one depth per in-view pair, masks out
        b["point_mask"] = 12 * N + 8 * N + 4 * m_fov + 4 * N * ((V + 31) // 32)
    else:
        b["bin_count"] = V * 12 * N + 16 * m_fov + seen_row * N        # cloud per view, records out, flags cleared
        b["bin_fill"] = 16 * m_fov + 16 * pairs                        # records in, entries out
        b["tile_zbuffer"] = 16 * pairs + M                             # entries in, one flag per seen pair out
        if not splat:
            b["centre"] = seen_row * N + 12 * N + 8 * M                # flags, cloud, one table slot per seen pair
            b["point_mask"] = seen_row * N + 8 * N + 12 * N + 8 * M + 4 * N * ((V + 31) // 32)
        else:
            b["tile_zbuffer"] += 4 * map_px                            # the id maps, written once
            b["pixel_boxes"] = 16 * m_fov + M + 16 * M + 16 * chunks   # records and flags in, pair records and chunks out
            b["pixel_mark"] = 4 * map_px + 16 * M + entries            # id maps and pair records in, verdict bytes out (cleared first)
            b["pixel_scan"] = 2 * entries + entries // 2               # verdicts in twice, one count per 8 entries out
            b["pixel_count"] = 16 * m_fov + M + 16 * M + 8 * M + 4 * M
            b["point_mask"] = seen_row * N + 8 * N + 4 * M + 4 * N * ((V + 31) // 32)
            b["pixel_fill"] = 16 * chunks + 16 * M + entries + entries // 2 + 4 * P
    b["csr_fill"] = 4 * N * ((V + 31) // 32) + 8 * N + 12 * N + 8 * (N + 1) + M * (8 + 4 * F) + (
        (8 + 16 + 16) * M if splat else 4 * P)         # plain splat: view pointers out, pair records in and out
    # the fused two-level pool: pixel features read once, max and mean view pools written once
    b["pools"] = 4 * C * P + 2 * 4 * C * N + 8 * (N + 1) + 8 * (M + 1)
    b["total"] = sum(b.values())
    return b


def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mvf_oracle as orc
    return orc


def cpu_baseline(cfg, C_feat, target_s=12.0):
    """The CPU oracle (single-threaded C + numpy port of the reference's CPU
    algorithm) on the same workload, timed on this host."""
    orc = oracle()
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


def pmc_traffic(cfg, key):
    """Measured HBM bytes per launch (rocprofv3 PMC passes, profiles/pmc_traffic.json)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        return json.load(open(tpath)).get(cfg, {}).get(key)
    return None


def qkv_flops(N, Vw, F_main, F_map, C, D, G, inner=32):
    """Multiply-adds x 2 of QKVBimodalCSRPool.forward (pooling.py:454-547) with a
    DeepSetFeat map encoder: E_main, Q on N rows; mlp_elt_1, mlp_elt_2, E_mod, K,
    the compatibilities and the weighted sum on Vw rows; mlp_set on N rows."""
    per_point = F_main * inner + inner * inner + inner * D * G + inner * inner + inner * inner
    per_view = (F_map * inner + inner * inner + 2 * inner * inner + inner * inner
                + C * C + C * C + inner * D * G + D * G + C)
    return 2 * (N * per_point + Vw * per_view)


def pmc_matrix_pipe():
    """Matrix-pipe counters of the C3 forward per kernel (one rocprofv3 --pmc pass,
    profiles/pmc_mfma.json written by profiles/make_mfma.py)."""
    path = os.path.join(ROOT, "profiles", "pmc_mfma.json")
    return json.load(open(path)) if os.path.exists(path) else None


def qkv_roofline(flops, fwd_ms, lanes_ms, bytes_, traffic, N, Vw, Cm):
    """`flops` are those of the module's float32 layers (2 per multiply-add).  `frac` prices
    them against the float32-input MFMA peak as a RATE (fp32-equivalent TFLOP/s); it is not
    the utilisation of one pipe, because with the default arithmetic the value encoder and
    the three-layer key / query chains issue on the bf16 matrix cores (six bf16 products per
    float32 product, float32 accuracy).  `matrix_pipe` gives the utilisation figures: the
    measured busy fraction of the matrix pipe per kernel (PMC) and the time the issued
    products need at the peak of the pipe they run on."""
    split = L.load().mvf_get_mlp_precision() != 0
    bf16_peak = 2516.6e12
    # flops of the chains that run on the bf16 cores when the split is on: E_mod (two 64-wide
    # layers on Vw rows), mlp_elt_2 + K (64->32->32->16) on Vw rows, E_main + Q on N rows
    f_wide = 2 * Vw * (Cm * Cm + Cm * Cm)
    f_narrow3 = 2 * (Vw * (2 * 32 * 32 + 32 * 32 + 32 * 16) + N * (32 * 32 + 32 * 32 + 32 * 16))
    f_bf16 = (f_wide + f_narrow3) if split else 0
    f_f32 = flops - f_bf16
    pipe_s = 6 * f_bf16 / bf16_peak + f_f32 / MFMA_F32_PEAK     # matrix-pipe time at the peaks
    t = fwd_ms * 1e-3
    out = {"bound": "mfma", "kernel": "QKVBimodalCSRPool.forward (all launches)",
           "achieved": flops / t / 1e12, "peak": MFMA_F32_PEAK / 1e12,
           "unit": "TFLOP/s", "frac": flops / t / MFMA_F32_PEAK,
           "rate_note": "fp32-equivalent rate: float32 layer flops / time against the fp32-input MFMA peak; "
                        "see matrix_pipe for pipe utilisation",
           "traffic": traffic, "traffic_ratio": None if traffic is None else traffic / bytes_,
           "flops_per_launch": flops, "kernel_ms": fwd_ms,
           "arithmetic": {"wide_and_three_layer_chains": split_mode(),
                          "two_layer_narrow_chains": "f32 MFMA 32x32x2", "bf16_peak": bf16_peak / 1e12,
                          "flops_on_bf16_cores": f_bf16, "flops_on_f32_cores": f_f32},
           "matrix_pipe": {"issued_bf16_flops": 6 * f_bf16, "bf16_pipe_frac": 6 * f_bf16 / t / bf16_peak,
                           "f32_pipe_frac": f_f32 / t / MFMA_F32_PEAK,
                           "pipe_time_at_peak_ms": pipe_s * 1e3, "pipe_time_frac": pipe_s / t,
                           "busy_per_kernel": pmc_matrix_pipe()},
           "overlapped": {"ms_per_forward": lanes_ms, "achieved": flops / (lanes_ms * 1e-3) / 1e12,
                          "frac": flops / (lanes_ms * 1e-3) / MFMA_F32_PEAK},
           "hbm": {"algorithmic_bytes": bytes_, "frac": bytes_ / t / HBM_PEAK,
                   "measured_frac": None if traffic is None else traffic / t / HBM_PEAK}}
    return out


def split_mode():
    from mvfusion import _lib
    return "bf16 MFMA 32x32x16, operands split 3-way (6 products)" if _lib.load().mvf_get_mlp_precision() \
        else "f32 MFMA 32x32x2"


def qkv_training_step(mod, x_main, x_mod, x_map, ptr, Vw):
    """One training step (forward + backward with batch statistics) of the C3 module and rows
    (left out of the profiler runs, --no-extras: their kernel statistics are the forward's)."""
    mod.train()
    xs = [t.clone().requires_grad_(True) for t in (x_main, x_mod, x_map)]

    def train_step():
        mod.zero_grad(set_to_none=True)
        for t in xs:
            t.grad = None
        mod(xs[0], xs[1], xs[2], ptr).sum().backward()

    for _ in range(3):
        train_step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        train_step()
    ev[1].record()
    torch.cuda.synchronize()
    mod.eval()
    ms = ev[0].elapsed_time(ev[1]) / 10
    return {"what": "QKVBimodalCSRPool in .train(): forward + backward to inputs and parameters, "
                    "BatchNorm batch statistics, same rows as the forward",
            "ms_per_step": ms, "steps": 10, "points_views_per_s": Vw / (ms * 1e-3)}


def run_qkv(args, rank, world, device, dist, cpu_leg=True):
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
        with torch.no_grad():
            for j in range(k):
                with torch.cuda.stream(streams[j % n_lanes]):
                    out = mod(x_main, x_mod, x_map, ptr)
        for s in streams:
            cur.wait_stream(s)
        return out

    # W untimed forwards (+ one untimed gather), then exactly K forwards + the final gather
    # of the pooled features and labels between barrier + synchronise brackets
    keep = {}

    def finish(o):
        keep["out"] = o
        return gather_pooled(o, labels) if dist is not None else o

    elapsed, _ = timed_steps(forwards, args.steps, args.warmup, finish, device)
    out = keep["out"]
    if rank != 0:
        return None
    # launch duration of the whole forward with HIP events on the launch stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    with torch.no_grad():
        for _ in range(args.steps):
            mod(x_main, x_mod, x_map, ptr)
    ev[1].record()
    torch.cuda.synchronize()
    fwd_ms = ev[0].elapsed_time(ev[1]) / args.steps
    flops = qkv_flops(N, Vw, F_main, F_map, Cm, D, G)
    traffic = pmc_traffic("c3", "forward")
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
        "roofline": qkv_roofline(flops, fwd_ms, elapsed / args.steps * 1e3, bytes_, traffic, N, Vw, Cm),
    }
    if not args.no_extras:
        line["training_step"] = qkv_training_step(mod, x_main, x_mod, x_map, ptr, Vw)
    if cpu_leg:
        # the oracle's numpy restatement of the same forward on the first 50k points
        orc = oracle()
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
    return line


def zbuffer_ops(plan):
    """LDS atomic work of the z-buffer pass of the plan's last launch: wave-level
    atomic instructions, lane-level atomics, (splat, tile) entries; and the in-view
    (point, view) pairs of the pass."""
    lib = L.load()
    out = torch.zeros(4, dtype=torch.int64, device=plan.device)
    ws = plan._workspace()
    with torch.cuda.device(plan.device):
        L.check(lib.mvf_profile_zbuffer_ops(
            L.ptr(ws), ws.numel(), plan.N, plan.V, C.byref(plan.params), plan.num_points,
            plan.pair_cap, plan.centre_slots, L.ptr(out), L.stream_ptr(plan.device)))
    return [int(v) for v in out.tolist()]


def run_map_pool(cfg, args, rank, world, device, dist, cpu_leg=True):
    """C1 / C2 / C5 (one sample per step) and C4 (a batch of samples per step):
    fused MapImages + the pixel -> view -> point pools."""
    from mvfusion.mapping import ImageMapping, ImageMappingBatch
    wl = WORKLOADS[cfg]
    C_feat = wl["channels"]
    depth = cfg == "c5"
    n_samples = wl.get("samples", 1)
    n_views = args.views if depth else None
    exact = bool(getattr(args, "exact", 1))
    camera = getattr(args, "camera", "pinhole") if depth else "pinhole"
    staged = [stage_sample(wl["scene"], seed=2 + rank * n_samples + k, device=device, depth=depth,
                           n_views=n_views, exact=exact, camera=camera) for k in range(n_samples)]
    sc, plan = staged[0]
    N, V = plan.N, plan.V
    W, H = sc["img_size"]
    N_all = sum(p.N for _, p in staged)
    # A batch of samples goes through one launch set (mvf_map_images_exact_batch: every
    # point faces the views of its own sample, the output is the collated batch);
    # --batched 0 maps the samples one by one and collates them with torch.
    batched = n_samples > 1 and exact and not depth and bool(getattr(args, "batched", 1))
    if batched:
        from mvfusion.transforms import FusedMapPlan
        plan = FusedMapPlan.batched([p for _, p in staged])
        staged = [(sc, plan)]
        unit = 1                       # samples per plan run: the whole batch is one pass
    else:
        unit = n_samples
    gen = torch.Generator(device=device).manual_seed(rank)
    labels = torch.randint(0, 19, (N_all,), device=device, generator=gen)
    # pixel features for every possible mapping row, resident in HBM
    for _, p in staged:
        p.run()
    splat = plan.splat                  # plain splat visibility: several pixels per (point, view)
    cap = sum(p.pix_cap if splat else p.out_cap for _, p in staged)
    x_pix = torch.rand((cap, C_feat), device=device, generator=gen)
    atomic_ptr = None if splat else torch.arange(cap + 1, dtype=torch.int64, device=device)
    lib = L.load()

    def pools(pointers, n_rows, view_ptr=None, n_pix=None):
        # pixel -> view max, then view -> point max and mean, in one pass over x_pix
        if view_ptr is None:
            view_ptr, n_pix = atomic_ptr[:n_rows + 1], n_rows
        return segment_csr_two_level(x_pix[:n_pix], view_ptr, pointers,
                                     inner="max", outer=("max", "mean"))

    def pools_of(res):
        if splat:
            return pools(res[0], res[1].shape[0], res[5], res[2].shape[0])
        return pools(res[0], res[1].shape[0])

    # Consecutive samples are independent: they alternate between streams (each
    # with its own plan: outputs and workspace), so that the latency-bound
    # kernels of one pass overlap with another pass's.  On each stream the
    # mapping pass of the next sample is enqueued before the sizes of the
    # current one are waited for, so the device does not idle on the host
    # round trip.
    n_lanes = max(1, args.lanes)
    streams = [torch.cuda.Stream(device) for _ in range(n_lanes)]
    if unit == 1:
        plans = [plan] + [plan.twin(k) for k in range(1, n_lanes)]
    else:
        plans = [p for _, p in staged]
        for k, p in enumerate(plans):
            p.lane = k % n_lanes
    # set-up: every plan allocates its outputs and workspace once
    for k, p_ in enumerate(plans):
        with torch.cuda.stream(streams[k % n_lanes]):
            p_.run()
    torch.cuda.synchronize()

    def run_steps_single(k):
        """k full steps (mapping pass, size read-back, pools each)."""
        out = None
        pending = [None] * n_lanes
        cur = torch.cuda.current_stream(device)
        for s in streams:
            s.wait_stream(cur)
        for j in range(min(n_lanes, k)):
            with torch.cuda.stream(streams[j]):
                pending[j] = plans[j].run_async()
        for j in range(k):
            ln = j % n_lanes
            with torch.cuda.stream(streams[ln]):
                nxt = plans[ln].run_async() if j + n_lanes < k else None
                out = pools_of(pending[ln].result())
                pending[ln] = nxt
        for s in streams:
            cur.wait_stream(s)
        return out

    def run_steps_batch(k):
        """k steps of: map every sample of the batch (spread over the streams),
        concatenate the mappings into one ImageMappingBatch (csr.py:347-416,
        image.py:1627-1682), pool the whole batch in one launch.  The mapping
        passes of the next step are enqueued before the current step's sizes
        are waited for (every plan alternates between two output sets)."""
        out = None
        cur = torch.cuda.current_stream(device)
        pool_stream = streams[0]

        def enqueue():
            pend = []
            for p in plans:
                with torch.cuda.stream(streams[p.lane]):
                    pend.append(p.run_async())
            return pend

        for s in streams:
            s.wait_stream(cur)
        pend = enqueue() if k > 0 else None
        for j in range(k):
            nxt = enqueue() if j + 1 < k else None
            maps = []
            for p, pd in zip(plans, pend):
                with torch.cuda.stream(streams[p.lane]):
                    res = pd.result()
                maps.append(ImageMapping.from_csr_arrays(res[0], res[1], res[2], res[3], *res[5:]))
            # result() has waited for every pass of this step: the pools may read them
            with torch.cuda.stream(pool_stream):
                batch = ImageMappingBatch.from_csr_list(maps)
                if splat:
                    out = pools(batch.view_csr_indexing, batch.images.shape[0],
                                batch.atomic_csr_indexing, batch.pixels.shape[0])
                else:
                    out = pools(batch.view_csr_indexing, batch.images.shape[0])
            pend = nxt
        for s in streams:
            cur.wait_stream(s)
        return out

    run_steps = run_steps_single if unit == 1 else run_steps_batch

    def finish(o):
        # the path's only exchange: final gather of pooled features and labels
        return gather_pooled(torch.cat(o[:2], dim=1), labels) if dist is not None else o

    elapsed, _ = timed_steps(run_steps, args.steps, args.warmup, finish, device)
    if rank != 0:
        return None

    # ---- per-kernel timing with HIP events on the launch stream (rank 0) ----
    lib.mvf_profile_enable(1)
    ms = (C.c_float * 16)()
    names = (C.c_char_p * 16)()
    stages = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n_prof = min(args.steps, 50)
    for _ in range(n_prof):
        res = plan.run()
        n = lib.mvf_profile_read(ms, names, 16)
        for i in range(n):
            k = names[i].decode()
            stages[k] = stages.get(k, 0.0) + ms[i]
        ev[0].record()
        pools_of(res)
        ev[1].record()
        torch.cuda.synchronize()
        stages["pools"] = stages.get("pools", 0.0) + ev[0].elapsed_time(ev[1])
    lib.mvf_profile_enable(0)
    stages = {k: v / n_prof for k, v in stages.items()}          # ms per sample
    per = n_samples if batched else 1      # a batched pass launches every kernel once for all samples
    M = plan.last_counts["mappings"] // per                 # view-level rows per sample
    Pn = plan.last_counts["pixels"] if splat else M         # mapped pixels per sample
    pairs = plan.last_counts["pairs"] // per
    instr = lane_ops = entries = m_fov = None
    if not depth:
        # in-view pairs and the z-buffer's LDS-atomic work, counted from what the pass left in
        # its workspace (the op counter reads the tile bins of the exact pass's layout)
        if exact and not batched:
            plan.run()
            instr, lane_ops, entries, m_fov = zbuffer_ops(plan)
        else:
            m_fov = pairs                   # (a lower bound of the in-view pairs is their tile entries)
    else:
        m_fov = M                           # the pairs that pass the depth test are a lower bound of those tested
    D = design_bytes(N, V, M, Pn, plan.F, C_feat, m_fov, pairs, depth=depth, splat=splat, map_px=V * W * H,
                     entries=plan.last_counts.get("verdict_entries", 0), chunks=plan.last_counts.get("chunks", 0))
    S_ = survey_model_bytes(N, V, W, H, Pn, plan.F, C_feat, depth=depth, m_fov=m_fov, Vw=M)
    ms_step = elapsed / args.steps * 1e3
    tcfg = cfg if (exact or depth) else cfg + "_nonexact"      # key of the PMC traffic table

    def hbm(stage):
        """Design bytes and PMC-measured bytes of one launch of a stage against the HBM peak."""
        t = stages[stage] * 1e-3
        traffic = pmc_traffic(tcfg, stage)
        return dict(design_bytes_per_launch=D[stage] * per, frac=D[stage] * per / t / HBM_PEAK,
                    traffic=traffic, measured_frac=None if traffic is None else traffic / t / HBM_PEAK)

    per_stage = {k: dict(hbm(k), kernel_ms=round(stages[k], 4)) for k in D if k in stages}
    dom = max(per_stage, key=lambda k: stages[k])
    step_traffic = pmc_traffic(tcfg, "step")
    pipeline = dict(
        design_bytes_per_step=D["total"] * n_samples, survey_model_bytes_per_step=S_["total"] * n_samples,
        ms_per_step=ms_step, achieved=D["total"] * n_samples / (ms_step * 1e-3) / 1e9,
        frac=D["total"] * n_samples / (ms_step * 1e-3) / HBM_PEAK,
        measured_traffic_per_step=step_traffic,
        measured_frac=None if step_traffic is None else step_traffic / (ms_step * 1e-3) / HBM_PEAK,
        note="design bytes: what the stages of this design must move (inputs once, outputs once, z-buffer "
             "in LDS); survey model: the unfused byte model (z-buffer cells and view-level pool rows in HBM)")
    if dom == "tile_zbuffer" and instr is not None:
        # The z-buffer of a tile lives in LDS: what bounds the kernel is the issue rate of LDS
        # atomics (wave-instructions counted per workload from the tile bins), not HBM.
        zt = stages["tile_zbuffer"] * 1e-3
        roofline = dict(
            bound="lds_atomic", kernel="tile_zbuffer", achieved=instr / zt / 1e9,
            peak=LDS_ATOMIC_PEAK / 1e9, unit="G wave-instr/s", frac=instr / zt / LDS_ATOMIC_PEAK,
            traffic=per_stage[dom]["traffic"], kernel_ms=stages[dom], floor_ms=instr / LDS_ATOMIC_PEAK * 1e3,
            wave_instructions_per_launch=instr, lane_atomics_per_launch=lane_ops,
            splat_tile_entries=entries, lanes_per_instruction=lane_ops / max(instr, 1),
            peak_note="256 CUs x 2.4 GHz / 6.58 cycles per 64-bit ds_min wave-instruction "
                      "(profiles/microbench/lds_ops_bench.hip)",
            hbm=hbm(dom))
    else:
        h = hbm(dom)
        roofline = dict(
            bound="hbm", kernel=dom, achieved=h["design_bytes_per_launch"] / (stages[dom] * 1e-3) / 1e9,
            peak=HBM_PEAK / 1e9, unit="GB/s", frac=h["frac"], traffic=h["traffic"],
            measured_frac=h["measured_frac"], design_bytes_per_launch=h["design_bytes_per_launch"],
            kernel_ms=stages[dom])
        if instr is not None:
            zt = stages["tile_zbuffer"] * 1e-3
            roofline["lds_atomic"] = dict(
                kernel="tile_zbuffer", frac=instr / zt / LDS_ATOMIC_PEAK, kernel_ms=stages["tile_zbuffer"],
                floor_ms=instr / LDS_ATOMIC_PEAK * 1e3, wave_instructions_per_launch=instr)
    roofline["in_view_pairs_per_sample"] = m_fov
    roofline["stages"] = per_stage
    roofline["pipeline"] = pipeline
    roofline["stage_ms"] = {k: round(v, 4) for k, v in stages.items()}

    label = wl["label"]
    if depth:
        label += f" ({V} views per rank)"
        if camera == "equirect":
            label = label.replace("4 MP views", "2048x1024 equirectangular views")
    line = {
        "metric": "points*views mapped+fused/sec",
        "value": world * N_all * V * args.steps / elapsed,
        "unit": "points*views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": label
                   + (", depth-map visibility" if depth else
                      ", exact splat visibility" if exact else ", plain splat visibility (exact=False)")
                   + f" + max atomic pool + max/mean view pools (C={C_feat}, pools fused in one pass)",
                   "name": cfg, "points_per_gpu": N_all, "views": V, "image": [W, H],
                   "samples_per_step": n_samples,
                   "point_order": os.environ.get("MVF_BENCH_POINT_ORDER", "voxel"),
                   "batched_pass": batched,
                   "mappings_per_sample": plan.last_counts["mappings"] // (n_samples if batched else 1),
                   "pixels_per_sample": plan.last_counts["pixels"] if splat else plan.last_counts["mappings"],
                   "splat_tile_pairs": plan.last_counts["pairs"],
                   "streams_per_gpu": n_lanes,
                   "parallelism": f"sample-sharded x{world}, final RCCL all-gather"},
        "roofline": roofline,
    }
    if cpu_leg and not depth and n_samples == 1 and exact:
        line["cpu_baseline"] = cpu_baseline(wl["scene"], C_feat)
    return line


def end_to_end(cfg, device, reps=5):
    """Wall clock of MapImages.__call__(data, images) as a data loader calls it:
    ImageMapping on the device out (point order, staging, kernels, size
    read-back, CSR wrap).  `cold`: a sample the transform has never seen (fresh
    Data and image objects, nothing cached: the local point order is built on the
    host and everything is staged), in a process that has mapped a sample before
    -- what the reference's WildScenes loader pays on every fetch
    (datasets/segmentation/multimodal/wildscenes.py:506-510); `first_call`: the
    very first call of the transform (buffer sizing on top); `steady`: the same
    Data object again (order and staging cached on it, as the reference caches
    its tree)."""
    sc, data, images = host_sample(WORKLOADS[cfg]["scene"], seed=2)
    N, V = data.pos.shape[0], images.num_views
    t = MapImages(method="SplattingVisibility", cylinder=True, output_device=device, **VIS)

    def fresh_images():
        return SameSettingImageData(
            pos=images.pos.clone(), extrinsic=images.extrinsic.clone(), fx=images.fx.clone(),
            fy=images.fy.clone(), mx=images.mx.clone(), my=images.my.clone(),
            ref_size=sc["img_size"], proj_upscale=1)

    def fresh_data(on_device):
        d = Data()
        d.pos = data.pos.clone().to(device) if on_device else data.pos.clone()
        d.mapping_index = data.mapping_index.clone()
        return d

    def call(d, im=None):
        im = im or fresh_images()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t(d, im)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, dict(t.timings)

    first_ms, _ = call(fresh_data(False))
    # The fresh samples are made before the clock starts (a loader's prefetch queue), and the
    # host gets a moment to settle: torch's CPU operators (the clones above) leave their OpenMP
    # workers spinning for a few milliseconds, which would otherwise compete with the host
    # threads of the point-order routine (7 ms instead of 3 ms on the box's 16-CPU share).
    samples = [(fresh_data(False), fresh_images()) for _ in range(reps)]
    time.sleep(0.05)
    cold = []
    for d, im in samples:
        cold.append(call(d, im))
        time.sleep(0.01)
    cold_ms = float(np.median([r[0] for r in cold]))
    cold_t = {k: float(np.median([r[1].get(k, 0.0) for r in cold])) for k in cold[0][1]}
    samples = [(fresh_data(True), fresh_images()) for _ in range(reps)]
    time.sleep(0.05)
    cold_dev_ms = float(np.median([call(d, im)[0] for d, im in samples]))
    d = fresh_data(True)                      # the cloud itself is resident (it feeds the 3D branch too)
    call(d)
    runs = [call(d) for _ in range(reps)]
    steady_ms = float(np.median([r[0] for r in runs]))
    phases = {k: float(np.median([r[1].get(k, 0.0) for r in runs])) for k in runs[0][1]}
    return dict(
        what="MapImages.__call__(data, images) wall clock, one C2 sample, exact splat visibility",
        first_call_ms=first_ms, cold_ms=cold_ms, cold_ms_device_cloud=cold_dev_ms, steady_ms=steady_ms,
        cold_points_views_per_s=N * V / (cold_ms * 1e-3),
        points_views_per_s=N * V / (steady_ms * 1e-3),
        phases_ms={k: round(v, 4) for k, v in phases.items()},
        cold_phases_ms={k: round(v, 4) for k, v in cold_t.items()},
        note="cold = a sample never seen before (host cloud; cold_ms_device_cloud: cloud already in HBM), "
             "median of %d, the fresh sample objects built before the clock starts; t_sphere_sampling = local point order (the library's host KD-tree routine when "
             "cold, cached on data when steady); t_stage = cloud / ids / order / views to the device and "
             "permutation (no reference counterpart); t_visibility = all kernels of the fused pass + read-back "
             "of the sizes, which covers the reference's t_visibility, t_coord_pixels, t_unique_pixels, "
             "t_stack_pixels, t_append; t_ImageMapping_init = wrap of the device CSR arrays" % reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the nested short runs of the other workloads and the end-to-end timing")
    ap.add_argument("--views", type=int, default=16, help="views of the c5 scene held by this rank")
    ap.add_argument("--exact", type=int, default=1,
                    help="0: plain splat visibility (every pixel a point wins is a mapping) for c1 / c2 / c4")
    ap.add_argument("--camera", choices=["pinhole", "equirect"], default="pinhole",
                    help="c5: equirect = S3DIS-style 2048x1024 equirectangular panoramas instead of 4 MP pinhole views")
    ap.add_argument("--batched", type=int, default=1,
                    help="c4: 1 maps the samples of a batch in one launch set, 0 one by one + torch collation")
    ap.add_argument("--lanes", type=int, default=3,
                    help="streams that consecutive samples alternate between (1 = no overlap)")
    args = ap.parse_args()

    rank, world, local = rank_env()
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    cfg = args.config
    cpu_leg = not args.no_cpu_baseline and world == 1      # the CPU baseline is timed at N = 1 only
    if cfg == "c3":
        line = run_qkv(args, rank, world, device, dist, cpu_leg=cpu_leg)
    else:
        line = run_map_pool(cfg, args, rank, world, device, dist, cpu_leg=cpu_leg)

    if rank == 0 and world == 1 and cfg == "c2" and args.exact and not args.no_extras:
        # short runs of the other workloads, so that one invocation records them all
        line["end_to_end"] = end_to_end("c2", device)
        others = {}
        for name in ("c1", "c2_nonexact", "c3", "c4", "c5"):
            torch.cuda.empty_cache()
            short = argparse.Namespace(**vars(args))
            short.steps = 30 if name in ("c3", "c2_nonexact") else 100
            short.warmup = 5
            if name == "c3":
                r = run_qkv(short, 0, 1, device, None, cpu_leg=False)
            elif name == "c2_nonexact":
                # config 2 as BASELINE words it: plain splat visibility (exact=False), every
                # pixel a point wins is a mapping
                short.exact = 0
                r = run_map_pool("c2", short, 0, 1, device, None, cpu_leg=False)
            else:
                r = run_map_pool(name, short, 0, 1, device, None, cpu_leg=False)
            rf = r["roofline"]
            keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio",
                    "measured_frac", "kernel_ms", "floor_ms", "stage_ms", "stages", "lds_atomic", "hbm",
                    "pipeline", "arithmetic", "rate_note", "matrix_pipe", "overlapped")
            others[name] = dict(workload=r["config"]["workload"], value=r["value"], unit=r["unit"],
                                ms_per_step=r["ms_per_step"], steps=r["steps"],
                                points_per_gpu=r["config"].get("points_per_gpu"),
                                views=r["config"].get("views"),
                                mappings_per_sample=r["config"].get("mappings_per_sample"),
                                pixels_per_sample=r["config"].get("pixels_per_sample"),
                                roofline={k: rf[k] for k in keep if k in rf})
            if "training_step" in r:
                others[name]["training_step"] = r["training_step"]
        line["other_workloads"] = others
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
