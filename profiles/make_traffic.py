"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
written to profiles/pmc_traffic.json for bench.py's roofline.traffic field.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request
as 64 bytes for wide streaming reads, so it is doubled (an upper bound for
narrow reads).
usage: python profiles/make_traffic.py <cfg> <fetch_dir> <write_dir> [output.json]
(default output: profiles/pmc_traffic.json).  "step" = the kernels of one bench step: the
per-launch average of every kernel of the workload's pipeline times its launches per step
(kernels that only run while the inputs are set up, such as the rendering of the depth maps
of c5, are not part of a step); "forward" = one QKVBimodalCSRPool forward."""
import collections
import csv
import glob
import json
import os
import sys

# kernels of one step of a mapping workload and their launches per step
EXACT = {"bin_count_kernel": 1, "tile_scan_kernel": 1, "bin_fill_kernel": 1, "tile_zbuffer_kernel": 1,
         "mi_centre_kernel": 1, "mi_point_mask_kernel": 1, "mi_fill_kernel": 1, "two_level_pool_kernel": 1}
DEPTH = {"mi_point_mask_kernel": 1, "mi_fill_kernel": 1, "two_level_pool_kernel": 1}
SPLAT = {"bin_count_kernel": 1, "tile_scan_kernel": 1, "bin_fill_kernel": 1, "tile_zbuffer_kernel": 1,
         "ne_box_kernel": 1, "ne_mark_kernel": 1, "ne_bits_reduce_kernel": 1, "scan_spine_kernel": 1,
         "ne_bits_apply_kernel": 1, "ne_count_kernel": 1, "ne_point_kernel": 1, "ne_fill_kernel": 1,
         "ne_emit_kernel": 1, "segment_long_kernel": 1, "two_level_pool_kernel": 1}
PIPELINE = {"c1": EXACT, "c2": EXACT, "c4": EXACT, "c5": DEPTH, "c2_nonexact": SPLAT}

STAGE = {"bin_count_kernel": "bin_count", "tile_scan_kernel": "tile_scan", "bin_fill_kernel": "bin_fill",
         "tile_zbuffer_kernel": "tile_zbuffer", "mi_centre_kernel": "centre",
         "mi_point_mask_kernel": "point_mask", "mi_fill_kernel": "csr_fill",
         "two_level_pool_kernel": "pools", "ne_point_kernel": "point_mask", "ne_fill_kernel": "csr_fill",
         "ne_box_kernel": "pixel_boxes", "ne_mark_kernel": "pixel_mark", "ne_count_kernel": "pixel_count",
         "ne_emit_kernel": "pixel_fill", "segment_long_kernel": "pools", "ne_bits_reduce_kernel": "pixel_scan",
         "ne_bits_apply_kernel": "pixel_scan"}


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "mvf::" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("mvf::")[1].split("<")[0].split("(")[0]
                acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return acc


cfg, fdir, wdir = sys.argv[1:4]
out_path = sys.argv[4] if len(sys.argv) > 4 else None
fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
out = {}
detail = {}
for k in sorted(set(fetch) | set(write)):
    f = 2.0 * sum(fetch[k]) / max(len(fetch[k]), 1)
    w = sum(write[k]) / max(len(write[k]), 1)
    detail[k] = dict(fetch_bytes_x2=f, write_bytes=w, launches_sampled=len(fetch[k]))
    out[STAGE.get(k, k)] = out.get(STAGE.get(k, k), 0.0) + f + w      # (a stage may be several kernels)
# one QKVBimodalCSRPool.forward = all launches between two score launches
nfwd = len(fetch.get("attention_scores_kernel", []))
total = sum(2.0 * sum(fetch[k]) + sum(write[k]) for k in set(fetch) | set(write))
if nfwd:
    out["forward"] = total / nfwd
if not nfwd:
    pipe = PIPELINE.get(cfg, EXACT)
    per_launch = lambda k: 2.0 * sum(fetch[k]) / max(len(fetch[k]), 1) + sum(write[k]) / max(len(write[k]), 1)
    out["step"] = sum(per_launch(k) * mult for k, mult in pipe.items() if k in fetch or k in write)
    out["step_kernels"] = sorted(k for k in pipe if k in fetch or k in write)
path = out_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
data[cfg] = out
data[cfg + "_detail"] = detail
json.dump(data, open(path, "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if not isinstance(v, list):
        print(f"{k:28s} {v / 1e6:10.2f} MB per launch")
