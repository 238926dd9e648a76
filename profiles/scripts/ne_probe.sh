# plain splat visibility (exact=False) on the C2 scene: tests, step time on 3 streams and on one, kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_wide_cameras.py -x -q -m gpu --timeout 180 --timeout-method thread -k "non_exact or exact_false or splat or wide" > gpurun_out/ne_tests.log 2>&1 && \
timeout -k 10 120 python3 bench.py --config c2 --exact 0 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/ne_l3.json 2> gpurun_out/ne_l3.err && \
timeout -k 10 120 python3 bench.py --config c2 --exact 0 --steps 30 --warmup 5 --lanes 1 --no-cpu-baseline --no-extras > gpurun_out/ne_l1.json 2>> gpurun_out/ne_l3.err && \
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ne_prof -o ne -- python3 bench.py --config c2 --exact 0 --steps 5 --warmup 2 --lanes 1 --no-cpu-baseline --no-extras > gpurun_out/ne_trace.log 2>&1
timeout -k 10 120 python3 - > gpurun_out/ne_counts.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "multimodal-fusion_amd")
import bench as B
sc, plan = B.stage_sample("c2", 2, torch.device("cuda", 0), exact=False)
plan.run()
print(plan.last_counts, "mask_cap", plan.mask_cap, "pix_cap", plan.pix_cap)
PY
