set -e
mkdir -p gpurun_out/r2k
for L in 1 2 3 4 6; do
  timeout -k 10 120 python bench.py --lanes $L --no-extras --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r2k/lanes$L.json
done
for L in 3 4; do
  MVFUSION_SO=$PWD/multimodal-fusion_amd/mvfusion/libmvfusion_wg2.so timeout -k 10 120 python bench.py --lanes $L --no-extras --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r2k/wg2_lanes$L.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2k/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'])
PY
