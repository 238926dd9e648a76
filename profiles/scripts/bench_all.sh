set -e
out=gpurun_out/$1; mkdir -p $out
for cfg in c1 c2 c4 c5; do
  timeout -k 10 200 python bench.py --config $cfg --no-extras --no-cpu-baseline > $out/$cfg.json 2>> $out/err
done
timeout -k 10 200 python bench.py --config c2 --exact 0 --no-extras --no-cpu-baseline --steps 30 --warmup 3 > $out/c2_ne.json 2>> $out/err
MVF_BENCH_POINT_ORDER=random timeout -k 10 200 python bench.py --config c2 --no-extras --no-cpu-baseline > $out/c2_random.json 2>> $out/err
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['ms_per_step'],4), '%.3g'%d['value'], d['roofline'].get('stage_ms'))
PY
