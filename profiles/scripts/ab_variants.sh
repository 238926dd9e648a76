# Usage (GPU box, repo root): bash profiles/scripts/ab_variants.sh <config> <lanes list> <variant> [<variant> ...]
# Step time of bench.py for the production library and for instrumented / experimental variants
# (libmvfusion_<variant>.so, built with `make variant V=<variant> EXTRA=...`), interleaved so
# that drift of the box shows up in both.
cfg=$1; lanes=$2; shift 2
out=gpurun_out/ab
mkdir -p $out
for rep in 1 2; do
  for v in base "$@"; do
    for L in $lanes; do
      so=$PWD/multimodal-fusion_amd/mvfusion/libmvfusion.so
      [ $v != base ] && so=$PWD/multimodal-fusion_amd/mvfusion/libmvfusion_$v.so
      MVFUSION_SO=$so timeout -k 10 120 python bench.py --config $cfg --lanes $L --no-extras --no-cpu-baseline --steps 300 --warmup 20 > $out/${v}_L${L}_$rep.json 2> $out/err.log || { echo "$v lanes $L failed"; tail -3 $out/err.log; exit 1; }
      python - $out/${v}_L${L}_$rep.json $v $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:12s} lanes {sys.argv[3]}: {d['ms_per_step']:.4f} ms/step  zb {d['roofline']['stage_ms'].get('tile_zbuffer')}", flush=True)
PY
    done
  done
done
