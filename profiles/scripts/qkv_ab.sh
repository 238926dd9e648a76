# Usage (GPU box, repo root): bash profiles/scripts/qkv_ab.sh <variant> [<variant> ...]
# C3 forward alone (one stream) for the production library and for variants of it.
for rep in 1 2; do
  for v in base "$@"; do
    so=$PWD/multimodal-fusion_amd/mvfusion/libmvfusion.so
    [ $v != base ] && so=$PWD/multimodal-fusion_amd/mvfusion/libmvfusion_$v.so
    MVFUSION_SO=$so timeout -k 10 120 python bench.py --config c3 --lanes 1 --no-extras --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'forward alone %.4f ms' % d['roofline']['kernel_ms'], flush=True)"
  done
done
