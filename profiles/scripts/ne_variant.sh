# usage: bash profiles/scripts/ne_variant.sh <variant> ... : kernel times of the plain-splat pass per library variant
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  so=$PWD/multimodal-fusion_amd/mvfusion/libmvfusion${v:+_$v}.so
  MVFUSION_SO=$so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ne_prof_${v:-base} -o ne -- python3 bench.py --config c2 --exact 0 --steps 5 --warmup 2 --lanes 1 --no-cpu-baseline --no-extras > gpurun_out/ne_trace_${v:-base}.log 2>&1 || exit 1
done
