#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/collect_r03.sh <tag>
# Kernel-trace statistics (rocprofv3 --kernel-trace --stats) of every workload of bench.py, the
# bench lines printed under the profiler, two PMC passes (FETCH_SIZE, WRITE_SIZE) for the HBM
# traffic of c1 / c2 / c2 plain splat / c3 / c4 / c5, one PMC pass with the matrix-pipe counters
# of c3, and the plain default bench line, all into gpurun_out/<tag>_*.
# Traces run with --lanes 1 (samples back to back on one stream): the kernel durations are then
# those of the kernels alone, as the roofline object of the bench line uses them.
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
trace() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${name}_trace -- \
    python3 $root/bench.py --no-cpu-baseline --no-extras "$@" > $out/${tag}_${name}_bench.json 2> $out/${tag}_${name}_trace.log \
    || { echo "trace $name failed"; tail -5 $out/${tag}_${name}_trace.log; exit 1; }
  cp "$(find $out/${tag}_${name}_trace -name '*kernel_stats.csv' | head -1)" $out/${tag}_${name}_kernel_stats.csv
  rm -rf $out/${tag}_${name}_trace
  echo "trace $name done"
}
pmc() {     # name, counters, bench args...
  local name=$1 counter=$2; shift 2
  timeout -s KILL 180 rocprofv3 --pmc $counter --output-format csv -d $out/${tag}_pmc_${name} -- \
    python3 $root/bench.py --no-cpu-baseline --no-extras --lanes 1 "$@" > $out/${tag}_pmc_${name}.log 2>&1 \
    || { echo "pmc $name failed"; tail -5 $out/${tag}_pmc_${name}.log; exit 1; }
  echo "pmc $name done"
}
trace c2 --steps 20 --warmup 3 --lanes 1
trace c2_lanes3 --steps 20 --warmup 3
trace c1 --config c1 --steps 20 --warmup 3 --lanes 1
trace c3 --config c3 --steps 20 --warmup 3 --lanes 1
trace c4 --config c4 --steps 5 --warmup 2
trace c5 --config c5 --steps 10 --warmup 2 --lanes 1
trace c2_nonexact --exact 0 --steps 5 --warmup 2 --lanes 1
cd $root
rm -f $out/${tag}_pmc_traffic.json
for cfg in c1 c2 c3 c4 c5 c2_nonexact; do
  args="--config $cfg"
  [ $cfg = c2_nonexact ] && args="--config c2 --exact 0"
  ( cd /tmp && pmc ${cfg}_FETCH_SIZE FETCH_SIZE $args --steps 3 --warmup 1 && pmc ${cfg}_WRITE_SIZE WRITE_SIZE $args --steps 3 --warmup 1 ) || exit 1
  python3 profiles/make_traffic.py $cfg $out/${tag}_pmc_${cfg}_FETCH_SIZE $out/${tag}_pmc_${cfg}_WRITE_SIZE $out/${tag}_pmc_traffic.json > $out/${tag}_traffic_${cfg}.txt || exit 1
  rm -rf $out/${tag}_pmc_${cfg}_FETCH_SIZE $out/${tag}_pmc_${cfg}_WRITE_SIZE
done
( cd /tmp && pmc c3_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" --config c3 --steps 3 --warmup 1 ) || exit 1
python3 profiles/make_mfma.py $out/${tag}_pmc_c3_mfma $out/${tag}_pmc_mfma.json > $out/${tag}_c3_mfma_counters.txt || exit 1
rm -rf $out/${tag}_pmc_c3_mfma
echo "counters done"
