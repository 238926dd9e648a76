#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/collect_r02.sh <tag>
# Kernel-trace statistics (rocprofv3 --kernel-trace --stats) of every workload of bench.py, the
# bench lines printed under the profiler, two PMC passes (FETCH_SIZE, WRITE_SIZE) for the HBM
# traffic of c2 / c3 / c5, and the plain default bench line, all into gpurun_out/<tag>_*.
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
}
pmc() {     # name, counter, bench args...
  local name=$1 counter=$2; shift 2
  timeout -s KILL 180 rocprofv3 --pmc $counter --output-format csv -d $out/${tag}_pmc_${name}_${counter} -- \
    python3 $root/bench.py --no-cpu-baseline --no-extras --lanes 1 "$@" > $out/${tag}_pmc_${name}_${counter}.log 2>&1 \
    || { echo "pmc $name $counter failed"; tail -5 $out/${tag}_pmc_${name}_${counter}.log; exit 1; }
}
trace c2 --steps 20 --warmup 3 --lanes 1
trace c2_lanes3 --steps 20 --warmup 3
trace c1 --config c1 --steps 20 --warmup 3 --lanes 1
trace c3 --config c3 --steps 20 --warmup 3 --lanes 1
trace c4 --config c4 --steps 5 --warmup 2
trace c5 --config c5 --steps 10 --warmup 2 --lanes 1
trace c2_nonexact --exact 0 --steps 5 --warmup 2 --lanes 1
for cfg in c2 c3 c4 c5; do
  for counter in FETCH_SIZE WRITE_SIZE; do
    pmc $cfg $counter --config $cfg --steps 3 --warmup 1
  done
done
cd $root
for cfg in c2 c3 c4 c5; do
  python3 profiles/make_traffic.py $cfg $out/${tag}_pmc_${cfg}_FETCH_SIZE $out/${tag}_pmc_${cfg}_WRITE_SIZE $out/${tag}_pmc_traffic.json > $out/${tag}_traffic_${cfg}.txt || exit 1
  rm -rf $out/${tag}_pmc_${cfg}_FETCH_SIZE $out/${tag}_pmc_${cfg}_WRITE_SIZE
done
timeout -k 10 400 python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err || exit 1
echo done
