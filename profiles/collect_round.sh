#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/collect_round.sh <tag>
# Kernel-trace statistics and bench lines of configs c2 and c3 into gpurun_out/<tag>_*.
# The traces run with --lanes 1 (steps back to back on one stream), so that the kernel
# durations are those of the kernels alone, as the roofline object of the bench line uses
# them; <tag>_c2_lanes3_kernel_stats.csv is the trace of the default three-stream run, whose
# per-kernel durations are inflated by the overlap.
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_${cfg}_trace -- \
    python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --lanes 1 --config $cfg > $root/gpurun_out/${tag}_${cfg}_bench.json 2> $root/gpurun_out/${tag}_${cfg}_trace.log || { echo "trace $cfg failed"; tail -5 $root/gpurun_out/${tag}_${cfg}_trace.log; exit 1; }
  f=$(find $root/gpurun_out/${tag}_${cfg}_trace -name '*kernel_stats.csv' | head -1)
  cp "$f" $root/gpurun_out/${tag}_${cfg}_kernel_stats.csv
  rm -rf $root/gpurun_out/${tag}_${cfg}_trace
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_c2l3_trace -- \
  python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $root/gpurun_out/${tag}_c2_lanes3_bench.json 2> $root/gpurun_out/${tag}_c2l3_trace.log || { echo "trace lanes3 failed"; exit 1; }
f=$(find $root/gpurun_out/${tag}_c2l3_trace -name '*kernel_stats.csv' | head -1)
cp "$f" $root/gpurun_out/${tag}_c2_lanes3_kernel_stats.csv
rm -rf $root/gpurun_out/${tag}_c2l3_trace
cd $root
timeout -k 10 300 python3 bench.py > gpurun_out/${tag}_c2_bench_with_cpu_baseline.json || exit 1
timeout -k 10 300 python3 bench.py --config c3 > gpurun_out/${tag}_c3_bench_with_cpu_baseline.json || exit 1
echo done
