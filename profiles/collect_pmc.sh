#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/collect_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
# BENCH_ARGS adds arguments to bench.py (e.g. BENCH_ARGS='--config c3').
# One rocprofv3 --pmc run per pass (counters are not split over passes), csv output under gpurun_out/pmc_<tag>_<i>.
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for counters in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d $root/gpurun_out/pmc_${tag}_$i -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $root/gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $root/gpurun_out/pmc_${tag}_$i.log; exit 1; }
done
echo done
