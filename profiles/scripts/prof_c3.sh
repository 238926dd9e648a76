# Usage (GPU box, repo root): bash profiles/scripts/prof_c3.sh <tag>
# Kernel-trace statistics of the c3 workload (attentive pool forward), one stream.
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c3_trace -- python3 $root/bench.py --no-cpu-baseline --no-extras --config c3 --steps 20 --warmup 3 --lanes 1 > $out/${tag}_c3_bench.json 2> $out/${tag}_c3_trace.log || { tail -5 $out/${tag}_c3_trace.log; exit 1; }
cp "$(find $out/${tag}_c3_trace -name '*kernel_stats.csv' | head -1)" $out/${tag}_c3_kernel_stats.csv
rm -rf $out/${tag}_c3_trace
python3 - $out/${tag}_c3_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('  %-62s calls %4s avg %9.1f us' % (r['Name'][:62], r['Calls'], float(r['AverageNs']) / 1e3))
PY
