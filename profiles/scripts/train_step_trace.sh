# Usage (GPU box, repo root): bash profiles/scripts/train_step_trace.sh [group]
# Kernel-trace statistics of QKVBimodalCSRPool training steps at C3 size.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trn -- python3 $root/profiles/scripts/train_step_trace.py $1 > $out/trn.log 2>&1 || { tail -5 $out/trn.log; exit 1; }
cp "$(find $out/trn -name '*kernel_stats.csv' | head -1)" $out/train_step_kernel_stats.csv
rm -rf $out/trn
python3 - $out/train_step_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:30]:
    print("   %-100s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
