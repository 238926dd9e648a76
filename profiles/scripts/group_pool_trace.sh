# Usage (GPU box, repo root): bash profiles/scripts/group_pool_trace.sh
# Kernel-trace statistics of the GroupBimodalCSRPool forward for 1 and 4 groups.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for g in 1 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/gp$g -- python3 $root/profiles/scripts/group_pool_trace.py $g > $out/gp$g.log 2>&1 || { tail -5 $out/gp$g.log; exit 1; }
  echo "groups $g"
  python3 - "$(find $out/gp$g -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("   %-70s calls %3s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out/gp$g
done
