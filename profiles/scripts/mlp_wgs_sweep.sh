root=$PWD
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in 2 3 4 6 8; do
  MVF_MLP_WGS=$w MVFUSION_SO=$root/multimodal-fusion_amd/mvfusion/libmvfusion_stamp.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sw_$w -- python3 $root/bench.py --no-cpu-baseline --no-extras --config c3 --steps 10 --warmup 2 --lanes 1 > /dev/null 2> $out/sw_$w.log || exit 1
  echo "per_cu $w"
  python3 - "$(find $out/sw_$w -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print('   %-55s avg %9.1f us' % (r['Name'][:55], float(r['AverageNs']) / 1e3))
PY
  rm -rf $out/sw_$w
done
