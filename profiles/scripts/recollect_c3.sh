#!/bin/bash
# c3 part of profiles/collect_r03.sh for tag r03_d, keeping the other workloads' entries
set -u
tag=r03_d
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cp $root/profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c3_trace -- python3 $root/bench.py --no-cpu-baseline --no-extras --config c3 --steps 20 --warmup 3 --lanes 1 > $out/${tag}_c3_bench.json 2> $out/${tag}_c3_trace.log || exit 1
cp "$(find $out/${tag}_c3_trace -name '*kernel_stats.csv' | head -1)" $out/${tag}_c3_kernel_stats.csv
rm -rf $out/${tag}_c3_trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_c3_$c -- python3 $root/bench.py --no-cpu-baseline --no-extras --lanes 1 --config c3 --steps 3 --warmup 1 > $out/${tag}_pmc_c3_$c.log 2>&1 || exit 1
done
cd $root
python3 profiles/make_traffic.py c3 $out/${tag}_pmc_c3_FETCH_SIZE $out/${tag}_pmc_c3_WRITE_SIZE $out/${tag}_pmc_traffic.json > $out/${tag}_traffic_c3.txt || exit 1
rm -rf $out/${tag}_pmc_c3_FETCH_SIZE $out/${tag}_pmc_c3_WRITE_SIZE
cd /tmp
timeout -s KILL 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_c3_mfma -- python3 $root/bench.py --no-cpu-baseline --no-extras --lanes 1 --config c3 --steps 3 --warmup 1 > $out/${tag}_pmc_c3_mfma.log 2>&1 || exit 1
cd $root
python3 profiles/make_mfma.py $out/${tag}_pmc_c3_mfma $out/${tag}_pmc_mfma.json > $out/${tag}_c3_mfma_counters.txt || exit 1
rm -rf $out/${tag}_pmc_c3_mfma
timeout -k 10 300 python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err || exit 1
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_args.json 2>/dev/null || exit 1
echo c3-done
