#!/bin/bash
# Long run of the randomised parity sweeps (GPU box, repo root): bash profiles/scripts/soak.sh <out dir>
# A sweep that ends abnormally (fault, time limit) ends the run: nothing is started after it.
out=${1:-gpurun_out/soak}; mkdir -p $out
timeout -k 10 900 python -u profiles/scripts/fuzz_parity.py 3000 20000 > $out/parity.log 2>&1 && tail -1 $out/parity.log && \
timeout -k 10 600 python -u profiles/scripts/fuzz_parity.py 1200 40000 wide > $out/wide.log 2>&1 && tail -1 $out/wide.log && \
timeout -k 10 600 python -u profiles/scripts/fuzz_pools.py 3000 60000 > $out/pools.log 2>&1 && tail -1 $out/pools.log && \
timeout -k 10 900 python -u profiles/scripts/fuzz_attentive.py 700 80000 > $out/attentive.log 2>&1 && tail -1 $out/attentive.log
rc=$?
grep -h "^FAIL" $out/*.log | cut -c1-500 | head -40
grep -c "ILL-CONDITIONED" $out/attentive.log
exit $rc
