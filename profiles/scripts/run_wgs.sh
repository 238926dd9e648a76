cd /tmp && export TMPDIR=/tmp
for w in 1 2 3 4 64; do
  export MVF_MLP_WGS=$w
  timeout -k 10 100 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wgs_$w -- python3 $GRAFT_REPO_ROOT/profiles/scripts/mlp_ab.py > /dev/null 2>&1 || exit 1
done
