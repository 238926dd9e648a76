# step time at the short run length the round-end bench uses (20 steps), against the number of streams
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for L in 2 3 4 5; do
    timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --lanes $L --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes', $L, round(d['ms_per_step'],4))" || exit 1
  done
done
