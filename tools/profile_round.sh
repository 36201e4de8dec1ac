# One GPU call that refreshes everything under profiles/ for the current tree:
#   bench line, rocprofv3 kernel stats of the same command, PMC traffic passes, all-config table.
# Run on the GPU box from the repo root:  bash tools/profile_round.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 240 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 240 python3 tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --nms-steps 5 --loss-steps 5 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1 || exit 1
done
echo done
