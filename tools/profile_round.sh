# One GPU call that refreshes everything under profiles/ for the current tree:
#   bench line, all-config table, rocprofv3 kernel stats of the bench command, PMC traffic passes
#   (FETCH_SIZE and WRITE_SIZE, one run each, per mode of the headline and for the loss / scan kernels).
# Run on the GPU box from the repo root:  bash tools/profile_round.sh ; then python tools/pmc_to_json.py gpurun_out/round/pmc
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT          # (gpurun merges into the local gpurun_out/: delete the local gpurun_out/round before the call, too)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 240 python3 tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-nms-pipeline > $OUT/stats.log 2>&1 || exit 1
# the headline alone: every match_encode_kernel<true, false> launch of this run is a timed headline step
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-configs --no-loss --no-nms > $OUT/stats_headline.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  for m in stream auto resident; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/$m/$c -- python3 $GRAFT_REPO_ROOT/tools/pmc_headline.py $m > $OUT/pmc_${m}_$c.log 2>&1 || exit 1
  done
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/rest/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --nms-steps 5 --loss-steps 5 --no-cpu-baseline --no-configs --no-nms-pipeline > $OUT/pmc_rest_$c.log 2>&1 || exit 1
done
echo done
