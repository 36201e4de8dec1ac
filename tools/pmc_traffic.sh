# HBM traffic of the two dominant kernels from the L2 memory-side counters, one PMC pass per run
# (MI355X_MICROARCH.md: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 -> separate passes).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --nms-steps 5 --no-cpu-baseline > $OUT/$c.log 2>&1 || exit 1
done
