# per-kernel times of the decode+NMS pipeline (cfg 5 per-GPU share)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof5
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --cfg 5 > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof5/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.3: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1000,1), r['MinNs'], r['MaxNs'])
PY
