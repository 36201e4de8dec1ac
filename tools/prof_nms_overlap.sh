cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_overlap
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/exp_nms_overlap.py $OVERLAP_ARGS > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_overlap'
f=glob.glob(root+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.3: print('%-60s %5s %8.1f us  min %s max %s' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1000, r['MinNs'], r['MaxNs']))
PY
