# per-kernel times of one decode+NMS call after the other (tools/exp_nms_single.py) + the timeline of the last call
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_single
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/exp_nms_single.py --groups 3 > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_single'
f=glob.glob(root+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.3: print('%-60s %5s %8.1f us  min %s max %s' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1000, r['MinNs'], r['MaxNs']))
f=glob.glob(root+'/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'score_scan' in r['Kernel_Name']]
start=idx[-2]
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:idx[-1]+1]:
    print('%-28s %8.1f -> %8.1f us' % (r['Kernel_Name'].split('(')[0][-28:], (int(r['Start_Timestamp'])-t0)/1000, (int(r['End_Timestamp'])-t0)/1000))
PY
