# kernel timeline of decode+NMS with two batches in flight (tools/exp_two_streams.py): start/end per kernel and queue
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace2s
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/exp_two_streams.py 12 ${1:-2} > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/trace2s/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if any(k in r['Kernel_Name'] for k in ('score_scan','partition','nms_kernel','finalize','zero_header'))]
rows=rows[-40:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name']
    n='scan' if 'score_scan' in n else 'part' if 'partition' in n else 'nms' if 'nms_kernel' in n else 'fin' if 'finalize' in n else 'zero'
    print('%-5s q%-3s %8.1f -> %8.1f us  (%.1f)' % (n, r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1000, (int(r['End_Timestamp'])-t0)/1000, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000))
PY
