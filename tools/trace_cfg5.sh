# kernel timeline of the decode+NMS step (start/end per kernel, relative to the step's first kernel)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace5
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --cfg 5 --iters 3 > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/trace5/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: from the last fillBuffer
idx=[i for i,r in enumerate(rows) if 'fillBuffer' in r['Kernel_Name']]
start=idx[-1]
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    print('%-28s q%-3s %8.1f -> %8.1f us' % (r['Kernel_Name'].split('(')[0][-28:], r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1000, (int(r['End_Timestamp'])-t0)/1000))
PY
