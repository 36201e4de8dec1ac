# per-kernel durations of decode+NMS with 16-bit and fp32 scores (tools/exp_scan16.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof16
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/exp_scan16.py > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,os
f=sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof16/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'at::' in n: continue
    print('%-90s calls %4s avg %8.1f min %8.1f max %8.1f' % (n[:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
