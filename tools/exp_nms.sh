cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3; do
ODBOX_NMS_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nmsdbg$d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --nms-steps 10 --no-cpu-baseline > /dev/null 2>&1
echo "dbg=$d"; grep -h "nms_kernel\|finalize" $GRAFT_REPO_ROOT/gpurun_out/nmsdbg$d/*/*kernel_stats.csv | sed 's/(HIP_vector.*)",/",/' | cut -c1-120
done
