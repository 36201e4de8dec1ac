cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/poleX -- python3 $GRAFT_REPO_ROOT/tools/exp_pole.py > /dev/null 2>&1
