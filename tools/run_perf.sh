timeout -k 10 400 python -u -m pytest tests/test_gpu_scan_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tp.log 2>&1; tail -3 gpurun_out/tp.log
BZ_PROFILE=0 timeout -k 10 120 python tools/quick_perf.py > gpurun_out/qp.txt 2>&1; grep "back-to-back" gpurun_out/qp.txt
export TMPDIR=/tmp; rm -rf gpurun_out/trq; cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trq -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python3 tools/trace_summary.py gpurun_out/trq 10 8
