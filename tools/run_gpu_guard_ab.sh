cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r06guard; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "asynchronous or lane" > $O/pytest_async.log 2>&1; tail -n 3 $O/pytest_async.log
for i in 1 2 3; do
  for eg in 0 1; do
    NVRX_DEBUG_EAGER_GUARD=$eg timeout 300 python tools/cadence_kernels_breakdown.py --async > $O/kbreak_async_eager${eg}_$i.txt 2>&1
    echo "eager=$eg round $i"; grep -E "TOTAL|score kernel launched|nvrx_window_report|inside lane" $O/kbreak_async_eager${eg}_$i.txt
  done
done
