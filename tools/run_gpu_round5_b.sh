#!/bin/bash
# Round 5, GPU call B: the whole -m gpu suite, the reference's unit suite in both GPU-timing modes, the kernels-mode bench legs.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out
mkdir -p $O
echo "== reference suite"; timeout 1700 bash tools/run_reference_tests_gpu.sh $O; echo "rc $?"
echo "== kernels-mode bench child"; timeout 600 python bench.py --child kernels_mode > $O/b_kernels_child.log 2>&1; echo "rc $?"; tail -n 1 $O/b_kernels_child.log | cut -c1-2500
echo "== gpu tests"; timeout 2400 python -m pytest tests -m gpu -q -x > $O/b_gpu_tests.log 2>&1; echo "rc $?"; tail -n 30 $O/b_gpu_tests.log
