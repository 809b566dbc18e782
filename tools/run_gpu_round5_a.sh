#!/bin/bash
# Round 5, GPU call A: the new per-kernel data path on real hardware -- its tests, the reference's whole unit suite in both
# GPU-timing modes, the tracer's per-dispatch cost with / without dispatch counting, the kernels-mode bench legs.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out
mkdir -p $O
python -c "import torch; print(torch.__version__, torch.cuda.is_available())" > $O/a_env.log 2>&1
echo "== ktrace tests" ; timeout 900 python -m pytest tests/test_gpu_00_ktrace.py tests/test_gpu_01_ktrace_datapath.py -q -s -x > $O/a_ktrace_tests.log 2>&1; echo "rc $?"; tail -n 25 $O/a_ktrace_tests.log
echo "== attached cost"
for cnt in 1 0; do
  NVRX_GPU_TIMING=kernels NVRX_KTRACE_COUNT=$cnt timeout 300 python tools/ktrace_attached_cost.py > $O/a_attached_cost_count$cnt.log 2>&1; tail -n 2 $O/a_attached_cost_count$cnt.log
done
NVRX_GPU_TIMING=stamp timeout 300 python tools/ktrace_attached_cost.py > $O/a_attached_cost_stamp.log 2>&1; tail -n 1 $O/a_attached_cost_stamp.log
echo "== kernels-mode bench child"; timeout 600 python bench.py --child kernels_mode > $O/a_kernels_child.log 2>&1; echo "rc $?"; tail -n 3 $O/a_kernels_child.log | cut -c1-3000
echo "== reference suite"; timeout 1700 bash tools/run_reference_tests_gpu.sh $O; echo "rc $?"
