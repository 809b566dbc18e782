#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out
mkdir -p $O
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_00_ktrace.py tests/test_gpu_01_ktrace_datapath.py tests/test_gpu_stamps.py tests/test_gpu_score.py -q -x -s 2>&1 | grep -v "^\[ktrace graph\]" | tail -n 25 | cut -c1-600
echo "== one matmul, its kernels"; NVRX_GPU_TIMING=kernels timeout 200 python - <<'PY' 2>&1 | tail -n 6 | cut -c1-900
import sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "nvidia-resiliency-ext_amd"), os.getcwd()]
import nvrx_cupti_module as m
import torch
p = m.CuptiProfiler()
a = torch.randn(1000, 1000, device="cuda"); b = torch.randn(1000, 1000, device="cuda"); torch.cuda.synchronize()
p.initialize(); p.start(); torch.matmul(a, b); torch.cuda.synchronize(); p.stop()
for k, v in p.get_stats().items():
    print("KEY", k[:40] + "..." + k[-60:], v.num_calls, round(v.median, 2))
PY
echo "== reference suite"; timeout 1700 bash tools/run_reference_tests_gpu.sh $O; echo "rc $?"
grep -h "^FAILED\|^ERROR" $O/reference_suite_kernels.log | cut -c1-150
echo "== kernels-mode bench child"; timeout 600 python bench.py --child kernels_mode > $O/e_kernels_child.log 2>&1; echo "rc $?"; tail -n 1 $O/e_kernels_child.log | cut -c1-4000
