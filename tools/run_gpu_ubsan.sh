#!/bin/bash
# UBSan build of both native libraries under the GPU tests of the tracer's data path and the stamp slots (make -C csrc ubsan first)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
SAN=ubsan timeout 800 bash tools/run_sanitized.sh tests/test_gpu_01_ktrace_datapath.py tests/test_gpu_stamps.py tests/test_gpu_00_ktrace.py tests/test_gpu_detector.py -m gpu -s > gpurun_out/ubsan_gpu.log 2>&1; echo "rc $?"
grep -v "ktrace graph\|amdgpu.ids" gpurun_out/ubsan_gpu.log | tail -n 30 | cut -c1-600
