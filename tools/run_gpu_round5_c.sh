#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out
mkdir -p $O
echo "== section probe"; NVRX_GPU_TIMING=kernels timeout 300 python tools/probe_ktrace_sections.py > $O/c_probe_sections.log 2>&1; tail -n 2 $O/c_probe_sections.log | cut -c1-1500
echo "== hang dump"
export NVRX_REPO="$PWD" NVRX_REFTEST=1 PYTHONPATH="$PWD/tools/reftests:$PWD/nvidia-resiliency-ext_amd:$PWD/tests:$PWD"
cd oracle/_ref/reference/tests/straggler
NVRX_GPU_TIMING=kernels NVRX_REFTEST_HANGDUMP_S=12 timeout 200 python -m pytest -p no:cacheprovider -q -x "unit/test_sections.py::test_straggler_sections_detected[test_scenario0]" > $O/c_hang.log 2>&1
grep -n "File \|Thread\|Current thread\|most recent" $O/c_hang.log | head -60
