#!/bin/bash
# Round 5 on one MI355X box (through gpurun): tools/run_gpu_round5.sh [tag]
#   1. the whole -m gpu suite
#   2. the reference's own unit suite (12 modules, staged unmodified by tools/stage_reference_tests.sh) on the HIP engine,
#      once per GPU-timing mode                                   -> reference_suite_{kernels,stamp}.log (+ process maps)
#   3. the driver's bench line                                    -> bench_driver.json
#   4. rocprofv3 --kernel-trace --stats and the two PMC passes of the bench command (tools/run_gpu_measure.sh)
#   5. per-dispatch cost of the attached tracer, section entry costs in per-kernel mode
TAG=${1:-r05}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/$TAG
mkdir -p $O
timeout -s KILL 1800 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 18 $O/pytest.log
timeout 1700 bash tools/run_reference_tests_gpu.sh $O; echo "reference suite rc=$?"
grep -h "^FAILED\|^ERROR" $O/reference_suite_kernels.log $O/reference_suite_stamp.log | cut -c1-160
bash tools/run_gpu_measure.sh $TAG 2>&1 | tail -n 12 | cut -c1-3000
# the N > 1 flow on this ONE GPU (two ranks over gloo share it): not a scaling point, it shows the exchange fields of the line
timeout 600 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs --no-cadence 2>/dev/null | tail -n 1 > $O/bench_gloo_shared_gpu_n2.json; cut -c1-300 $O/bench_gloo_shared_gpu_n2.json
for cnt in 1 0; do
  NVRX_GPU_TIMING=kernels NVRX_KTRACE_COUNT=$cnt timeout 300 python tools/ktrace_attached_cost.py 2>&1 | tail -n 1 > $O/attached_cost_count$cnt.txt; cat $O/attached_cost_count$cnt.txt
done
NVRX_GPU_TIMING=stamp timeout 300 python tools/ktrace_attached_cost.py 2>&1 | tail -n 1 > $O/attached_cost_stamp.txt; cat $O/attached_cost_stamp.txt
NVRX_GPU_TIMING=kernels timeout 300 python tools/probe_ktrace_sections.py 2>&1 | tail -n 1 > $O/section_entry_kernels_mode.txt; cat $O/section_entry_kernels_mode.txt
du -sh $O
