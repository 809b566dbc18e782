#!/bin/bash
# Round 6 on one MI355X box (through gpurun): tools/run_gpu_round6.sh [tag]
#   1. the whole -m gpu suite
#   2. the driver's bench line (N = 1) + the N = 2 flow on this ONE GPU (two gloo ranks sharing it: the route table's shape)
#   3. rocprofv3 --kernel-trace --stats and the two PMC passes of the bench command (tools/run_gpu_measure.sh)
#   4. the per-kernel-mode report taken apart: lane on / off, synchronous / asynchronous, callback / buffered delivery
# (Rounds 4-5 also shipped the reference's own unit tests to the box -- oracle/_ref/reference, tools/stage_reference_tests.sh --
#  and ran them there.  The build rules say a PYTHON reference must not travel to the GPU box in any form; that staging is
#  gone, its logs stay under profiles/r05*_reference_suite_* as history, and the GPU-only reference tests have own-code twins
#  in tests/test_gpu_00_ktrace.py / test_gpu_detector.py.)
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/$TAG
mkdir -p $O
timeout -s KILL 1800 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 18 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log
bash tools/run_gpu_measure.sh $TAG 2>&1 | tail -n 12 | cut -c1-3000
timeout 900 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-overhead --no-host-inputs --no-extra-legs --no-cadence --cpu-reps 9 2>/dev/null | tail -n 1 > $O/bench_gloo_shared_gpu_n2.json; cut -c1-400 $O/bench_gloo_shared_gpu_n2.json
for args in "" "--async" "--no-lane" "--no-lane --async"; do
  name=kbreak$(echo "$args" | tr -d ' ' | tr -- '-' '_')
  timeout 300 python tools/cadence_kernels_breakdown.py $args > $O/$name.txt 2>&1; grep -E "^===|TOTAL|inside lane" $O/$name.txt
done
NVRX_DEBUG_KTRACE_DELIVERY=buffer timeout 300 python tools/cadence_kernels_breakdown.py --no-lane > $O/kbreak_no_lane_buffered_delivery.txt 2>&1; grep -E "^===|TOTAL|profiler.harvest" $O/kbreak_no_lane_buffered_delivery.txt
NVRX_GPU_TIMING=kernels timeout 300 python tools/ktrace_attached_cost.py 2>&1 | tail -n 1 > $O/attached_cost.txt; cat $O/attached_cost.txt
du -sh $O
