#!/bin/bash
# Reduced refresh of the round-5 evidence after a host-only change (kernel text unchanged: the PMC passes stand):
# pytest -m gpu, the reference unit suite in per-kernel mode, the driver's bench line, rocprofv3 --kernel-trace --stats.
TAG=${1:-r05z}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout -s KILL 900 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log
REF="$PWD/oracle/_ref/reference/tests/straggler"
( export NVRX_REPO="$PWD" NVRX_REFTEST=1 PYTHONPATH="$PWD/tools/reftests:$PWD/nvidia-resiliency-ext_amd:$PWD/tests:$PWD"; cd $REF && \
  echo "==== reference unit suite, NVRX_GPU_TIMING=kernels, $(date -u +%FT%TZ) ====" > $O/reference_suite_kernels.log && \
  NVRX_GPU_TIMING=kernels NVRX_REFTEST_MAPS=$O/reference_suite_kernels_maps.txt timeout 600 python -m pytest -p no:cacheprovider -p nvrx_reftest_plugin -rA -q --timeout=600 unit >> $O/reference_suite_kernels.log 2>&1; tail -n 1 $O/reference_suite_kernels.log )
T0=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; echo "bench wall $(( $(date +%s) - T0 )) s" | tee $O/bench_wall.txt; tail -1 $O/bench_driver.log > $O/bench_driver.json; cut -c1-200 $O/bench_driver.json
P="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-cadence"
rm -rf $O/stats; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $P > $O/prof_stats.log 2>&1; echo "stats rc=$?"; echo "$P" > $O/stats/command.txt
find $O -name "*kernel_trace.csv" -size +3M -delete
