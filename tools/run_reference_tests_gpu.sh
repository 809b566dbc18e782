#!/bin/bash
# On an MI355X box: the reference's WHOLE straggler unit suite (13 modules, staged unmodified by
# tools/stage_reference_tests.sh), run against this package with the HIP engine -- once per GPU-timing mode:
#   kernels  per-kernel tracing by name (rocprofiler-sdk): the reference's CUPTI data model, the mode of multi-rank jobs
#   stamp    one GPU-time row per profiled region: the default of single-process jobs
# Logs: $OUT/reference_suite_<mode>.log (+ the process maps of the pytest process, proving which native code ran).
# Usage: tools/run_reference_tests_gpu.sh [out_dir]
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${1:-$REPO/gpurun_out}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT" ;; esac
REF="$REPO/oracle/_ref/reference"
T="$REF/tests/straggler"
mkdir -p "$OUT"
[ -d "$T/unit" ] || { echo "staged reference tests not found at $T/unit (run tools/stage_reference_tests.sh where /root/reference exists)" >&2; exit 2; }
( cd "$T/unit" && sha256sum -c ../SHA256SUMS >/dev/null ) || { echo "staged tests differ from the reference's" >&2; exit 2; }
export NVRX_REPO="$REPO" NVRX_REFTEST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
export PYTHONPATH="$REPO/tools/reftests:$REPO/nvidia-resiliency-ext_amd:$REPO/tests:$REPO"
cd "$T"
rc=0
for mode in kernels stamp; do
  log="$OUT/reference_suite_$mode.log"
  echo "==== reference unit suite, NVRX_GPU_TIMING=$mode, $(date -u +%FT%TZ) ====" > "$log"
  NVRX_GPU_TIMING=$mode NVRX_REFTEST_MAPS="$OUT/reference_suite_${mode}_maps.txt" timeout 1500 \
    python -m pytest -p no:cacheprovider -p nvrx_reftest_plugin -rA -q --timeout=600 unit >> "$log" 2>&1 || rc=1
  tail -n 3 "$log"
done
# The reference's FUNCTIONAL run (tests/straggler/func/ddp_test.py, unmodified): a DDP model on an NCCL (= RCCL) process
# group, Detector.wrap_callables on its forward, a report every 400 iterations, judged by the reference's own
# check_log.py.  One rank here (the box has one GPU; the script hard-codes backend='nccl', and RCCL refuses two ranks on
# one device): it exercises the real RCCL group, DDP's collectives next to the sections and the whole report path, not the
# cross-GPU exchange.
if [ -f "$T/func/ddp_test.py" ] && ( cd "$T" && sha256sum -c SHA256SUMS.func >/dev/null ); then
  for mode in kernels stamp; do
    log="$OUT/reference_func_ddp_$mode.log"
    NVRX_GPU_TIMING=$mode timeout 600 python -m torch.distributed.run --standalone --nnodes=1 --nproc-per-node=1 --local-addr 127.0.0.1 \
      func/ddp_test.py --iters 1500 --report_iter_interval 400 --max_runtime 200 > "$log" 2>&1 || rc=1
    for check in "num_reports --min 3 --max 3" "relative_gpu_stragglers" "individual_gpu_stragglers"; do
      python func/check_log.py --log "$log" $check >> "$log.checks" 2>&1 && echo "func ddp_test [$mode] check_log $check: ok" || { echo "func ddp_test [$mode] check_log $check: FAILED"; rc=1; }
    done
    grep -c "STRAGGLER REPORT" "$log" | sed "s/^/func ddp_test [$mode] reports: /"
  done
fi
exit $rc
