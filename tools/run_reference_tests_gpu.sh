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
exit $rc
