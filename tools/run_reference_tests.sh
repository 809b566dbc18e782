#!/bin/bash
# Runs the reference's own straggler unit tests, UNMODIFIED and read in place from /root/reference, against this
# package (acceptance suite, SURVEY.md section 4 / VERDICT r01 item 1).  CPU box: the checker backend of the repo's
# CPU tests is injected through tools/reftests/sitecustomize.py; GPU box with /root/reference present: the HIP
# backend runs.  Usage: tools/run_reference_tests.sh [extra pytest args]
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
REF=${NVRX_REFERENCE:-/root/reference}
T="$REF/tests/straggler/unit"
[ -d "$T" ] || { echo "reference tests not found at $T" >&2; exit 2; }
make -s -C "$REPO/oracle" >/dev/null
export NVRX_REPO="$REPO" NVRX_REFTEST=1
export PYTHONPATH="$REPO/tools/reftests:$REPO/nvidia-resiliency-ext_amd:$REPO/tests:$REPO${PYTHONPATH:+:$PYTHONPATH}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
# CPU-runnable modules.  Not run here: test_cupti_ext, test_cupti_manager, test_det_section_api, test_reporting, test_reporting_elapsed
# create CUDA tensors / NCCL groups (their twins run on the GPU in tests/test_gpu_detector.py).
MODS="test_relative_gpu_scores.py test_individual_gpu_scores.py test_name_mapper.py test_data_shared.py test_sections.py
test_wrap_callables.py"
cd "$REF/tests/straggler"
ARGS=()
for m in $MODS; do [ -f "unit/$m" ] && ARGS+=("unit/$m"); done
python -m pytest -p no:cacheprovider -q "${ARGS[@]}" "$@"
# test_interval_tracker.py::test_estimate asserts |0.5 s / median(time.sleep(0.01))| within 5 of 50, i.e. a sleep
# overshoot below 10 %: on a loaded / sandboxed host it is timing-sensitive for ANY implementation (the tracker logic is
# the reference's, interval_tracker.py:43-70), so it gets up to 4 attempts and is reported on its own line.
for attempt in 1 2 3 4; do
  if python -m pytest -p no:cacheprovider -q unit/test_interval_tracker.py "$@"; then exit 0; fi
  echo "test_interval_tracker attempt $attempt failed (sleep granularity); retrying" >&2
done
exit 1
