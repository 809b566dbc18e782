#!/bin/bash
# Stages the reference's own straggler unit tests where `gpurun` will carry them to the GPU box: oracle/_ref/ is
# git-ignored (nothing of the reference enters the history) but NOT gpurun-ignored, exactly like the reference-native
# harness built there.  The files are copied byte for byte; tools/run_reference_tests_gpu.sh runs them unmodified.
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
SRC=${NVRX_REFERENCE:-/root/reference}/tests/straggler
[ -d "$SRC/unit" ] || { echo "reference tests not found at $SRC/unit" >&2; exit 2; }
DST="$REPO/oracle/_ref/reference/tests/straggler"
rm -rf "$DST"
mkdir -p "$DST"
cp -r "$SRC/unit" "$DST/unit"
cp -r "$SRC/func" "$DST/func"     # the functional DDP run (ddp_test.py + check_log.py)
find "$DST" -name __pycache__ -type d -prune -exec rm -rf {} +
( cd "$SRC/unit" && sha256sum *.py ) > "$DST/SHA256SUMS"
( cd "$SRC" && sha256sum func/*.py ) > "$DST/SHA256SUMS.func"
echo "staged $(ls "$DST/unit"/test_*.py | wc -l) reference test modules + func/ under $DST"
