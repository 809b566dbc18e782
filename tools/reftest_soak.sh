#!/bin/bash
# the reference's sleep-timed unit module (tests/straggler/unit/test_sections.py, read in place, unmodified)
# N times in a row against this package, every run's full output kept -- pytest's fd-level capture holds the children's
# stdout / stderr, -rA prints it for every test -- with PYTHONFAULTHANDLER=1 in every interpreter.  Summary: per run passed /
# failed, and for a failed scenario which assertion / which child exit code.  Build container only (needs /root/reference).
# NVRX_REFTEST=ref runs the CONTROL: the same tests against the reference itself (stub native module), same harness, same load.
# Usage: [NVRX_REFTEST=ref] [NVRX_SOAK_LOAD=6] tools/reftest_soak.sh [runs=30] [out_dir=gpurun_out/reftest_soak] [module=unit/test_sections.py]
REPO="$(cd "$(dirname "$0")/.." && pwd)"
N=${1:-30}
OUT=${2:-$REPO/gpurun_out/reftest_soak}
MOD=${3:-unit/test_sections.py}
REF=${NVRX_REFERENCE:-/root/reference}
[ -d "$REF/tests/straggler/unit" ] || { echo "reference tests not found" >&2; exit 2; }
mkdir -p "$OUT"
make -s -C "$REPO/oracle" >/dev/null
export NVRX_REPO="$REPO" NVRX_REFTEST=${NVRX_REFTEST:-1} PYTHONFAULTHANDLER=1 HSA_ENABLE_IPC_MODE_LEGACY=0
# who calls std::terminate: a preloaded handler prints the native stack and the thread names (tools/reftests/terminate_trace.cpp)
SHIM="$REPO/tools/reftests/_build/libterminate_trace.so"
mkdir -p "$REPO/tools/reftests/_build"
[ -f "$SHIM" ] || g++ -O1 -g -fPIC -shared -o "$SHIM" "$REPO/tools/reftests/terminate_trace.cpp"
[ -f "$SHIM" ] && export LD_PRELOAD="$SHIM${LD_PRELOAD:+:$LD_PRELOAD}"
export PYTHONPATH="$REPO/tools/reftests:$REPO/nvidia-resiliency-ext_amd:$REPO/tests:$REPO"
cd "$REF/tests/straggler"
# NVRX_SOAK_LOAD=n: n busy interpreters beside the runs (the child abort at interpreter exit shows on a LOADED host)
LOADPIDS=""
for _ in $(seq 1 "${NVRX_SOAK_LOAD:-0}"); do python -c "while True: pass" & LOADPIDS="$LOADPIDS $!"; done
trap '[ -n "$LOADPIDS" ] && kill $LOADPIDS 2>/dev/null' EXIT
pass=0; fail=0
for i in $(seq -w 1 "$N"); do
  log="$OUT/run_$i.log"
  python -m pytest -p no:cacheprovider -p nvrx_reftest_plugin -q -rA "$MOD" > "$log" 2>&1
  rc=$?
  tail -n 1 "$log" | sed "s/^/run $i (rc $rc): /"
  if [ $rc -eq 0 ]; then pass=$((pass+1)); else fail=$((fail+1)); grep -n "^FAILED\|^ERROR\|terminate called\|Fatal Python error\|AssertionError\|ret_code\|exitcode\|terminate_trace\|\.so" "$log" | head -n 60 | sed "s/^/    /"; fi
done
echo "reftest soak of $MOD: $pass runs passed, $fail failed (logs: $OUT)"
