#!/bin/bash
# backtraces of the large reads of the first HIP call with the tool attached (dispatch tracing only), + the env knobs tried
for extra in "" "HIP_ENABLE_DEFERRED_LOADING=1" "HSA_TOOLS_REPORT_LOAD_FAILURE=0"; do
  echo "== env: $extra"
  env $extra NVRX_KTRACE_NAMES=none LD_PRELOAD=$PWD/tools/debug/readtrace.so timeout 300 python tools/debug/ktrace_eager_load.py tool_none 2>&1 | grep -E "readtrace|first HIP" | head -120
done
