#!/bin/bash
# Which rocprofiler-sdk service makes the first HIP call load every code object?  (none / api / codeobj name modes)
for m in none api codeobj; do
  NVRX_KTRACE_NAMES=$m timeout 400 python tools/debug/ktrace_eager_load.py tool_$m 2>&1 | tail -2
done
timeout 100 python tools/debug/ktrace_eager_load.py plain 2>&1 | tail -1
