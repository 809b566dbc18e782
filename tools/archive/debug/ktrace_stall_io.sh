#!/bin/bash
# Is the slow start of per-kernel mode (first HIP call with the rocprofiler-sdk tool registered) I/O?  Sample the probe's
# process state, wait channel and bytes read from storage every 5 s until HIP + SDK are up.
O=gpurun_out/ktrace_stall/io_$(date +%s); mkdir -p $O
NVRX_KTRACE_DEBUG=1 OPENBLAS_NUM_THREADS=1 timeout 240 python tools/debug/ktrace_probe.py > $O/probe.out 2> $O/probe.err &
sleep 1
P=$(pgrep -x python | tail -1)
for i in $(seq 1 40); do
  if grep -q "ready after init" $O/probe.err; then break; fi
  st=$(ps -o stat=,wchan:24= -p $P 2>/dev/null)
  rb=$(grep -E "^read_bytes|^rchar" /proc/$P/io 2>/dev/null | tr '\n' ' ')
  nmaps=$(wc -l < /proc/$P/maps 2>/dev/null)
  last=$(tail -1 $O/probe.err | cut -c1-60)
  echo "t=$((i*5-4))s state/wchan: $st | $rb | maps $nmaps | last probe line: $last"
  sleep 5
done
wait
grep -E "ready after init|exiting" $O/probe.err | head -3
grep -E "rocprof|comgr|hsa-runtime|amdhip" /proc/self/maps | head -0
python3 - <<'PY'
import os
for lib in ("/opt/rocm/lib/librocprofiler-sdk.so", "/opt/rocm/lib/libamd_comgr.so", "/opt/rocm/lib/libhsa-runtime64.so", "/opt/rocm/lib/libamdhip64.so", "/opt/rocm/lib/librocprofiler-register.so"):
    for p in (lib, lib + ".1", lib + ".0", lib + ".3", lib + ".7"):
        if os.path.exists(p):
            print(p, os.path.getsize(os.path.realpath(p)) // (1 << 20), "MiB")
            break
PY
