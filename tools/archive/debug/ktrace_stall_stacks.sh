#!/bin/bash
# Per-kernel mode: rocprofiler-sdk start-up (inside the first HIP call) stalls for minutes on some boxes of the pool.
# The probe runs as the inferior of rocgdb (attaching later is not permitted in the container); if HIP initialisation has
# not returned after 40 s the inferior gets a SIGINT, gdb regains control and prints every thread's native stack.
O=gpurun_out/ktrace_stall/$(date +%s); mkdir -p $O
export NVRX_KTRACE_DEBUG=1
timeout 200 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGINT stop print nopass" -ex "run" \
    -ex "thread apply all bt 40" --args python tools/debug/ktrace_probe.py > $O/gdb.out 2> $O/probe.err &
G=$!
for i in $(seq 1 45); do
  sleep 1
  if grep -q "ready after init" $O/probe.err $O/gdb.out 2>/dev/null; then break; fi
done
if grep -q "ready after init" $O/probe.err $O/gdb.out 2>/dev/null; then
  echo "no stall: HIP + SDK came up within ${i}s"; wait $G
else
  P=$(pgrep -x python | tail -1)
  echo "STALL after 45 s: SIGINT to inferior $P"
  kill -INT $P
  sleep 25
  python3 - $O/gdb.out <<'PY'
import re, sys
txt = open(sys.argv[1], errors="replace").read()
for blk in re.split(r"\n(?=Thread \d+ \()", txt):
    if blk.startswith("Thread") and "blas_thread_server" not in blk:
        print("\n".join(l for l in blk.splitlines() if l.startswith(("Thread", "#")))[:2500])
PY
  kill $G 2>/dev/null
fi
