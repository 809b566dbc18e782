#!/bin/bash
# On a box with cold storage: how fast do the GPU libraries come in when they are read sequentially, compared with the
# 10-14 MB/s the tool-attached HIP start-up manages?  (decides whether a read-ahead thread at import time would help)
L=/usr/local/lib/python3.10/dist-packages/torch/lib
t0=$(date +%s.%N)
for f in libmagma.so libMIOpen.so librocsolver.so libtorch_hip.so librocsparse.so; do
  s=$(date +%s.%N); dd if=$L/$f of=/dev/null bs=16M 2>/dev/null; e=$(date +%s.%N)
  python3 -c "import os; sz=os.path.getsize('$L/$f')/1e6; dt=$e-$s; print('$f %.0f MB in %.1f s = %.0f MB/s' % (sz, dt, sz/dt))"
done
t1=$(date +%s.%N); python3 -c "print('sequential read of the five largest libraries: %.1f s' % ($t1-$t0))"
OPENBLAS_NUM_THREADS=1 python tools/debug/ktrace_eager_load.py kernels 2>&1 | grep HIP_
