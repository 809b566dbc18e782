#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O; export TMPDIR=/tmp
for r in 64 128; do timeout 120 tools/kb/kb_split $r 10000 | tee -a $O/split.log; done
timeout 600 env SAN=ubsan tools/run_sanitized.sh -m gpu -k "ring or bulk or tracer_records or bad_arguments or test_gpu_detector or stamps or score" > $O/ubsan.log 2>&1; echo "ubsan rc=$?" >> $O/ubsan.log; tail -6 $O/ubsan.log
