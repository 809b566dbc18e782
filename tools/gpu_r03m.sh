#!/bin/bash
export TMPDIR=/tmp
B="--steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-cadence"
L=nvidia-resiliency-ext_amd/nvrx_straggler
for i in 1 2 3; do
  for d in lib lib_r02 lib_vA lib_vC lib_vB lib_vAC lib_vABC; do
      NVRX_LIB_DIR=$PWD/$L/$d timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$d', 'value', d['value'], 'median', d['us_per_report_median'], 'kernel512', d['roofline']['kernel_us_avg'], 'cold', d['roofline']['cold']['kernel_us_avg'], 'kernel64', d['roofline_n8_shape']['kernel_us_avg'], 'report64', d['roofline_n8_shape']['report_us_median'])"
  done
done
