#!/bin/bash
# in-bench A/B on one box: round-2 statistics kernel (lib_r02/) vs round-3 (lib/), resident and queued score kernel
export TMPDIR=/tmp
B="--steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-cadence"
for i in 1 2; do
  for d in "" nvidia-resiliency-ext_amd/nvrx_straggler/lib_r02; do
    for m in 2 0; do
      NVRX_LIB_DIR=${d:+$PWD/$d} NVRX_RESIDENT_SCORER=$m timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${d:-lib(r03)}'.split('/')[-1], 'resident=$m', 'value', d['value'], 'median', d['us_per_report_median'], 'kernel512', d['roofline']['kernel_us_avg'], 'cold', d['roofline']['cold']['kernel_us_avg'], 'kernel64', d['roofline_n8_shape']['kernel_us_avg'], 'report64', d['roofline_n8_shape']['report_us_median'], 'score', d['score_kernel']['last_row_to_completion_word_us'])"
    done
  done
done
