#!/bin/bash
# A/B: kernel-argument lines touched up front in k_score1 (lib/) vs not (lib_notouch/), same box, alternating; resident and queued
export TMPDIR=/tmp
B="--steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs --no-cadence"
for i in 1 2; do
  for d in "" nvidia-resiliency-ext_amd/nvrx_straggler/lib_notouch; do
    for m in 2 0; do
      NVRX_LIB_DIR=${d:+$PWD/$d} NVRX_RESIDENT_SCORER=$m timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${d:-lib(touch)}'.split('/')[-1], 'resident=$m', 'value', d['value'], 'median', d['us_per_report_median'], 'kernel', d['roofline']['kernel_us_avg'], 'score', d['score_kernel']['last_row_to_completion_word_us'], d['score_kernel']['last_row_to_scores_staged_us'])"
    done
  done
done
