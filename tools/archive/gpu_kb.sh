#!/bin/bash
# usage (through gpurun): tools/gpu_kb.sh <tag> "<bin> [args]" ...   each command's summary lines are printed
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "### $cmd"
  timeout 120 tools/kb/$cmd > $O/kb_$i.log 2>&1
  grep -E "between=|TOTAL|phase|sub-marks|block total" $O/kb_$i.log
done
