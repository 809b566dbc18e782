#!/bin/bash
# headline + Detector report with a HIP runtime switch off / on, alternating: tools/run_gpu_env_ab.sh <tag> <VAR> <off> <on>
TAG=${1:-env_ab}; VAR=${2:-HIP_FORCE_DEV_KERNARG}; OFF=${3:-0}; ON=${4:-1}
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2 3; do
  for v in unset $OFF $ON; do
    if [ $v = unset ]; then unset $VAR; else export $VAR=$v; fi
    timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-host-inputs --no-overhead --no-cadence > $O/b_${v}_$round.log 2>&1
    python - <<PY
import json
d = json.loads([l for l in open("$O/b_${v}_$round.log") if l.startswith("{")][-1])
print("$VAR=$v round $round value", d["value"], "median", d["us_per_report_median"], "call", d["us_per_call_median"], "| k_row_stats", d["roofline"]["kernel_us_avg"],
      "| detector", d["detector_report"]["us_median"], "| n8 shape", d["roofline_n8_shape"]["report_us_median"], "| entry", d["section_entry_us"]["profile_cuda_true_us"])
PY
  done
done
