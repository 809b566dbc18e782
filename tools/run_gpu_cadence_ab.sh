#!/bin/bash
# report at production cadence with the resident scorer on (default rule) and off, alternating: tools/run_gpu_cadence_ab.sh <tag>
TAG=${1:-cadence_ab}
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2 3; do
  for rs in 1 0; do
    NVRX_RESIDENT_SCORER=$rs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs > $O/b_${rs}_$round.log 2>&1
    python - <<PY
import json
d = json.loads([l for l in open("$O/b_${rs}_$round.log") if l.startswith("{")][-1])
c = d["report_at_cadence"]
print("resident=$rs round $round value", d["value"], "| cadence headline", c["headline_workload"]["us_median"], c["headline_workload"]["us_p95"],
      "| sync", c["synchronous"]["us_median"], c["synchronous"]["us_p95"], "| async", {k: v for k, v in c["asynchronous"].items() if "us_" in k})
PY
  done
done
