#!/bin/bash
# per-step overhead (config #4, synchronous and asynchronous) and the cadence legs with the detector's streams at normal / high priority
TAG=${1:-prio_ab}
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2 3; do
  for pr in normal high; do
    NVRX_STREAM_PRIORITY=$pr timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-inputs --no-extra-legs > $O/b_${pr}_$round.log 2>&1
    python - <<PY
import json
d = json.loads([l for l in open("$O/b_${pr}_$round.log") if l.startswith("{")][-1])
c = d["report_at_cadence"]
print("priority=$pr round $round value", d["value"], "| overhead sync %.2f%% (+%.1f us) async %.2f%% (+%.1f us)" % (d["per_step_overhead"]["pct"], d["per_step_overhead"]["added_us_per_step"], d["per_step_overhead_async"]["pct"], d["per_step_overhead_async"]["added_us_per_step"]),
      "| cadence headline", c["headline_workload"]["us_median"], "sync", c["synchronous"]["us_median"], "async", c["asynchronous"]["us_median"])
PY
  done
done
