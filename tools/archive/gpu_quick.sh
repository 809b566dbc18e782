#!/bin/bash
# quick validation on the GPU box: tools/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-quick}; K=${2:-}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
if [ -n "$K" ]; then
  timeout -s KILL 900 python -m pytest tests -m gpu -x -q -k "$K" --durations=8 > $O/pytest.log 2>&1
else
  timeout -s KILL 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1
fi
echo "pytest rc=$?" >> $O/pytest.log; tail -14 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-steps > $O/bench20.log 2>&1; tail -1 $O/bench20.log > $O/bench20.json
python - <<P
import json
d=json.load(open("$O/bench20.json"))
print("value",d["value"],"median",d["us_per_report_median"],"p95",d["us_per_report_p95"],"steps",d.get("per_step_us"))
print("roofline",{k:d["roofline"][k] for k in ("kernel_us_avg","frac")},"cold",d["roofline"].get("cold",{}).get("kernel_us_avg"))
for k in ("score_kernel","report_read","us_per_report_fully_read","roofline_n8_shape","detector_report","per_step_overhead","per_step_overhead_async","report_at_cadence"):
    print(k, json.dumps(d.get(k))[:600])
P
