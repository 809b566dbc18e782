#!/bin/bash
O=gpurun_out/r03f; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -9 $O/pytest.log
timeout 600 tools/run_sanitized.sh -m gpu -k "ring or bulk or tracer_records or bad_arguments or test_gpu_detector or stamps" > $O/asan.log 2>&1; echo "asan rc=$?" >> $O/asan.log; tail -12 $O/asan.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-host-inputs --no-overhead --no-cadence > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json
python - <<P
import json
d=json.load(open("$O/bench.json"))
print("value",d["value"],"median",d["us_per_report_median"]); print(d.get("per_kernel_mode"))
P
