#!/bin/bash
# per-kernel mode on one MI355X box: the three ktrace GPU tests, then what the attached tool costs a report
TAG=${1:-ktrace}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_00_ktrace.py -x -q -s > $O/pytest_ktrace.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ktrace.log
grep -E "^\[ktrace|passed|failed|rc=|Error|assert" $O/pytest_ktrace.log | cut -c1-1500 | head -40
B="--steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-cadence --no-extra-legs"
for m in stamp kernels; do
  NVRX_GPU_TIMING=$m timeout 300 python bench.py $B > $O/bench_$m.log 2>&1; tail -1 $O/bench_$m.log > $O/bench_$m.json
  python - $O/bench_$m.json $m <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[2], "value", d["value"], "median", d["us_per_report_median"], "call", d["us_per_call_median"], "kernel", d["roofline"]["kernel_us_avg"])
PY
done
