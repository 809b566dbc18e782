#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | tail -3
for m in 1 0 1 0; do
  echo "### NVRX_REPORT_REHOME=$m"
  NVRX_REPORT_REHOME=$m timeout 300 python tools/cadence_detector_breakdown.py 2>&1 | grep -E "===|TOTAL|nvrx_report|identify"
done
for m in 1 0; do
  NVRX_REPORT_REHOME=$m timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-host-inputs --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('REHOME=$m', 'value', d['value'], 'sync', d['per_step_overhead']['pct'], d['per_step_overhead']['added_us_per_step'], 'async', d['per_step_overhead_async']['pct'], 'cadence', {k:(v['us_median'], v['us_p95']) for k,v in d['report_at_cadence'].items() if isinstance(v,dict)})"
done
