#!/bin/bash
export TMPDIR=/tmp
for m in 1 2 1 2; do
  NVRX_REPORT_REHOME=$m timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-host-inputs --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('REHOME=$m', 'sync', d['per_step_overhead']['pct'], d['per_step_overhead']['added_us_per_step'], 'async', d['per_step_overhead_async']['pct'], d['per_step_overhead_async']['added_us_per_step'], 'cadence', {k:(v['us_median'], v['us_p95']) for k,v in d['report_at_cadence'].items() if isinstance(v,dict)})"
done
NVRX_REPORT_REHOME=2 timeout 600 python -m pytest tests -m gpu -x -q -k "async or stamps or detector" 2>&1 | tail -3
