#!/bin/bash
# r03a: where is the slow step of a short timed region; self-launch over gloo; kbench baseline
O=gpurun_out/r03a; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-steps --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs --no-cadence > $O/bench20_$i.log 2>&1
  tail -1 $O/bench20_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['us_per_report_median'], d['per_step_us'], d.get('report_read'))"
done
timeout 300 python tools/outlier_probe.py > $O/probe.log 2>&1; cat $O/probe.log
NVRX_RESIDENT_SCORER=0 timeout 300 python tools/outlier_probe.py > $O/probe_nonres.log 2>&1; tail -40 $O/probe_nonres.log
timeout 300 python bench.py --gpus 2 --backend gloo --steps 50 --warmup 10 --no-cpu-baseline --no-overhead > $O/bench_gloo2.log 2>&1; tail -3 $O/bench_gloo2.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log
for r in 512 64; do timeout 60 tools/kb/kb_base $r 10000 512 0 | grep -E "between=|TOTAL"; done
