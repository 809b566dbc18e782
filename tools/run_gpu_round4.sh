#!/bin/bash
# round-4 validation on one MI355X box: the whole GPU suite, the driver's bench line, the cadence breakdowns
TAG=${1:-r04}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; tail -1 $O/bench_driver.log > $O/bench_driver.json; cat $O/bench_driver.json
timeout 200 python tools/cadence_detector_breakdown.py > $O/cadence_sync.txt 2>&1; tail -30 $O/cadence_sync.txt
timeout 200 python tools/cadence_detector_breakdown.py --async > $O/cadence_async.txt 2>&1; tail -30 $O/cadence_async.txt
