#!/bin/bash
# Full GPU validation of HEAD on one MI355X box: tools/run_gpu_round.sh <tag>   (run through gpurun)
# Everything lands in gpurun_out/<tag>/; tools/summarize_profiles.py turns it into profiles/<tag>_* afterwards.
set -x
TAG=${1:-round}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
# the driver's own invocation (20 timed steps after 5 warm-up steps), then the long one
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-steps > $O/bench_driver.log 2>&1; tail -1 $O/bench_driver.log > $O/bench_driver.json; cat $O/bench_driver.json
B="--steps 200 --warmup 20"
timeout 600 python bench.py $B --no-cpu-baseline --no-host-inputs > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; cat $O/bench.json
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# A/B on the same box: the score kernel always resident on a stream of its own (2) / always queued behind the statistics
# kernel (0); the default run above lets the library choose per report
for m in 2 0; do
  n=$([ $m = 2 ] && echo resident || echo nonresident)
  NVRX_RESIDENT_SCORER=$m timeout 300 python bench.py $B --no-cpu-baseline --no-host-inputs --no-overhead --no-cadence --no-extra-legs > $O/bench_$n.log 2>&1
  tail -1 $O/bench_$n.log > $O/bench_$n.json; cat $O/bench_$n.json
done
# the multi-rank flow with ranks SHARING this GPU, launched the way the driver launches N > 1 (no launcher: bench.py
# spawns its ranks): gloo group, once with the host-hop exchange and once through IPC peer windows
for n in 2 4; do
  timeout 300 python bench.py --gpus $n $B --backend gloo --no-cpu-baseline --no-overhead > $O/bench_gloo_n$n.log 2>&1
  grep '^{"metric' $O/bench_gloo_n$n.log | tail -1 > $O/bench_gloo_n$n.json; cat $O/bench_gloo_n$n.json
  NVRX_EXCHANGE=peer NVRX_REPORT_TIMEOUT_S=30 timeout 300 python bench.py --gpus $n $B --backend gloo --no-cpu-baseline --no-overhead > $O/bench_gloo_peer_n$n.log 2>&1
  grep '^{"metric' $O/bench_gloo_peer_n$n.log | tail -1 > $O/bench_gloo_peer_n$n.json; cat $O/bench_gloo_peer_n$n.json
done
P="python bench.py $B --no-cpu-baseline --no-overhead --no-host-inputs --no-cadence"
echo "$P" > $O/command.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $P > $O/prof_stats.log 2>&1
cp $O/command.txt $O/stats/command.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $P > $O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $P > $O/prof_write.log 2>&1
# the cadence leg under the profiler (one report per 100 training steps): what the report's kernels cost when they are cold
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cadence -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs --cadence-reports 10 > $O/prof_cadence.log 2>&1
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
