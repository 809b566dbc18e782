#!/bin/bash
# Measurement of HEAD's shipped kernel text on one MI355X box: tools/run_gpu_measure.sh <tag>   (run through gpurun)
# driver's bench line, rocprofv3 --kernel-trace --stats, and the two PMC passes (separate runs) of the same command.
set -x
TAG=${1:-measure}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; echo "bench wall $(( $(date +%s) - T0 )) s" | tee $O/bench_wall.txt; tail -1 $O/bench_driver.log > $O/bench_driver.json; cut -c1-600 $O/bench_driver.json
P="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-cadence"
echo "$P" > $O/command.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $P > $O/prof_stats.log 2>&1; echo "stats rc=$?"
cp $O/command.txt $O/stats/command.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $P > $O/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $P > $O/prof_write.log 2>&1; echo "write rc=$?"
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
