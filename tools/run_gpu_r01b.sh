set -x
mkdir -p gpurun_out/r01b
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r01b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r01b/pytest.log
tail -5 gpurun_out/r01b/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r01b/bench.log 2>&1; tail -2 gpurun_out/r01b/bench.log
for a in 0 1 3; do for s in "512 10000" "64 10000" "512 8192"; do timeout 60 tools/bin/kbench_a$a $s; done; done > gpurun_out/r01b/kbench.log 2>&1
cat gpurun_out/r01b/kbench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/stats -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r01b/prof_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01b/pmc_fetch -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r01b/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01b/pmc_write -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r01b/prof_write.log 2>&1
find gpurun_out/r01b -name "*.csv" | head; du -sh gpurun_out/r01b
# drop the big traces, keep stats + counter csvs
find gpurun_out/r01b -name "*kernel_trace.csv" -size +5M -delete
