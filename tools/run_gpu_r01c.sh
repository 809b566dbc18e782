set -x
O=gpurun_out/r01c
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench.log 2>&1; tail -2 $O/bench.log
for a in 0 1 3; do for s in "512 10000" "64 10000" "512 8192"; do timeout 60 tools/bin/kbench_a$a $s; done; done > $O/kbench.log 2>&1
timeout 60 tools/bin/kbench_pc 512 10000 >> $O/kbench.log 2>&1
cat $O/kbench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/prof_stats.log 2>&1
find $O -name "*kernel_trace.csv" -size +5M -delete
du -sh $O
