set -x
O=gpurun_out/r01d
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench.log 2>&1; tail -1 $O/bench.log
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overhead > $O/prof_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-overhead > $O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-overhead > $O/prof_write.log 2>&1
find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
