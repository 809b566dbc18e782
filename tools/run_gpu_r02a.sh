set -x
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 400 python bench.py --steps 200 --warmup 20 > $O/bench.log 2>&1; tail -1 $O/bench.log
NVRX_SCORE_FENCE=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs > $O/bench_fence.log 2>&1; tail -1 $O/bench_fence.log
NVRX_SCORE_SINGLE_WG=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs > $O/bench_multiwg.log 2>&1; tail -1 $O/bench_multiwg.log
timeout 120 python tools/host_breakdown.py > $O/host_breakdown.log 2>&1; tail -30 $O/host_breakdown.log
for t in 256 512 1024; do timeout 60 tools/bin/kbench_r02 64 10000 $t 0 >> $O/kbench.log 2>&1; done
timeout 60 tools/bin/kbench_r02 512 10000 512 0 >> $O/kbench.log 2>&1
grep "rows=" $O/kbench.log
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
