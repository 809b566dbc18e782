set -x
O=gpurun_out/r02c
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_00_ktrace.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 500 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log
