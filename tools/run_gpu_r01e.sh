set -x
O=gpurun_out/r01e
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
