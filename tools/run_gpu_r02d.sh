set -x
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_gpu_multiproc.py -m gpu -x -q --durations=8 -k "peer" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
