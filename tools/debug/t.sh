export TMPDIR=/tmp
B="--steps 200 --warmup 20 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs"
NVRX_EXCHANGE=peer NVRX_REPORT_TIMEOUT_S=30 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 $B --backend gloo 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gloo+peer n2', d['value'], 'median', d['us_per_report_median'], 'p95', d['us_per_report_p95'], 'kernel', d['roofline']['kernel_us_avg'], 'exch', d['exchange']['us_median'])"
timeout -s KILL 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_00_ktrace.py 2>&1 | grep -E "passed|failed" | tail -3
for r in 1 0; do NVRX_RESIDENT_SCORER=$r timeout 200 python bench.py $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n1 resident=$r', d['value'], 'median', d['us_per_report_median'], 'kernel', d['roofline']['kernel_us_avg'])"; done
