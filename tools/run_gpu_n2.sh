#!/bin/bash
# bench.py as the driver launches it at N > 1, on a box with ONE GPU (ranks share it, gloo): tools/run_gpu_n2.sh <tag>
# The default GPU timing mode of a multi-rank job is per-kernel tracing; the second run pins region stamps for comparison.
TAG=${1:-n2}
O=gpurun_out/$TAG
mkdir -p $O
for mode in default stamp; do
  for n in 2 4; do
    if [ $mode = stamp ]; then export NVRX_GPU_TIMING=stamp; else unset NVRX_GPU_TIMING; fi
    timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
        bench.py --gpus $n --steps 20 --warmup 5 --backend gloo > $O/bench_${mode}_n$n.log 2>&1
    echo "mode=$mode n=$n rc=$?"
    grep '^{' $O/bench_${mode}_n$n.log | tail -1 > $O/bench_gloo_${mode}_n$n.json
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_gloo_${mode}_n$n.json"))
    print({k: d.get(k) for k in ("value", "n_gpus", "ranks", "gpu_timing_mode")}, json.dumps(d.get("per_step_overhead"))[:400])
    print(json.dumps(d.get("exchange"))[:600])
except Exception as e:
    print("no line:", e)
PY
  done
done
tail -5 $O/bench_default_n2.log | cut -c1-400
