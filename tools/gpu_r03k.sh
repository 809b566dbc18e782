#!/bin/bash
export TMPDIR=/tmp
M="--no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs --no-cadence --dump-steps"
for a in "20 5" "20 5" "20 5" "200 20" "200 20"; do
  set -- $a
  timeout 300 python bench.py --gpus 1 --steps $1 --warmup $2 $M 2>/dev/null | tail -1 | python -c "
import json,sys,numpy as np; d=json.loads(sys.stdin.read()); p=np.array(d['per_step_us']); m=np.median(p)
print('steps',d['steps'],'value',d['value'],'median',round(float(m),2),'outliers(idx,us):',[(int(i),float(p[i])) for i in np.nonzero(p>1.3*m)[0]], 'first5',p[:5].tolist())"
done
timeout 300 python tools/cadence_breakdown.py 2>&1 | grep -v amdgpu.ids
