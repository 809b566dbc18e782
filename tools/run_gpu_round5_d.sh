#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out
mkdir -p $O
echo "== reference suite"; timeout 1700 bash tools/run_reference_tests_gpu.sh $O; echo "rc $?"
grep -h "^FAILED\|^ERROR" $O/reference_suite_kernels.log | cut -c1-150
echo "-- stamp"; grep -h "^FAILED\|^ERROR" $O/reference_suite_stamp.log | cut -c1-150
echo "== async test"; timeout 600 python -m pytest tests/test_gpu_01_ktrace_datapath.py -q -s -x 2>&1 | tail -n 6 | cut -c1-400
for v in "1 1" "0 1" "1 0"; do set -- $v
  echo "== kernels-mode bench child COUNT=$1 PUMP=$2"; NVRX_KTRACE_COUNT=$1 NVRX_KTRACE_PUMP=$2 timeout 600 python bench.py --child kernels_mode > $O/d_kernels_child_$1$2.log 2>&1; echo "rc $?"
  tail -n 1 $O/d_kernels_child_$1$2.log | python -c "
import json,sys
l=sys.stdin.read(); d=json.loads(l[l.index('{'):])
o=d['per_step_overhead_kernels']; print({k:o[k] for k in ('pct','pct_at_profiling_interval_10')}, o['profiling_interval_1'], d.get('report_at_cadence_kernels'))" | cut -c1-1800
done
