#!/bin/bash
# same-box A/B of the cold report path: round-4 head (oracle/_ref/r04tree) against this tree, alternating.
# The round-4 tree is not kept: build it where git is available, then run this through gpurun --
#   mkdir /tmp/r04tree && git archive 28aa94f | tar -x -C /tmp/r04tree && make -C /tmp/r04tree/nvidia-resiliency-ext_amd/csrc
#   mkdir -p oracle/_ref/r04tree/profiles && cp -r /tmp/r04tree/{bench.py,nvidia-resiliency-ext_amd,tests} oracle/_ref/r04tree/ \
#     && cp /tmp/r04tree/profiles/pmc_row_stats.json oracle/_ref/r04tree/profiles/
# (oracle/_ref/ is git-ignored and travels with gpurun).  Result: profiles/r05_ab_r04_r05.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r05ab; mkdir -p $O
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-overhead --no-host-inputs --no-extra-legs --cadence-reports 30"
for i in 1 2 3; do
  (cd oracle/_ref/r04tree && timeout 300 python bench.py $ARGS 2>/dev/null | tail -n 1 > $O/r04_$i.json)
  timeout 300 python bench.py $ARGS 2>/dev/null | tail -n 1 > $O/r05_$i.json
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "r05ab")
for tag in ("r04", "r05"):
    for f in sorted(glob.glob(f"{O}/{tag}_*.json")):
        d = json.load(open(f)); c = d["report_at_cadence"]
        print(tag, d["value"], d["us_per_report_median"], "| headline", c["headline_workload"]["us_median"], c["headline_workload"]["us_p95"],
              "| sync", c["synchronous"]["us_median"], c["synchronous"]["us_p95"], "| async", c["asynchronous"]["us_median"], c["asynchronous"]["us_p95"])
PY
