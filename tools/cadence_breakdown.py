#!/usr/bin/env python3
"""Where a report's time goes when it runs at production cadence (one report per 100 training steps): the C call
(two launches + the wait for the completion word), the Python around it, reading the flagged set.  Compared with the
same report back to back.

    python tools/cadence_breakdown.py
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
from nvrx_straggler.folded import FoldedJob  # noqa: E402

S, N, R = 64, 10_000, int(os.environ.get("PROBE_RANKS", "8"))
job = FoldedJob(total_ranks=R, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N, node_name="n")
for lr, r in enumerate(job.logical_ranks()):
    job.load(lr, synth.stress_samples(r, S, N, slow_rank=3 % R, slow_factor=1.5))
torch.cuda.synchronize()

lib = job.rings.lib
c_time = [0]
orig = lib.nvrx_report


def timed_report(*a):
    t0 = time.perf_counter_ns()
    rc = orig(*a)
    c_time[0] = time.perf_counter_ns() - t0
    return rc


lib.nvrx_report = timed_report
x = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")


def one():
    t0 = time.perf_counter_ns()
    job.rearm(N)
    rep = job.report()
    t1 = time.perf_counter_ns()
    rep.identify_stragglers()
    t2 = time.perf_counter_ns()
    return (t1 - t0) / 1e3, c_time[0] / 1e3, (t2 - t1) / 1e3


def show(label, rows):
    a = np.array(rows)
    print(f"{label:46s} report {np.median(a[:,0]):7.1f} us (C call {np.median(a[:,1]):6.1f}, Python around it {np.median(a[:,0]-a[:,1]):6.1f})"
          f" | identify_stragglers {np.median(a[:,2]):6.1f} | p95 total {np.percentile(a[:,0]+a[:,2],95):7.1f}", flush=True)


for _ in range(30):
    one()
show("back to back", [one() for _ in range(100)])
for mode in ("100 steps of 10 matmuls + synchronize", "sleep 50 ms (host idle, GPU idle)", "100 steps, synchronize, then 1 ms of host spinning"):
    rows = []
    for i in range(22):
        if mode.startswith("100 steps"):
            for _ in range(100):
                y = x
                for _ in range(10):
                    y = torch.matmul(x, y)
            torch.cuda.synchronize()
            if "spinning" in mode:
                end = time.perf_counter() + 0.001
                while time.perf_counter() < end:
                    pass
        else:
            time.sleep(0.05)
        r = one()
        if i >= 2:
            rows.append(r)
    show(mode, rows)
job.close()
