#!/usr/bin/env python3
"""What does the torch.cuda.synchronize() that closes bench.py's timed region cost?  (20 timed steps: value - mean of the
per-step times = 1.9 us per step, i.e. ~38 us for the one call.)"""
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

S, N, R = 64, 10_000, 8
torch.cuda.synchronize()
t = []
for _ in range(20):
    t0 = time.perf_counter_ns(); torch.cuda.synchronize(); t.append(time.perf_counter_ns() - t0)
print("idle process, before any stream of ours exists: synchronize", np.median(t) / 1e3, "us")
job = FoldedJob(total_ranks=R, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N, node_name="n")
for lr, r in enumerate(job.logical_ranks()):
    job.load(lr, synth.stress_samples(r, S, N, slow_rank=3, slow_factor=1.5))
torch.cuda.synchronize()
for _ in range(30):
    job.rearm(N); job.report().identify_stragglers()
for label, fn in (("torch.cuda.synchronize()", torch.cuda.synchronize), ("backend stream synchronize", job.backend.synchronize),
                  ("torch.cuda.current_stream().synchronize()", lambda: torch.cuda.current_stream().synchronize())):
    t, t_idle = [], []
    for _ in range(20):
        for _ in range(3):
            job.rearm(N); job.report().identify_stragglers()
        t0 = time.perf_counter_ns(); fn(); t.append(time.perf_counter_ns() - t0)
        t0 = time.perf_counter_ns(); fn(); t_idle.append(time.perf_counter_ns() - t0)
    print(f"{label:44s} right after a report {np.median(t)/1e3:6.1f} us (p95 {np.percentile(t,95)/1e3:6.1f}) | again, nothing pending {np.median(t_idle)/1e3:6.1f} us", flush=True)
print("NVRX_RESIDENT_SCORER", os.environ.get("NVRX_RESIDENT_SCORER"))
job.close()
