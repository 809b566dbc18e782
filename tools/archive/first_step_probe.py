#!/usr/bin/env python3
"""Why is the FIRST timed step of bench.py slow (52-95 us against a 35 us median)?  Replays bench.py's sequence --
guard reports, warm-up, torch.cuda.synchronize() twice, timed loop -- several times in one process, with the report and
its read timed apart, and with variations of what sits between the warm-up and the timed loop."""
import gc
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

import synth  # noqa: E402
from nvrx_straggler.folded import FoldedJob  # noqa: E402

S, N, R = 64, 10_000, 8
job = FoldedJob(total_ranks=R, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N, node_name="n")
for lr, r in enumerate(job.logical_ranks()):
    job.load(lr, synth.stress_samples(r, S, N, slow_rank=3, slow_factor=1.5))
torch.cuda.synchronize()


def step():
    job.rearm(N)
    rep = job.report()
    return rep, (rep.identify_stragglers() if rep is not None else None)


def timed(n, label):
    out = []
    for _ in range(n):
        t0 = time.perf_counter_ns()
        job.rearm(N)
        rep = job.report()
        t1 = time.perf_counter_ns()
        found = rep.identify_stragglers()
        t2 = time.perf_counter_ns()
        out.append(((t1 - t0) / 1e3, (t2 - t1) / 1e3))
    print(f"{label:58s} " + " ".join(f"{a:5.1f}+{b:4.1f}" for a, b in out[:6]), flush=True)


gc.collect()
gc.freeze()
rep = found = None
for _ in range(30):
    rep, found = step()
for rnd in range(2):
    torch.cuda.synchronize(); torch.cuda.synchronize()
    timed(8, "after synchronize x2 (bench.py)")
    timed(8, "straight on (no pause)")
    torch.cuda.synchronize()
    timed(8, "after synchronize x1")
    for _ in range(5):
        rep, found = step()
    timed(8, "after 5 step() calls holding rep/found (warm-up style)")
    for _ in range(5):
        rep, found = step()
    torch.cuda.synchronize(); torch.cuda.synchronize()
    timed(8, "warm-up style steps, then synchronize x2")
    for _ in range(5):
        rep, found = step()
    rep = found = None
    torch.cuda.synchronize(); torch.cuda.synchronize()
    timed(8, "warm-up style steps, references dropped, synchronize x2")
    job.backend.synchronize()
    timed(8, "after the detector stream's own synchronize")
job.close()
