#!/usr/bin/env python3
"""Where does the one slow report of a short timed region come from?  (--steps 20 --warmup 5 has one ~175 us
step.)  Runs the headline report (8 x 64 x 10 000 folded on one GPU) after different kinds of pause and prints the
first five reports that follow each.

    python tools/outlier_probe.py
"""
import gc
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

import synth  # noqa: E402
from nvrx_straggler.folded import FoldedJob  # noqa: E402

S, N, R = 64, 10_000, int(os.environ.get("PROBE_RANKS", "8"))
job = FoldedJob(total_ranks=R, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N, node_name="n")
for lr, r in enumerate(job.logical_ranks()):
    job.load(lr, synth.stress_samples(r, S, N, slow_rank=3 % R, slow_factor=1.5))
torch.cuda.synchronize()


def step(read=True):
    t0 = time.perf_counter_ns()
    job.rearm(N)
    rep = job.report()
    t1 = time.perf_counter_ns()
    if read:
        rep.identify_stragglers()
    t2 = time.perf_counter_ns()
    return (t1 - t0) / 1e3, (t2 - t1) / 1e3


def burst(label, pause, n=5, read=True):
    pause()
    t = [step(read) for _ in range(n)]
    print(f"{label:58s} report: " + " ".join(f"{a:7.1f}" for a, _ in t) + "   read: " + " ".join(f"{b:5.1f}" for _, b in t), flush=True)


for _ in range(30):
    step()
x = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")


def matmuls():
    y = x
    for _ in range(100):
        y = torch.matmul(x, y)
    torch.cuda.synchronize()


def busy_wait():
    end = time.perf_counter() + 0.1
    while time.perf_counter() < end:
        pass


pauses = [
    ("nothing (back to back)", lambda: None),
    ("torch.cuda.synchronize()", torch.cuda.synchronize),
    ("gc.collect()", gc.collect),
    ("gc.collect() + synchronize (what bench.py does)", lambda: (gc.collect(), torch.cuda.synchronize())),
    ("sleep 1 ms", lambda: time.sleep(0.001)),
    ("sleep 10 ms", lambda: time.sleep(0.01)),
    ("sleep 100 ms", lambda: time.sleep(0.1)),
    ("sleep 1 s", lambda: time.sleep(1.0)),
    ("busy-wait 100 ms (host spinning, GPU idle)", busy_wait),
    ("100 matmuls + synchronize (GPU busy, caches evicted)", matmuls),
    ("sleep 100 ms + one tiny kernel + sync", lambda: (time.sleep(0.1), torch.zeros(1, device="cuda").add_(1), torch.cuda.synchronize())),
]
for rnd in range(3):
    print(f"--- round {rnd}", flush=True)
    for label, pause in pauses:
        burst(label, pause)
# the same without reading the report (what r02's loop timed)
burst("gc.collect() + synchronize, reports NOT read", lambda: (gc.collect(), torch.cuda.synchronize()), read=False)
burst("sleep 100 ms, reports NOT read", lambda: time.sleep(0.1), read=False)
print("NVRX_RESIDENT_SCORER", os.environ.get("NVRX_RESIDENT_SCORER"), flush=True)
job.close()
