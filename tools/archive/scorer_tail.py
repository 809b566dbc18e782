#!/usr/bin/env python3
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch, gc
import synth
from nvrx_straggler.folded import FoldedJob
S, N = 64, 10_000
for tr in (8, 1):
    job = FoldedJob(total_ranks=tr, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
    for lr in range(tr): job.load(lr, synth.stress_samples(lr, S, N))
    for _ in range(30):
        job.rearm(N); job.report()
    ws = job.reporter._ring_plan.ws
    gc.collect()
    wait, tail, tot = [], [], []
    for _ in range(300):
        job.rearm(N)
        t0 = time.perf_counter_ns(); job.report(); tot.append(time.perf_counter_ns() - t0)
        wait.append(int(ws.meta[6])); tail.append(int(ws.meta[7]) & 0xFFFF)
    print(f"total_ranks={tr} resident={os.environ.get('NVRX_RESIDENT_SCORER','1')}: report {np.median(tot)/1e3:.2f} us | scorer waited for rows {np.median(wait)/100:.2f} us, last row -> completion store {np.median(tail)/100:.2f} us (p95 {np.percentile(tail,95)/100:.2f})", flush=True)
    job.close()
