#!/usr/bin/env python3
"""Where does a 1-rank x 64 x 10000 FoldedJob report spend its time? (diagnostic)"""
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth
from nvrx_straggler.folded import FoldedJob

S, N = 64, 10_000
for tr in (1, 8, 1):
    job = FoldedJob(total_ranks=tr, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
    for lr in range(tr):
        job.load(lr, synth.stress_samples(lr, S, N))
    for _ in range(30):
        job.rearm(N)
        job.report()
    t = []
    for _ in range(200):
        job.rearm(N)
        t0 = time.perf_counter_ns()
        job.report()
        t.append(time.perf_counter_ns() - t0)
    print(f"total_ranks={tr}: report median {np.median(t)/1e3:.2f} us  p95 {np.percentile(t,95)/1e3:.2f}  plan={job.reporter._ring_plan is not None}")
    if tr == 1:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(100):
            job.rearm(N)
            job.report()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(8)
    job.close()
