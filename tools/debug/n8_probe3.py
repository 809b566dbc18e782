#!/usr/bin/env python3
import os, sys, time, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from nvrx_straggler.folded import FoldedJob
S, N = 64, 10_000
def timeit(job, tag, prof=False):
    for _ in range(20):
        job.rearm(N); job.report()
    torch.cuda.synchronize()
    pr = cProfile.Profile() if prof else None
    if pr: pr.enable()
    t0 = time.perf_counter()
    for _ in range(200):
        job.rearm(N); job.report()
    torch.cuda.synchronize()
    dt = (time.perf_counter()-t0)/200*1e6
    if pr:
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(6)
    print(tag, f"{dt:.2f} us/report  plan={job.reporter._ring_plan is not None}", flush=True)
main = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
for lr in range(8): main.load(lr, synth.stress_samples(lr, S, N))
timeit(main, "main")
small = FoldedJob(total_ranks=1, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
timeit(main, "main after creating small (never used)")
small.load(0, synth.stress_samples(0, S, N))
timeit(main, "main after small.load")
timeit(small, "small")
timeit(main, "main again", prof=True)
os.environ["X"]="1"
main.rings.timing_enable(True)
timeit(main, "main again, kernel timing on")
print("k_row_stats us avg:", [x for x in [main.rings.timing_read()]])
main.rings.timing_enable(False)
