#!/usr/bin/env python3
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from nvrx_straggler.folded import FoldedJob
S, N = 64, 10_000
def timeit(job, tag):
    for _ in range(20):
        job.rearm(N); job.report()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        job.rearm(N); job.report()
    torch.cuda.synchronize()
    print(tag, f"{(time.perf_counter()-t0)/200*1e6:.2f} us/report", flush=True)
main = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
for lr in range(8): main.load(lr, synth.stress_samples(lr, S, N))
timeit(main, "main")
small = FoldedJob(total_ranks=1, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
small.load(0, synth.stress_samples(0, S, N))
timeit(small, "small with main alive")
timeit(main, "main again")
sweep = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")
sweep.add_(1.0); torch.cuda.synchronize()
timeit(small, "small with 1 GiB tensor alive")
del sweep
timeit(small, "small after del")
torch.cuda.empty_cache()
timeit(small, "small after empty_cache")
main.rings.timing_enable(True); main.rearm(N); main.report(); print(main.rings.timing_read()); main.rings.timing_enable(False)
timeit(small, "small after main timing")
small.rings.timing_enable(True)
timeit(small, "small with timing on")
print(small.rings.timing_read()); small.rings.timing_enable(False)
timeit(small, "small timing off")
