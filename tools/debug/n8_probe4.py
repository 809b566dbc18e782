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
    dt = (time.perf_counter()-t0)/200*1e6
    # staged: enqueue only, then poll
    rep = job.reporter; plan = rep._ring_plan; ws = plan.ws; be = job.backend
    enq, wait = [], []
    for _ in range(100):
        job.rearm(N)
        t0 = time.perf_counter_ns()
        seq = job.rings.report_fused(ws, plan.rows_used, plan.stats_needed, True, True, rep.thresholds, None, names_ok=True, wait=False)
        t1 = time.perf_counter_ns()
        be.wait_seq(ws, seq)
        t2 = time.perf_counter_ns()
        enq.append(t1-t0); wait.append(t2-t1)
        job.rings.reset()
    print(tag, f"{dt:.2f} us/report | enqueue {np.median(enq)/1e3:.2f} us, enqueue->visible {np.median(wait)/1e3:.2f} us (p95 {np.percentile(wait,95)/1e3:.2f})", flush=True)
main = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
for lr in range(8): main.load(lr, synth.stress_samples(lr, S, N))
timeit(main, "main")
small = FoldedJob(total_ranks=1, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
small.load(0, synth.stress_samples(0, S, N))
timeit(small, "small with main alive")
timeit(main, "main again")
timeit(small, "small again")
timeit(main, "main again 2")
os.environ["NVRX_X"] = "1"
