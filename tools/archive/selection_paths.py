#!/usr/bin/env python3
"""Which selection path do the rows of the bench data take, and how long does k_row_stats run on different data?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from nvrx_straggler.folded import FoldedJob
S, N = 64, 10_000
def run(tag, gen):
    job = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
    for lr in range(8):
        job.load(lr, gen(lr))
    for _ in range(20):
        job.rearm(N); job.report()
    ws = job.reporter._ring_plan.ws
    paths = ws.stats[:, 7].copy()
    job.rings.timing_enable(True); job.rings.timing_read(reset=True)
    for _ in range(200):
        job.rearm(N); job.report()
    torch.cuda.synchronize()
    us, n = job.rings.timing_read(reset=True)
    job.rings.timing_enable(False)
    # statistics kernel alone: report_local + stream sync (no score kernel in the queue)
    job.rings.timing_enable(True)
    for _ in range(200):
        job.rearm(N); job.rings.report_local(ws, True, rows_active=S); job.backend.synchronize(); job.rings.reset()
    us2, n2 = job.rings.timing_read(reset=True)
    job.rings.timing_enable(False)
    vals, cnt = np.unique(paths, return_counts=True)
    print(f"{tag}: k_row_stats {us/n:.2f} us in reports, {us2/n2:.2f} us alone; paths {dict(zip(vals.tolist(), cnt.tolist()))}", flush=True)
    job.close()
run("bench data (stress, rank 3 x1.5)", lambda r: synth.stress_samples(r, S, N, slow_rank=3, slow_factor=1.5))
run("stress, no slow rank", lambda r: synth.stress_samples(r, S, N))
rng = np.random.default_rng(0)
run("N(10, 0.3) every row", lambda r: rng.normal(10.0, 0.3, (S, N)).astype(np.float32))
