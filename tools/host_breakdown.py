#!/usr/bin/env python3
"""Where does a report's wall time go?  Times each host-visible stage of the folded N=1 report
(stage boundaries are made synchronous here, so the sum exceeds the pipelined end-to-end time)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth
from nvrx_straggler import _native
from nvrx_straggler.folded import FoldedJob

S, N = 64, 10_000
job = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
for lr in range(8):
    job.load(lr, synth.stress_samples(lr, S, N))
for _ in range(30):
    job.rearm(N)
    job.report()
be, rings, lib = job.backend, job.rings, job.backend.lib
ws = be.workspace(8, 0, S, 8, 8 * S)
st = be.stream_handle
T = {k: [] for k in ("rearm", "report_local_call", "report_local_sync", "score_call", "score_sync", "d2h_sync", "python_report", "full")}
for _ in range(300):
    t = time.perf_counter_ns(); job.rearm(N); T["rearm"].append(time.perf_counter_ns() - t)
    t = time.perf_counter_ns(); rings.report_local(ws, True, rows_active=S); T["report_local_call"].append(time.perf_counter_ns() - t)
    t = time.perf_counter_ns(); be.stream.synchronize(); T["report_local_sync"].append(time.perf_counter_ns() - t)
    t = time.perf_counter_ns(); lib.nvrx_score(ws.send_ptr, 8, 0, S, 1, 1, be._thr, ws.d_scores, ws.d_flags, ws.d_meta, None, 0, None, None, 0, st); T["score_call"].append(time.perf_counter_ns() - t)
    t = time.perf_counter_ns(); be.stream.synchronize(); T["score_sync"].append(time.perf_counter_ns() - t)
    t = time.perf_counter_ns(); be.stream.synchronize(); T["d2h_sync"].append(time.perf_counter_ns() - t)
    rings.reset()
for _ in range(300):
    job.rearm(N)
    t = time.perf_counter_ns(); job.report(); T["full"].append(time.perf_counter_ns() - t)
import cProfile, pstats
pr = cProfile.Profile()
job.rearm(N)
pr.enable()
for _ in range(200):
    job.rearm(N); job.report()
pr.disable()
for k, v in T.items():
    if v:
        print(f"{k:22s} median {np.median(v)/1e3:8.2f} us   p95 {np.percentile(v,95)/1e3:8.2f} us")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
