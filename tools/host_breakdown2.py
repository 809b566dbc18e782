#!/usr/bin/env python3
"""Raw C-call pipeline vs the full Python report path (N=1 folded shape)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import synth
from nvrx_straggler.folded import FoldedJob

S, N = 64, 10_000
job = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
for lr in range(8):
    job.load(lr, synth.stress_samples(lr, S, N))
for _ in range(50):
    job.rearm(N); job.report()
be, rings, lib = job.backend, job.rings, job.backend.lib
ws = be.workspace(8, 0, S, 8, 8 * S)
st = be.stream_handle
raw, full, parts = [], [], {"local": [], "score": [], "poll": []}
for _ in range(500):
    job.rearm(N)
    t0 = time.perf_counter_ns()
    lib.nvrx_report_local(rings.ctx, ws.d_stats, ws.send_ptr, 0, S, 1, S, st)
    t1 = time.perf_counter_ns()
    ws.seq += 1
    lib.nvrx_score(ws.send_ptr, 8, 0, S, 1, 1, be._thr, ws.d_scores, ws.d_flags, ws.d_meta, ws.d_counter, ws.seq, ws.d_stats, ws.h_stats_dst, 512, st)
    t2 = time.perf_counter_ns()
    lib.nvrx_poll_u32(ws.h_seq, ws.seq, 5.0)
    t3 = time.perf_counter_ns()
    raw.append(t3 - t0); parts["local"].append(t1 - t0); parts["score"].append(t2 - t1); parts["poll"].append(t3 - t2)
    rings.reset()
for _ in range(500):
    job.rearm(N)
    t0 = time.perf_counter_ns(); job.report(); full.append(time.perf_counter_ns() - t0)
print("raw C pipeline   median %.2f us  p95 %.2f" % (np.median(raw) / 1e3, np.percentile(raw, 95) / 1e3))
for k, v in parts.items():
    print("   %-6s median %.2f us" % (k, np.median(v) / 1e3))
print("job.report()     median %.2f us  p95 %.2f" % (np.median(full) / 1e3, np.percentile(full, 95) / 1e3))
