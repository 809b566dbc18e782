#!/usr/bin/env python3
"""Host-side budget of one planned report on the GPU box: Python around the C call, the C call, the device part."""
import os, sys, time, gc
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from nvrx_straggler.folded import FoldedJob
S, N = 64, 10_000
job = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N)
for lr in range(8): job.load(lr, synth.stress_samples(lr, S, N))
for _ in range(30):
    job.rearm(N); job.report()
ws = job.reporter._ring_plan.ws
inner = []
real = job.rings.report_fused
def timed(*a, **k):
    t0 = time.perf_counter_ns(); r = real(*a, **k); inner.append(time.perf_counter_ns() - t0); return r
job.rings.report_fused = timed
gc.collect()
tot, rearm, wait, tail = [], [], [], []
for _ in range(400):
    t0 = time.perf_counter_ns(); job.rearm(N); t1 = time.perf_counter_ns(); job.report(); t2 = time.perf_counter_ns()
    rearm.append(t1 - t0); tot.append(t2 - t1); wait.append(int(ws.meta[6])); tail.append(int(ws.meta[7]) & 0xFFFF)
f = job.rings.lib.nvrx_abi_version
empty = []
for _ in range(2000):
    t0 = time.perf_counter_ns(); f(); empty.append(time.perf_counter_ns() - t0)
m = lambda v: np.median(v) / 1e3
print(f"rearm {m(rearm):.2f} | report() {m(tot):.2f} = report_fused (ctypes + C + device) {m(inner):.2f} + python around it {m(tot) - m(inner):.2f} | "
      f"empty ctypes call {m(empty):.2f} | device: scorer waited {np.median(wait)/100:.2f}, last row -> store {np.median(tail)/100:.2f} us")
