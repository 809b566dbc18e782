#!/usr/bin/env python3
"""Host timeline of one warm step of the headline workload (re-arm -> generate_report -> identify_stragglers), split at
the C call: Python before `nvrx_report`, the call itself (two launches + the wait for the completion word), Python
after it (the Report object), reading the flagged set.  Medians over 2000 steps; the wrapper that takes the two extra
time stamps costs ~0.2 us, reported as `wrapper`."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
from nvrx_straggler.folded import FoldedJob  # noqa: E402

S, N, R = 64, 10_000, int(os.environ.get("TIMELINE_RANKS", "8"))
job = FoldedJob(total_ranks=R, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N, node_name="n")
for lr, r in enumerate(job.logical_ranks()):
    job.load(lr, synth.stress_samples(r, S, N, slow_rank=3 % R, slow_factor=1.5))
torch.cuda.synchronize()
now = time.perf_counter_ns
marks = [0, 0]


class _Lib:
    def __init__(self, lib):
        self._lib = lib
        self._report = lib.nvrx_report

    def nvrx_report(self, ctx, desc, stream):
        marks[0] = now()
        rc = self._report(ctx, desc, stream)
        marks[1] = now()
        return rc

    def __getattr__(self, name):
        return getattr(self._lib, name)


def run(n):
    rows = []
    for _ in range(n):
        t0 = now()
        job.rearm(N)
        t1 = now()
        rep = job.report()
        t2 = now()
        found = rep.identify_stragglers()
        t3 = now()
        rows.append((t1 - t0, marks[0] - t1, marks[1] - marks[0], t2 - marks[1], t3 - t2, t3 - t0))
    assert found["straggler_sections_relative"]
    return np.array(rows) * 1e-3


for _ in range(200):
    job.rearm(N); job.report().identify_stragglers()
plain = []
for _ in range(2000):
    t0 = now(); job.rearm(N); job.report().identify_stragglers(); plain.append(now() - t0)
real_lib, job.rings.lib = job.rings.lib, _Lib(job.rings.lib)
run(200)
m = np.median(run(2000), axis=0)
job.rings.lib = real_lib
names = ("re-arm (set_count_all)", "Python before the C call (plan key, flip, descriptor)", "nvrx_report (2 launches + wait)",
         "Python after the C call (Report object)", "identify_stragglers()", "whole step")
for k, v in zip(names, m):
    print(f"{k:58s} {v:6.2f} us")
print(f"{'whole step without the wrapper':58s} {np.median(plain) * 1e-3:6.2f} us   (wrapper: {m[5] - np.median(plain) * 1e-3:+.2f})")
job.close()
