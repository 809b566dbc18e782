#!/usr/bin/env python3
"""cProfile of the host side of a warm report (held + flagged set read), headline workload: where the Python time goes."""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

import synth  # noqa: E402
from nvrx_straggler.folded import FoldedJob  # noqa: E402

S, N = 64, 10_000
job = FoldedJob(total_ranks=8, section_names=[synth.section_name(s) for s in range(S)], ring_cap=N, node_name="n")
for lr, r in enumerate(job.logical_ranks()):
    job.load(lr, synth.stress_samples(r, S, N, slow_rank=3, slow_factor=1.5))
torch.cuda.synchronize()
rep = found = None
for _ in range(100):
    job.rearm(N); rep = job.report(); found = rep.identify_stragglers()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    job.rearm(N); rep = job.report(); found = rep.identify_stragglers()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
job.close()
