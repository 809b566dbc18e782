"""Per-field host cost of reading a FoldedJob report's mappings (previous report dropped), and of the pieces underneath."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests", "golden"), os.path.join(REPO, "tests")]
import torch, synth
from nvrx_straggler.folded import FoldedJob
torch.cuda.set_device(0)
job = FoldedJob(total_ranks=8, sections=64, ring_cap=10000)
for lr in job.logical_ranks():
    job.load(lr, synth.stress_samples(lr, 64, 10000, slow_rank=3, slow_factor=1.5))
FIELDS = ("gpu_relative_perf_scores", "section_relative_perf_scores", "gpu_individual_perf_scores", "section_individual_perf_scores",
          "local_section_summaries", "local_kernel_summaries")
per = {f: [] for f in FIELDS}
pieces = {"raw_scores": [], "statistics": []}
for it in range(80):
    job.rearm(10000)
    r = job.report()
    r.identify_stragglers()
    if it % 2:
        src = r.__dict__["_src"]
        t0 = time.perf_counter_ns(); src.raw_scores(); t1 = time.perf_counter_ns(); src.statistics(); t2 = time.perf_counter_ns()
        pieces["raw_scores"].append(t1 - t0); pieces["statistics"].append(t2 - t1)
        continue
    for f in FIELDS:
        t0 = time.perf_counter_ns()
        getattr(r, f)
        per[f].append(time.perf_counter_ns() - t0)
for f in FIELDS:
    print("%-34s %.2f us" % (f, np.median(per[f]) / 1e3))
print("sum %.2f us" % sum(np.median(per[f]) / 1e3 for f in FIELDS))
for k, v in pieces.items():
    print("first %-12s %.2f us" % (k, np.median(v) / 1e3))
job.close()
