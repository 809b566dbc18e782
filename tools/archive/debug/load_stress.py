"""(debug) many matrix loads + reports in one process: does the strided ring append ever fault or corrupt?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests", "golden"), os.path.join(REPO, "tests")]
import numpy as np, torch
import synth
from nvrx_straggler.folded import FoldedJob
torch.cuda.set_device(0)
job = FoldedJob(total_ranks=8, sections=64, ring_cap=10000)
host = [synth.stress_samples(lr, 64, 10000, slow_rank=3, slow_factor=1.5) for lr in job.logical_ranks()]
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    job.rings.reset()
    for lr in job.logical_ranks():
        job.load(lr, host[lr])
    r = job.report()
    got = sorted(x.rank for x in r.identify_stragglers()["straggler_gpus_relative"]), r.section_relative_perf_scores["section_005"][3]
    if ref is None:
        ref = got
    assert got == ref, (it, got, ref)
print("load stress ok", ref, flush=True)
job.close()
