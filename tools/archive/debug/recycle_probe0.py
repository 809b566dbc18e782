"""(debug) the first version of the probe, as it was when it faulted once"""
import os, sys, gc
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests", "golden"), os.path.join(REPO, "tests")]
import torch
import synth
from nvrx_straggler.folded import FoldedJob
torch.cuda.set_device(0)
job = FoldedJob(total_ranks=8, sections=64, ring_cap=10000)
for lr in job.logical_ranks():
    job.load(lr, synth.stress_samples(lr, 64, 10000, slow_rank=3, slow_factor=1.5))
ids = []
for i in range(4):
    job.rearm(10000)
    r = job.report()
    r.identify_stragglers()
    m = r.section_relative_perf_scores
    s = r.local_section_summaries
    v = r._source.view if hasattr(r, "_source") else None
    print(i, "outer id", id(m), "refcount", sys.getrefcount(m), "summaries id", id(s), sys.getrefcount(s), "inner refcount",
          sys.getrefcount(next(iter(m.values()))), file=sys.stderr)
    if i == 3:
        print("referrers of the mapping:", [type(x).__name__ for x in gc.get_referrers(m)][:8], file=sys.stderr)
    del m, s
print("before close", file=sys.stderr)
job.close()
print("after close", file=sys.stderr)
