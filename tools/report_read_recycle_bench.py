"""Host cost of a report's three big mappings with and without recycling (no GPU needed), and -- with --gpu -- of all six
mappings of real FoldedJob reports when the previous report was dropped / is still held."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests", "golden"), os.path.join(REPO, "tests")]
from nvrx_straggler import reporting
from nvrx_straggler.statistics import STAT_KEYS
pr = reporting._pyread
R, S = 8, 64; W = 2 + 2 * S
rng = np.random.default_rng(1)
sc = rng.uniform(0.5, 1, (R, W)).astype(np.float32); st = rng.uniform(1, 2, (S, 8)).astype(np.float32); st[:, 5] = 100
names = tuple(f"section_{i:03d}" for i in range(S)); ranks = tuple(range(R)); rows = tuple(range(S)); tmpl = dict.fromkeys(names)

def micro(recycle):
    ks, km = ([None, None], [None]) if recycle else (None, None)
    ts = []
    for _ in range(3000):
        t0 = time.perf_counter_ns()
        ab = pr.sections(names, ranks, sc, 0, R, W, 2, None, 2 + S, tmpl, ks); m = pr.summaries(names, STAT_KEYS, st, rows, tmpl, km)
        ts.append(time.perf_counter_ns() - t0)
        del ab, m
    return np.median(ts) / 1e3

for r in (False, True, False, True):
    print("pyread only: recycle", r, "%.2f us" % micro(r), flush=True)

if "--gpu" in sys.argv:
    import torch, synth
    from nvrx_straggler.folded import FoldedJob
    torch.cuda.set_device(0)
    job = FoldedJob(total_ranks=8, sections=64, ring_cap=10000)
    for lr in job.logical_ranks():
        job.load(lr, synth.stress_samples(lr, 64, 10000, slow_rank=3, slow_factor=1.5))
    FIELDS = ("gpu_relative_perf_scores", "section_relative_perf_scores", "gpu_individual_perf_scores", "section_individual_perf_scores",
              "local_section_summaries", "local_kernel_summaries")
    for hold in (False, True, False, True):
        ts, ids, held = [], set(), None
        for _ in range(60):
            job.rearm(10000)
            r = job.report()
            r.identify_stragglers()
            t0 = time.perf_counter_ns()
            for f in FIELDS:
                getattr(r, f)
            ts.append(time.perf_counter_ns() - t0)
            ids.add(id(r.section_relative_perf_scores))
            if hold:
                held = r
        print("six mappings, previous report", "HELD" if hold else "dropped", "%.2f us median," % (np.median(ts) / 1e3), len(ids), "distinct outer dicts in 60 reports", flush=True)
        del held, r
    job.close()
