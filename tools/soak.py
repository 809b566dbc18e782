"""Soak of the product flows on one GPU: context create / destroy cycles, matrix loads, synchronous and asynchronous
Detector reports with GPU-timed sections on two streams, reports read late or never, PyTorch allocations in between.
Every report is checked (a lone rank scores 1.0, NUM is what was pushed); a GPU memory fault aborts the process, so a clean
exit IS the result.  tools/soak.py [seconds]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests", "golden"), os.path.join(REPO, "tests")]
import numpy as np
import torch
import synth
from nvrx_straggler import Detector, Statistic
from nvrx_straggler.folded import FoldedJob

from nvrx_straggler import ktrace

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
per_kernel = ktrace.timing_mode() == "kernels"
import faulthandler
faulthandler.dump_traceback_later(budget + 45.0, exit=True)  # a flow that never comes back says where it is
torch.cuda.set_device(0)
rng = np.random.default_rng(int(time.time()) % 1000)
t_end = time.time() + budget
counts = {"folded_jobs": 0, "folded_reports": 0, "detector_cycles": 0, "detector_reports": 0}
side = torch.cuda.Stream()
x = torch.randn(512, 512, device="cuda")
junk = []
while time.time() < t_end:
    # --- a folded job: created, loaded by matrices, reported a few times, destroyed ---
    S = int(rng.choice([4, 16, 64])); n = int(rng.choice([100, 1000, 10000])); cap = int(rng.choice([n, 8192, 10000]))
    job = FoldedJob(total_ranks=8, sections=S, ring_cap=cap)
    host = [synth.stress_samples(lr, S, n, slow_rank=3, slow_factor=1.5) for lr in job.logical_ranks()]
    for rep in range(int(rng.integers(1, 5))):
        job.rings.reset()
        for lr in job.logical_ranks():
            job.load(lr, host[lr])
        r = job.report()
        found = r.identify_stragglers()
        assert sorted(s.rank for s in found["straggler_sections_relative"]["section_000"]) == [3], found
        if rng.random() < 0.7:
            assert abs(r.section_relative_perf_scores["section_000"][3] - 1 / 1.5) < 0.02
            assert r.local_section_summaries["section_000"][Statistic.NUM] == min(n, cap)
        counts["folded_reports"] += 1
        junk.append(torch.empty(int(rng.integers(1, 1 << 22)), device="cuda"))
        if len(junk) > 6:
            del junk[: int(rng.integers(1, 5))]
    if rng.random() < 0.5:
        del r                      # sometimes the last report outlives its job
    job.close()
    counts["folded_jobs"] += 1
    # --- a Detector: sections on two streams, synchronous or asynchronous, reports read late or never ---
    asynchronous = bool(rng.random() < 0.5)
    Detector.initialize(scores_to_compute="all", gather_on_rank0=bool(rng.random() < 0.5), node_name="soak", asynchronous=asynchronous)
    held = []
    for rep in range(int(rng.integers(2, 8))):
        steps = int(rng.integers(1, 12))
        for _ in range(steps):
            with Detector.detection_section("fwd", profile_cuda=True):
                y = x @ x
            if rng.random() < 0.5:
                with torch.cuda.stream(side):
                    with Detector.detection_section("side", profile_cuda=True):
                        z = x + 1
            with Detector.detection_section("cpu", profile_cuda=False):
                pass
        report = Detector.generate_report()
        held.append((report, steps))
        if rng.random() < 0.6:
            rep_, steps_ = held.pop(int(rng.integers(0, len(held))))
            assert rep_.local_section_summaries["fwd"][Statistic.NUM] == steps_
            g = rep_.gpu_relative_perf_scores[0]
            if per_kernel and asynchronous and g != g:
                # per-kernel timing, asynchronous: a report does not wait for its window's kernels -- a window whose kernels had not
                # completed when it was enqueued has no GPU samples (they count in the next one) and scores NaN, never a wrong number
                counts["async_windows_without_kernel_samples"] = counts.get("async_windows_without_kernel_samples", 0) + 1
            else:
                assert abs(g - 1.0) < 1e-5, (g, asynchronous, steps_, rep_.local_kernel_summaries)
            assert rep_.identify_stragglers()["straggler_gpus_relative"] == set()
        if len(held) > 3:
            held.pop(0)            # never read
        counts["detector_reports"] += 1
    Detector.shutdown()
    counts["detector_cycles"] += 1
torch.cuda.synchronize()
print("soak ok", counts, flush=True)
