#!/usr/bin/env python3
"""Where the per-step cost of Detector (config #4) goes: section enter/exit, harvest, report."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd")):
    sys.path.insert(0, p)
import torch
from nvrx_straggler import Detector

x = torch.randn(1024, 1024, device="cuda")
Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=True, node_name="n0")
def work():
    return x @ x
for _ in range(20):
    with Detector.detection_section("s", profile_cuda=True):
        work()
    Detector.generate_report()
N = 300
def timeit(f, n=N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def sec_only():
    with Detector.detection_section("s", profile_cuda=True):
        pass
def sec_cpu_only():
    with Detector.detection_section("c", profile_cuda=False):
        pass
def full():
    with Detector.detection_section("s", profile_cuda=True):
        work()
    Detector.generate_report()
def rep_only():
    Detector.generate_report()
print("work only            %.1f us" % timeit(work))
print("section(profile_cuda) empty body  %.1f us" % timeit(sec_only))
Detector.generate_report()
print("section(cpu only) empty body      %.1f us" % timeit(sec_cpu_only))
Detector.generate_report()
print("section+work+report  %.1f us" % timeit(full))
print("report only (no new samples)      %.1f us" % timeit(rep_only))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): full()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
Detector.shutdown()
