#!/usr/bin/env python3
"""What a GPU-timed section entry costs in per-kernel mode when NOTHING is launched inside it (the reference's
tests/straggler/unit/test_sections.py: time.sleep sections, profile_cuda left at its default True), and what the two SDK
calls behind it cost on their own.  NVRX_GPU_TIMING=kernels python tools/probe_ktrace_sections.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("NVRX_GPU_TIMING", "kernels")
import nvrx_straggler  # noqa: E402,F401
from nvrx_straggler import Detector, ktrace  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
lib = ktrace.load()
out = {"mode": ktrace.timing_mode(), "ready": int(lib.nvrx_ktrace_ready())}
if out["ready"]:
    t = []
    for _ in range(300):
        t0 = time.perf_counter_ns()
        lib.nvrx_ktrace_start()
        t1 = time.perf_counter_ns()
        lib.nvrx_ktrace_stop()
        t2 = time.perf_counter_ns()
        t.append((t1 - t0, t2 - t1))
    a = np.asarray(t[20:], dtype=np.float64) / 1e3
    out["start_context_us_median_p95"] = [round(float(np.median(a[:, 0])), 2), round(float(np.percentile(a[:, 0], 95)), 2)]
    out["stop_context_us_median_p95"] = [round(float(np.median(a[:, 1])), 2), round(float(np.percentile(a[:, 1], 95)), 2)]
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n")
x = torch.randn(256, 256, device="cuda")
for label, body in (("empty", lambda: None), ("one_small_kernel", lambda: x.add_(1.0)), ("sleep_1ms", lambda: time.sleep(0.001))):
    t = []
    for i in range(200):
        t0 = time.perf_counter_ns()
        with Detector.detection_section("s_" + label):
            body()
        t.append(time.perf_counter_ns() - t0)
    out["section_entry_us_" + label] = round(float(np.median(t[20:])) / 1e3, 2)
    t0 = time.perf_counter()
    Detector.generate_report()
    out["report_ms_" + label] = round((time.perf_counter() - t0) * 1e3, 3)
out["counters"] = ktrace.counters()
Detector.shutdown()
print(out)
