#!/usr/bin/env python3
"""What does the ATTACHED kernel tracer cost the job outside profiled sections?  rocprofiler-sdk intercepts the HSA queues
of a process as soon as a tool with the dispatch-tracing service is registered, whether or not its context is running.
Same process image twice (NVRX_GPU_TIMING=stamp / kernels): host time per launch of a launch-bound loop of tiny kernels,
GPU step time of the GEMM loop, and both again INSIDE a GPU-timed section."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd")):
    sys.path.insert(0, p)
import nvrx_straggler  # noqa: E402,F401  (registers the tracer when the mode is kernels)
from nvrx_straggler import Detector, ktrace  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.set_device(0)
small = torch.zeros(64, device="cuda")
x = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")


def tiny(n=2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        small.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def gemms(steps=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = x
        for _ in range(10):
            y = torch.matmul(x, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=False, node_name="n")
for _ in range(3):
    tiny(200), gemms(5)
out = {"mode": ktrace.timing_mode()}
a = [tiny() for _ in range(5)]
out["tiny kernel: host us per launch / us per launch incl. drain"] = (round(float(np.median([v[0] for v in a])), 2), round(float(np.median([v[1] for v in a])), 2))
out["GEMM step us (10 x matmul 4096^2)"] = round(float(np.median([gemms() for _ in range(5)])), 1)
with Detector.detection_section("s", profile_cuda=True):
    b = [tiny() for _ in range(5)]
    g = float(np.median([gemms() for _ in range(5)]))
out["inside a GPU-timed section: tiny kernel"] = (round(float(np.median([v[0] for v in b])), 2), round(float(np.median([v[1] for v in b])), 2))
out["inside a GPU-timed section: GEMM step us"] = round(g, 1)
t0 = time.perf_counter()
Detector.generate_report()
out["report after 10 000 traced launches, ms"] = round((time.perf_counter() - t0) * 1e3, 2)
Detector.shutdown()
print(out)
