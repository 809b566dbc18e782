#!/usr/bin/env python3
"""Config #4 loop (report every step around 10 matmuls) for a kernel trace: where does the step's extra time go?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd")):
    sys.path.insert(0, p)
import torch
from nvrx_straggler import Detector
x = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")
def work():
    y = x
    for _ in range(10):
        y = torch.matmul(x, y)
    return y
asyn = os.environ.get("ASYNC", "0") == "1"
Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=True, node_name="n", asynchronous=asyn)
prev = None
for i in range(120):
    with Detector.detection_section("train_step", profile_cuda=True):
        work()
    rep = Detector.generate_report()
    if asyn:
        if prev is not None: prev.identify_stragglers()
        prev = rep
torch.cuda.synchronize()
Detector.shutdown()
