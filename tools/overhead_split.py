#!/usr/bin/env python3
"""Where the per-step overhead of config #4 goes: the same A/B blocks as bench.py's per_step_overhead leg with the pieces
added one at a time -- (a) the bare loop, (b) + GPU-timed section around the step (two stamp kernels), (c) + a report every
step that nobody reads until the next step (asynchronous), (d) + a report every step read at once (synchronous), and the
report every 10th step for both."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvrx_straggler import Detector  # noqa: E402

x = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")
STEPS, BLOCKS = 100, 5


def work():
    y = x
    for _ in range(10):
        y = torch.matmul(x, y)
    return y


def run(asynchronous):
    Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=True, node_name="n", asynchronous=asynchronous)
    held = [None]

    def bare():
        work()

    def section_only():
        with Detector.detection_section("train_step", profile_cuda=True):
            work()

    def report_every(k):
        n = [0]

        def step():
            with Detector.detection_section("train_step", profile_cuda=True):
                work()
            n[0] += 1
            if n[0] % k:
                return
            rep = Detector.generate_report()
            if asynchronous:
                prev, held[0] = held[0], rep
                if prev is not None:
                    prev.identify_stragglers()
            else:
                rep.identify_stragglers()
        return step

    variants = [("bare loop", bare), ("+ GPU-timed section", section_only), ("+ report every 10th step", report_every(10)),
                ("+ report every step", report_every(1))]
    try:
        for _, fn in variants:
            for _ in range(10):
                fn()
        times = {name: [] for name, _ in variants}
        for _ in range(BLOCKS):
            for name, fn in variants:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(STEPS):
                    fn()
                torch.cuda.synchronize()
                times[name].append((time.perf_counter() - t0) / STEPS)
            Detector.generate_report()
    finally:
        Detector.shutdown()
    base = float(np.median(times["bare loop"]))
    print("asynchronous" if asynchronous else "synchronous", "reports; step = 10 x matmul(4096^2, bf16): %.1f us" % (base * 1e6))
    for name, _ in variants[1:]:
        t = float(np.median(times[name]))
        print("  %-28s +%6.1f us per step  (%.2f %%)" % (name, (t - base) * 1e6, (t - base) / base * 100))


for a in (False, True):
    run(a)
