#!/usr/bin/env python3
"""Detector.generate_report() at production cadence (config #2: 4 sections per step, 2 GPU-timed, one report per 100
steps), cold: which stage costs what.  Stages are timed by wrapping the callables the method goes through."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvrx_straggler import Detector  # noqa: E402

marks = {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter_ns()
        try:
            return fn(*a, **k)
        finally:
            marks[label] = marks.get(label, 0) + time.perf_counter_ns() - t0

    setattr(obj, name, timed)


x = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")


def work(n):
    y = x
    for _ in range(n):
        y = torch.matmul(x, y)


ASYNC = "--async" in sys.argv
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", **({"asynchronous": True} if ASYNC else {}))
if ASYNC:
    wrap(Detector.reporter, "_settle_inflight", "reporter._settle_inflight (previous report)")
wrap(Detector.cupti_manager, "harvest", "harvest")
wrap(Detector.rings, "counts", "rings.counts")
wrap(Detector.rings, "report_fused", "rings.report_fused (Python + C call)")
wrap(Detector.rings.lib, "nvrx_report", "  nvrx_report (C)")
wrap(Detector.rings, "reset", "rings.reset")
wrap(Detector.reporter, "generate_report_from_rings", "reporter.generate_report_from_rings")
import ctypes  # noqa: E402

clk = (ctypes.c_double * 8)()
rows = []
verdicts = {}
for cadence in (True, False):
    acc = []
    for i in range(24 if cadence else 60):
        n_steps = 100 if cadence else 1
        for _ in range(n_steps):
            with Detector.detection_section("data", profile_cuda=False):
                pass
            with Detector.detection_section("forward", profile_cuda=True):
                work(4 if cadence else 1)
            with Detector.detection_section("backward", profile_cuda=True):
                work(6 if cadence else 1)
            with Detector.detection_section("optimizer", profile_cuda=False):
                pass
        torch.cuda.synchronize()
        marks.clear()
        t0 = time.perf_counter_ns()
        rep = Detector.generate_report()
        t1 = time.perf_counter_ns()
        if not ASYNC:
            rep.identify_stragglers()
        t2 = time.perf_counter_ns()
        Detector.rings.lib.nvrx_report_clocks(clk)
        verdicts[Detector.rings.lib.nvrx_ctx_info(Detector.rings.ctx, 6)] = verdicts.get(Detector.rings.lib.nvrx_ctx_info(Detector.rings.ctx, 6), 0) + 1
        marks["C: entry -> stream ordering done (event record / wait pairs)"] = int((clk[1] - clk[0]) * 1e3)
        marks["C: -> staged samples flushed (k_scatter launch)"] = int((clk[2] - clk[1]) * 1e3)
        marks["C: -> k_row_stats launched"] = int((clk[3] - clk[2]) * 1e3)
        marks["C: -> score kernel launched"] = int((clk[5] - clk[3]) * 1e3)
        marks["C:    of which the launch call itself"] = int((clk[5] - clk[7]) * 1e3) if clk[7] > clk[3] else 0
        if not ASYNC:
            marks["C: -> completion word seen (poll)"] = int((clk[6] - clk[5]) * 1e3)
        if i >= 4:
            d = dict(marks)
            d["TOTAL generate_report"] = t1 - t0
            d["identify_stragglers"] = t2 - t1
            acc.append(d)
    print(("=== ASYNCHRONOUS " if ASYNC else "=== ") + ("one report per 100 training steps (cold)" if cadence else "a report every step, tiny steps (warm)"))
    for k in ("TOTAL generate_report", "reporter._settle_inflight (previous report)", "harvest", "rings.counts", "reporter.generate_report_from_rings", "rings.report_fused (Python + C call)",
              "  nvrx_report (C)", "C: entry -> stream ordering done (event record / wait pairs)",
              "C: -> staged samples flushed (k_scatter launch)", "C: -> k_row_stats launched", "C: -> score kernel launched",
              "C:    of which the launch call itself", "C: -> completion word seen (poll)", "rings.reset", "identify_stragglers"):
        v = [a.get(k, 0) for a in acc]
        print(f"  {k:44s} median {np.median(v)/1e3:7.1f} us   p95 {np.percentile(v,95)/1e3:7.1f}")
print("re-home verdicts of the reports (1 re-homed, 0 not a candidate, -1 streams, -2 side work, -3 async in flight):", verdicts)
Detector.shutdown()
