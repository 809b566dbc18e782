#!/usr/bin/env python3
"""Detector.generate_report() in PER-KERNEL mode at cadence (bench.py's ``report_at_cadence_kernels`` loop: a transformer step
with > 500 traced dispatches inside one GPU-timed section, 40 steps between reports), taken apart: which Python stage and
which part of the C call costs what.  Stages are timed by wrapping the callables the method goes through; a second pass runs
a few reports under cProfile and prints the functions by own time.

    python tools/cadence_kernels_breakdown.py [--async] [--reports N] [--profile] [--no-lane]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["NVRX_GPU_TIMING"] = os.environ.get("NVRX_GPU_TIMING", "kernels")
import nvrx_straggler  # noqa: E402,F401  (registers the tracer: nothing has touched HIP yet)
import ctypes  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvrx_straggler import Detector, ktrace, reporting, straggler  # noqa: E402

if "--no-lane" in sys.argv:
    Detector._lanes_enabled = False  # the general path at every report (what rounds 1-5 ran)

ASYNC = "--async" in sys.argv
REPORTS = int(sys.argv[sys.argv.index("--reports") + 1]) if "--reports" in sys.argv else 8
marks, calls = {}, {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter_ns()
        try:
            return fn(*a, **k)
        finally:
            marks[label] = marks.get(label, 0) + time.perf_counter_ns() - t0
            calls[label] = calls.get(label, 0) + 1

    setattr(obj, name, timed)


torch.cuda.set_device(0)
torch.manual_seed(0)
d_model, layers, batch, seq = 2048, 10, 8, 1024
blocks = torch.nn.ModuleList([torch.nn.TransformerEncoderLayer(d_model, 16, 4 * d_model, dropout=0.0, batch_first=True,
                                                               norm_first=True) for _ in range(layers)]).to("cuda", torch.bfloat16)
opt = torch.optim.SGD(blocks.parameters(), lr=1e-6, foreach=True)
x = torch.randn(batch, seq, d_model, device="cuda", dtype=torch.bfloat16)


def train_step():
    h = x
    for blk in blocks:
        h = blk(h)
    h.float().square().mean().backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    train_step()
torch.cuda.synchronize()

Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", asynchronous=ASYNC)
mgr, rings, rep_gen = Detector.cupti_manager, Detector.rings, Detector.reporter
ext = mgr.cupti_ext
LABELS = []
for obj, name, label in (
        (mgr, "harvest", "manager.harvest (lock + profiler.harvest)"),
        (ext, "harvest", "  profiler.harvest (nvrx_ktrace_sync + key learning)"),
        (ext, "_learn_keys", "    profiler._learn_keys"),
        (rings, "counts", "rings.counts"),
        (rep_gen, "generate_report_from_rings", "reporter.generate_report_from_rings"),
        (rep_gen, "_settle_inflight", "  reporter._settle_inflight (previous asynchronous report)"),
        (rep_gen, "_report_from_plan", "  reporter._report_from_plan (steady state)"),
        (rep_gen, "_score_round", "  reporter._score_round (general path)"),
        (rep_gen, "_assemble", "  reporter._assemble (general path)"),
        (rep_gen, "_build_ring_plan", "  reporter._build_ring_plan (cold)"),
        (rings, "report_fused", "    rings.report_fused (Python + C call)"),
        (rings.lib, "nvrx_report", "      nvrx_report (C)"),
        (rings, "reset", "rings.reset"),
        (ext, "hold", "profiler.hold (asynchronous bracket)"),
):
    if hasattr(obj, name):
        wrap(obj, name, label)
        LABELS.append(label)

wrap(straggler._Lane, "run", "lane.run (steady state: one function around nvrx_window_report)")
LABELS.insert(0, "lane.run (steady state: one function around nvrx_window_report)")
wrap(rings.lib, "nvrx_window_report", "  nvrx_window_report (C: wait for the window + occupancy look + nvrx_report + reset)")
LABELS.insert(1, "  nvrx_window_report (C: wait for the window + occupancy look + nvrx_report + reset)")
clk = (ctypes.c_double * 8)()
wclk = (ctypes.c_double * 2)()
C_LABELS = ["C: window entry -> nvrx_report entry (wait for the window's records, occupancy look)", "C: entry -> stream ordering done", "C: -> staged samples flushed (k_scatter launch)", "C: -> k_row_stats launched",
            "C: -> score kernel launched", "C: -> completion word seen (poll)"]


def one_report(record):
    for _ in range(40):
        with Detector.detection_section("train_step", profile_cuda=True):
            train_step()
    torch.cuda.synchronize()
    marks.clear()
    calls.clear()
    t0 = time.perf_counter_ns()
    rep = Detector.generate_report()
    t1 = time.perf_counter_ns()
    if not ASYNC:
        rep.identify_stragglers()
    t2 = time.perf_counter_ns()
    rings.lib.nvrx_report_clocks(clk)
    rings.lib.nvrx_window_clocks(wclk)
    if record is not None:
        d = dict(marks)
        d["C: window entry -> nvrx_report entry (wait for the window's records, occupancy look)"] = int((wclk[1] - wclk[0]) * 1e3) if wclk[1] <= clk[0] + 1 and clk[0] - wclk[1] < 50 else 0
        d["TOTAL generate_report"] = t1 - t0
        d["identify_stragglers"] = t2 - t1
        d[C_LABELS[1]] = int((clk[1] - clk[0]) * 1e3)
        d[C_LABELS[2]] = int((clk[2] - clk[1]) * 1e3)
        d[C_LABELS[3]] = int((clk[3] - clk[2]) * 1e3)
        d[C_LABELS[4]] = int((clk[5] - clk[3]) * 1e3)
        d[C_LABELS[5]] = 0 if ASYNC else int((clk[6] - clk[5]) * 1e3)
        d["_calls"] = dict(calls)
        record.append(d)
    return rep


acc = []
lane_marks = []
for i in range(REPORTS + 3):
    straggler._Lane.trace = [] if i >= 3 else None
    one_report(acc if i >= 3 else None)
    if straggler._Lane.trace:
        lane_marks.append(dict(straggler._Lane.trace))
straggler._Lane.trace = None
print(("=== ASYNCHRONOUS, " if ASYNC else "=== synchronous, ") + f"per-kernel mode ({ktrace.timing_mode()}), one report per 40 transformer steps, "
      f"{len(rings.kernel_row_names)} kernel keys, rows_used {rings.rows_used}, {len(acc)} reports")
for k in ["TOTAL generate_report"] + LABELS + C_LABELS + ["identify_stragglers"]:
    v = [a.get(k, 0) for a in acc]
    n = [a["_calls"].get(k, 0) for a in acc]
    print(f"  {k:62s} median {np.median(v) / 1e3:8.1f} us   p95 {np.percentile(v, 95) / 1e3:8.1f}   calls/report {np.median(n):.0f}")
if lane_marks:
    keys = [k for k in ("checks", "settle_flip_desc", "c_call", "live_block", "report_object") if k in lane_marks[0]]
    print("  inside lane.run, time from its first line (cumulative, median): " +
          ", ".join(f"{k} {np.median([m.get(k, 0) for m in lane_marks]) / 1e3:.1f} us" for k in keys))
print("tracer counters:", ktrace.counters())

if "--profile" in sys.argv:
    import cProfile
    import pstats

    pr = cProfile.Profile()
    for i in range(4):
        for _ in range(40):
            with Detector.detection_section("train_step", profile_cuda=True):
                train_step()
        torch.cuda.synchronize()
        pr.enable()
        rep = Detector.generate_report()
        if not ASYNC:
            rep.identify_stragglers()
        pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
Detector.shutdown()
