#!/usr/bin/env python3
"""Per-block step times of the bench's transformer workload with every entry of its section traced (per-kernel mode, budget
off) against the same loop without the detector, block by block -- to see WHICH blocks carry the tracing cost when the
paired median of bench.py's ``per_step_overhead_kernels.profiling_interval_1`` moves between 0.4 and 6.5 % from box to box.
usage (GPU box): NVRX_GPU_TIMING=kernels python tools/trace_overhead_probe.py [--rounds 12] [--steps 10] [--sync-exit]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nvidia-resiliency-ext_amd"))
os.environ.setdefault("NVRX_GPU_TIMING", "kernels")
import nvrx_straggler  # noqa: E402,F401  (registers the tracer before HIP starts)
import torch  # noqa: E402
from nvrx_straggler import Detector, ktrace  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--sync-exit", action="store_true", help="synchronise the device before the section closes (no traced dispatch in flight at stop)")
ap.add_argument("--layers", type=int, default=10)
ap.add_argument("--outer", action="store_true", help="one outer profile_cuda section around each timed block: the inner exits do not stop the tracing context")
a = ap.parse_args()

torch.cuda.set_device(0)
torch.manual_seed(0)
d_model, batch, seq = 2048, 8, 1024
blocks = torch.nn.ModuleList([torch.nn.TransformerEncoderLayer(d_model, 16, 4 * d_model, dropout=0.0, batch_first=True, norm_first=True)
                              for _ in range(a.layers)]).to("cuda", torch.bfloat16)
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


def timed(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        fn()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, host / steps * 1e3


Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", profiling_interval=1, kernel_trace_budget_pct=0.0)


body_ms, exit_ms, enter_ms = [], [], []


def with_section():
    t0 = time.perf_counter()
    cm = Detector.detection_section("train_step", profile_cuda=True)
    cm.__enter__()
    t1 = time.perf_counter()
    train_step()
    if a.sync_exit:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    cm.__exit__(None, None, None)
    t3 = time.perf_counter()
    enter_ms.append((t1 - t0) * 1e3), body_ms.append((t2 - t1) * 1e3), exit_ms.append((t3 - t2) * 1e3)


for _ in range(3):
    with_section()
Detector.generate_report()
print("mode", ktrace.timing_mode(), "env", {k: v for k, v in os.environ.items() if k.startswith("NVRX_")}, flush=True)
print("round  without_ms (host)   section_ms (host)    pct   pump_flushes  enqueued")
c_prev = ktrace.counters()
for r in range(a.rounds):
    order = [("without", train_step), ("section", with_section)]
    if r % 2:
        order.reverse()
    res = {}
    for name, fn in order:
        if name == "section" and a.outer:
            with Detector.detection_section("outer", profile_cuda=True):
                res[name] = timed(fn, a.steps)
        else:
            res[name] = timed(fn, a.steps)
        if name == "section":
            Detector.generate_report()
    c = ktrace.counters()
    w, s = res["without"], res["section"]
    print(f"{r:3d}   {w[0]:8.3f} ({w[1]:6.2f})   {s[0]:8.3f} ({s[1]:6.2f})   {100 * (s[0] - w[0]) / w[0]:6.2f}   {c['pump_flushes'] - c_prev['pump_flushes']:6d}  {c['enqueued'] - c_prev['enqueued']:7d}",
          flush=True)
    c_prev = c
import statistics
print("traced steps: enter / body / exit host ms (median)", round(statistics.median(enter_ms), 3), round(statistics.median(body_ms), 3), round(statistics.median(exit_ms), 3))
print("counters", ktrace.counters())
Detector.shutdown()
