"""GPU test of per-kernel tracing (NVRX_GPU_TIMING=kernels): kernels launched inside a
profiled section show up under the reference's key format with their launch geometry, their statistics
(computed on the device) match the oracle's restatement of computeStats on the very durations the tracer
drained, RCCL-style names are kept out of the GPU score, and nothing is recorded outside sections.

The tool has to register with rocprofiler-sdk before the HIP runtime initialises, so the scenario runs in a
fresh interpreter (and this file sorts first, so it runs before the test session itself holds a HIP context).

The first HIP call of a process with this tool attached is slow where storage is cold: HIP then loads every GPU code
object of every loaded library eagerly (10.7 GB of read() calls on this image, 0.00 GB without a tool) and the
loader's access pattern pulls them in at 10-14 MB/s -- 130-165 s on such a box, 3 s from a warm page cache
(tools/debug/ktrace_stall_io.sh, ktrace_eager_load.py; rounds 1-2 knew it as "rocprofiler-sdk's start-up stall").
An opt-in sequential read-ahead exists (``NVRX_KTRACE_PREFETCH=1``: ~900 MB/s on one box, no help on a box whose
storage is slow either way); the scenario gets 150 s and is reported as XFAIL with the measured wait if a box is
slower than that."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import faulthandler, json, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(140, exit=True)   # a hang becomes a traceback, not a lost GPU box
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import numpy as np
import nvrx_straggler                      # registers the tracer: no HIP call has happened yet
from nvrx_straggler import Detector, Statistic, ktrace
import torch
from oracle import oracle

out = {}
torch.cuda.set_device(0)
x = torch.randn(1024, 1024, device="cuda")
y = torch.randn(1 << 20, device="cuda")
(x @ x).sum().item()                        # warm-up OUTSIDE any section: must not be recorded
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0")
lib = ktrace.load()
out["ready"] = int(lib.nvrx_ktrace_ready())
out["pending_before"] = int(lib.nvrx_ktrace_pending())
REPS = 24
for i in range(REPS):
    with Detector.detection_section("step", profile_cuda=True):
        z = x @ x
        w = torch.relu(y) + 1.0
    torch.sigmoid(y)                        # between sections: not traced
torch.cuda.synchronize()
# raw durations straight from the tracer (test-only peek): drain, check, and hand them back to the profiler
recs = ktrace.drain_all()
names = {int(k): ktrace.key_name(int(k)) for k in np.unique(recs["key"])}
out["names"] = sorted(names.values())
per_key = {names[int(k)]: recs["us"][recs["key"] == k] for k in np.unique(recs["key"])}
prof = Detector.cupti_manager.cupti_ext
rings = Detector.rings
for name, vals in per_key.items():
    rings.push_many(rings.row_for(1, name), vals)
report = Detector.generate_report()
checks = []
for name, vals in per_key.items():
    exp = oracle.kernel_stats(np.asarray(vals, dtype=np.float32))   # computeStats restated (f32)
    got = report.local_kernel_summaries[name]
    checks.append(dict(name=name, n=int(vals.size), exp=[float(v) for v in exp[:6]],
                       got=[float(got[s]) for s in (Statistic.MIN, Statistic.MAX, Statistic.MED, Statistic.AVG, Statistic.STD, Statistic.NUM)]))
out["checks"] = checks
out["gpu_rel"] = {str(k): float(v) for k, v in report.gpu_relative_perf_scores.items()}
out["gpu_ind"] = {str(k): float(v) for k, v in report.gpu_individual_perf_scores.items()}
out["section_num"] = int(report.local_section_summaries["step"][Statistic.NUM])
# second window: end to end through harvest()
for i in range(5):
    with Detector.detection_section("step", profile_cuda=True):
        z = x @ x
report2 = Detector.generate_report()
out["second_keys"] = sorted(report2.local_kernel_summaries.keys())
out["second_nums"] = [int(v[Statistic.NUM]) for v in report2.local_kernel_summaries.values()]
out["dropped"] = prof.dropped
from nvrx_straggler import ktrace as _kt
out["prefetch"] = dict(_kt.prefetch_stats)
Detector.shutdown()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_kernels_are_traced_by_name_and_scored():
    env = dict(os.environ)
    env.pop("NVRX_GPU_TIMING", None)
    import time as _time

    t_start = _time.monotonic()
    try:
        p = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + SCRIPT], capture_output=True, text=True, timeout=150, env=env)
    except subprocess.TimeoutExpired:
        # visible in the GPU test summary as XFAIL with the measured wait (a silent skip would hide that the per-kernel
        # mode's evidence is missing on this box); the stall is inside rocprofiler-sdk's own start-up
        pytest.xfail(f"rocprofiler-sdk start-up did not complete within {_time.monotonic() - t_start:.0f} s on this box "
                     "(inside the first HIP call, before any nvrx code ran)")
    if p.returncode != 0 and "Timeout (0:02:20)" in p.stderr and ("_lazy_init" in p.stderr or "ktrace.py" in p.stderr):
        pytest.xfail(f"rocprofiler-sdk start-up stalled for {_time.monotonic() - t_start:.0f} s inside the first HIP call")
    print(f"[ktrace] subprocess wall time {_time.monotonic() - t_start:.1f} s (SDK start-up + test body)")
    for l in p.stdout.splitlines():
        if l.startswith("RESULT "):
            print("[ktrace] read-ahead of the GPU libraries:", json.loads(l[len("RESULT "):]).get("prefetch"))
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["ready"] == 1
    assert out["pending_before"] == 0  # the warm-up matmul ran outside a section
    names = out["names"]
    # reference key format (CuptiProfiler.cpp:186-189): <name>_blk_x_y_z_grid_x_y_z
    import re

    assert names and all(re.search(r"_blk_\d+_\d+_\d+_grid_\d+_\d+_\d+$", n) for n in names), names
    assert any("Cijk" in n or "gemm" in n.lower() for n in names), names           # the matmul
    assert any("elementwise" in n for n in names), names                           # relu / add
    assert not any("sigmoid" in n for n in names), names                           # launched between sections
    assert out["section_num"] == 24
    for c in out["checks"]:
        exp, got = c["exp"], c["got"]
        assert got[5] == c["n"] == exp[5], c
        assert got[0] == exp[0] and got[1] == exp[1] and got[2] == exp[2], c       # MIN MAX MED bit-exact
        assert abs(got[3] - exp[3]) <= 2e-4 * abs(exp[3]), c                        # AVG: reference sums in f32
        assert abs(got[4] - exp[4]) <= 2e-3 * max(abs(exp[4]), 1e-3), c
    total = sum(c["n"] for c in out["checks"])
    assert total >= 3 * 24
    assert abs(out["gpu_rel"]["0"] - 1.0) < 1e-6 and abs(out["gpu_ind"]["0"] - 1.0) < 1e-6
    assert out["second_keys"] and all(n == 5 for n in out["second_nums"]), out
    assert out["dropped"] == 0
