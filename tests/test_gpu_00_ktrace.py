"""GPU test of per-kernel tracing (NVRX_GPU_TIMING=kernels): kernels launched inside a
profiled section show up under the reference's key format with their launch geometry, their statistics
(computed on the device) match the oracle's restatement of computeStats on the very durations the tracer
drained, RCCL-style names are kept out of the GPU score, and nothing is recorded outside sections.

The tool has to register with rocprofiler-sdk before the HIP runtime initialises, so the scenario runs in a
fresh interpreter (and this file sorts first, so it runs before the test session itself holds a HIP context).

Start-up: rocprofiler-sdk looks for tools by reading EVERY loaded library front to back (10.7 GB of read() calls in a
PyTorch process: the 130-165 s "start-up stall" of rounds 1-3 on boxes with cold storage).  ``ktrace.setup`` hands the tool
over explicitly with the large libraries hidden from that one search (``nvrx_ktrace.cpp``, "tool discovery guard"), so
the scenarios below have no slow-box branch any more: registration has to finish in seconds and the test asserts it."""
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
import time
import numpy as np
import torch                               # (libraries resident, HIP not started)
def _rchar():
    return int(dict(l.split(": ") for l in open("/proc/self/io").read().strip().splitlines())["rchar"])
_t0, _r0 = time.monotonic(), _rchar()
import nvrx_straggler                      # registers the tracer: no HIP call has happened yet
from nvrx_straggler import Detector, Statistic, ktrace
from oracle import oracle

out = {"register_s": time.monotonic() - _t0, "register_read_gb": (_rchar() - _r0) / 1e9,
       "hidden_libraries": int(ktrace.load().nvrx_ktrace_hidden_libraries())}
_t0, _r0 = time.monotonic(), _rchar()
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
out["first_hip_call_s"] = time.monotonic() - _t0
out["first_hip_call_read_gb"] = (_rchar() - _r0) / 1e9
x = torch.randn(1024, 1024, device="cuda")
y = torch.randn(1 << 20, device="cuda")
(x @ x).sum().item()                        # warm-up OUTSIDE any section: must not be recorded
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0")
lib = ktrace.load()
out["ready"] = int(lib.nvrx_ktrace_ready())
_ = Detector.rings                          # the device side exists: the tracer's thread appends to these rings from now on
lib.nvrx_ktrace_tap(1)                      # test-only: a copy of every duration the rings are given
c0 = ktrace.counters()
out["pending_before"] = int(lib.nvrx_ktrace_pending())   # the warm-up matmul ran outside a section: nothing was recorded
REPS = 24
for i in range(REPS):
    with Detector.detection_section("step", profile_cuda=True):
        z = x @ x
        w = torch.relu(y) + 1.0
    torch.sigmoid(y)                        # between sections: not traced
report = Detector.generate_report()         # waits for the sections' kernels (nvrx_ktrace_sync), not for the device
c1 = ktrace.counters()
out["counters"] = {k: c1[k] - c0[k] for k in ("enqueued", "arrived", "delivered", "lost_no_row", "sink_errors")}
# the durations the tracer's thread handed to the rings (the tap), per key
recs = ktrace.drain_all()
lib.nvrx_ktrace_tap(0)
names = {int(k): ktrace.key_name(int(k)) for k in np.unique(recs["key"])}
out["names"] = sorted(names.values())
per_key = {names[int(k)]: recs["us"][recs["key"] == k] for k in np.unique(recs["key"])}
prof = Detector.cupti_manager.cupti_ext
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
Detector.shutdown()
print("RESULT " + json.dumps(out))
'''


def _run(script, env_extra=None, timeout=150):
    env = dict(os.environ)
    for k in ("NVRX_GPU_TIMING", "WORLD_SIZE", "RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + script], capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.gpu
def test_kernels_are_traced_by_name_and_scored():
    out = _run(SCRIPT)
    print(f"[ktrace] registration {out['register_s']:.2f} s / {out['register_read_gb']:.2f} GB read "
          f"({out['hidden_libraries']} libraries kept out of the SDK's tool search), first HIP call "
          f"{out['first_hip_call_s']:.2f} s / {out['first_hip_call_read_gb']:.2f} GB read")
    # the start-up that used to take minutes where storage is cold: seconds, and next to nothing read
    assert out["register_s"] < 30.0 and out["first_hip_call_s"] < 30.0, out
    assert out["register_read_gb"] < 1.0 and out["first_hip_call_read_gb"] < 1.0, out
    assert out["hidden_libraries"] > 0
    assert out["ready"] == 1
    assert out["pending_before"] == 0
    c = out["counters"]
    assert c["enqueued"] == c["arrived"] == c["delivered"] >= 3 * 24 and c["lost_no_row"] == c["sink_errors"] == 0, c
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


GRAPH_SCRIPT = r'''
import faulthandler, json, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(140, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import nvrx_straggler                      # registers the tracer before HIP starts
import nvrx_cupti_module as cupti_module
import torch
from torch import nn

torch.cuda.set_device(0)
model = nn.Sequential(nn.Linear(256, 256, bias=False), nn.ReLU(), nn.Linear(256, 64, bias=False), nn.Sigmoid()).to("cuda", torch.float32)
x = torch.randn(256, 256, device="cuda")
# PyTorch's capture recipe: one eager pass on a side stream first (BLAS handles / workspaces exist before capture)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(2):
        model(x)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g), torch.no_grad():
    _ = model(x)
torch.cuda.synchronize()

prof = cupti_module.KernelTraceProfiler()
prof.initialize()
prof.start()
g.replay()
torch.cuda.synchronize()
with_graph = prof.get_stats()
prof.reset()
with torch.no_grad():
    _ = model(x)
torch.cuda.synchronize()
no_graph = prof.get_stats()
prof.reset()
# REPS replays against REPS eager passes: the counts have to follow
REPS = 7
for _ in range(REPS):
    g.replay()
torch.cuda.synchronize()
with_graph_n = prof.get_stats()
prof.reset()
prof.stop()
g.replay()                                  # after stop(): not recorded
torch.cuda.synchronize()
after_stop = prof.get_stats()
prof.shutdown()
prof.close()
plain = lambda st: {k: int(v.num_calls) for k, v in st.items()}
print("RESULT " + json.dumps({"with_graph": plain(with_graph), "no_graph": plain(no_graph), "with_graph_n": plain(with_graph_n),
                              "after_stop": plain(after_stop), "reps": REPS,
                              "medians": {k: [with_graph[k].median, no_graph[k].median] for k in no_graph if k in with_graph}}))
'''


@pytest.mark.gpu
def test_kernels_of_a_replayed_graph_are_recorded_like_the_same_kernels_launched_one_by_one():
    """Twin of the reference's tests/straggler/unit/test_cupti_ext.py:125-174 (``test_with_cuda_graph``): the forward pass of
    the same four-layer model, once as a replayed graph and once eagerly, with the profiler running -- every kernel key of
    the eager pass must be among the keys of the replay with the same ``num_calls``.  rocprofiler-sdk reports the
    dispatches of a graph launch one by one with the kernel ids of the captured kernels, so key (name + launch geometry)
    and count carry over; the test also replays seven times (counts x 7) and once after ``stop()`` (nothing)."""
    out = _run(GRAPH_SCRIPT)
    ng, wg = out["no_graph"], out["with_graph"]
    print("[ktrace graph] eager keys:", ng, "\n[ktrace graph] replay keys:", wg)
    assert ng, "the eager forward pass recorded no kernel"
    for k, n in ng.items():                  # the reference's assertion, word for word
        assert k in wg, (k, sorted(wg))
        assert n == wg[k], (k, n, wg[k])
    assert sum(ng.values()) >= 4             # two GEMMs, ReLU, Sigmoid
    for k, n in wg.items():
        assert out["with_graph_n"].get(k) == out["reps"] * n, (k, out["with_graph_n"])
    assert out["after_stop"] == {}
    for k, (a, b) in out["medians"].items():  # the same kernel on the same data: durations of the same order
        assert 0.2 < a / b < 5.0, (k, a, b)


BASIC_SCRIPT = r'''
import faulthandler, json, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(140, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import nvrx_straggler                      # registers the tracer before HIP starts
import nvrx_cupti_module as cupti_module
from nvrx_straggler import ktrace
import torch

plain = lambda st: {k: int(v.num_calls) for k, v in st.items()}
out = {}
prof = cupti_module.CuptiProfiler()
a = torch.randn(1000, 1000, device="cuda")
b = torch.randn(1000, 1000, device="cuda")
torch.cuda.synchronize()
prof.initialize()
prof.start()
c0 = ktrace.counters()
torch.matmul(a, b)                          # the FIRST matmul of the process: the BLAS library sets its workspace up
torch.cuda.synchronize()
out["first"] = plain(prof.get_stats())
prof.stop()
out["after_stop"] = plain(prof.get_stats())
prof.reset()
out["after_reset"] = plain(prof.get_stats())
# explicit memsets / device-to-device copies inside a traced window: hipMemsetAsync and hipMemcpyAsync run as ROCclr blit kernels
prof.start()
raw = torch.empty(1 << 22, dtype=torch.uint8, device="cuda")
raw.zero_()                                 # fill
dst = torch.empty_like(a)
dst.copy_(a)                                # device-to-device copy
torch.matmul(a, b)
torch.cuda.synchronize()
out["with_memops"] = plain(prof.get_stats())
c1 = ktrace.counters()
out["blit_skipped"] = c1["blit_skipped"] - c0["blit_skipped"]
prof.reset()
# on request the runtime's blit kernels are recorded like any other kernel
ktrace.load().nvrx_ktrace_include_blits(1)
raw.zero_()
dst.copy_(a)
torch.cuda.synchronize()
out["blits_included"] = plain(prof.get_stats())
ktrace.load().nvrx_ktrace_include_blits(0)
prof.stop()
prof.shutdown()
prof.close()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_memsets_and_memcpys_are_not_kernels_only_the_matmul_is_a_key():
    """Twin of the reference's tests/straggler/unit/test_cupti_ext.py:22-47 (``test_basic_kernel_tracking``): profiler on,
    ONE matmul, ``len(get_stats()) == 1`` with ``num_calls == 1``, still there after ``stop()``, gone after ``reset()``.
    CUPTI's CONCURRENT_KERNEL activity kind (the only one the reference enables, CuptiProfiler.cpp:118,179) has no memset or
    memcpy records; ROCm runs those as ROCclr blit kernels (``__amd_rocclr_fillBufferAligned`` for the BLAS workspace at the
    first matmul was the second key that failed this scenario in round 5) and the tracer leaves them out -- counted, and
    recorded when asked for (``nvrx_ktrace_include_blits``)."""
    out = _run(BASIC_SCRIPT)
    print("[ktrace basic]", json.dumps(out))
    first = out["first"]
    assert len(first) == 1, first                                   # the reference's assertion
    (name, calls), = first.items()
    assert calls == 1 and ("Cijk" in name or "gemm" in name.lower()), first
    assert out["after_stop"] == first and out["after_reset"] == {}
    memops = out["with_memops"]
    assert not any("__amd_rocclr_" in k for k in memops), memops
    assert memops.get(name) == 1, memops
    # fills and copies went by inside the window (whether the runtime carries one out as a blit kernel or on an SDMA engine is its
    # choice; whatever was a kernel was counted as left out)
    assert out["blit_skipped"] >= 1, out
    assert any("__amd_rocclr_" in k for k in out["blits_included"]), out["blits_included"]


COLLECTIVE_SCRIPT = r'''
import faulthandler, json, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(170, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
import nvrx_straggler                      # WORLD_SIZE=2 is in the environment and HIP is not up: this settles the GPU-timing mode
from nvrx_straggler import Detector, Statistic, ktrace
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)                    # both ranks share the one MI355X of the box (gloo group; RCCL refuses that)
dist.init_process_group("gloo", init_method=os.environ["NVRX_TEST_INIT"], rank=rank, world_size=2)
Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=True, node_name=f"node{rank}")
cycles = int(float(os.environ["NVRX_TEST_CYCLES"]) * (1.25 if rank == 1 else 1.0))   # rank 1: the same kernel, 25 % slower
t = torch.ones(4096, device="cuda")
for i in range(40):
    with Detector.detection_section("train_step", profile_cuda=True):
        torch.cuda._sleep(cycles)           # the step's compute: one kernel, one key, duration set by its argument
        dist.all_reduce(t)                  # the step ends in a collective: nobody leaves before the slowest rank arrives
torch.cuda.synchronize()
local = Detector.cupti_manager.get_results()      # this rank's kernel statistics (every rank; the report exists on rank 0 only)
out = {"mode": ktrace.timing_mode(), "note": ktrace.mode_note(),
       "keys": {k: [float(v.median), int(v.num_calls)] for k, v in local.items()}}
report = Detector.generate_report()
if rank == 0:
    out["gpu_rel"] = {str(k): float(v) for k, v in report.gpu_relative_perf_scores.items()}
    out["flagged"] = sorted(s.rank for s in report.identify_stragglers(gpu_rel_threshold=0.85)["straggler_gpus_relative"])
    out["report_keys"] = sorted(report.local_kernel_summaries)
else:
    assert report is None
dist.barrier()
Detector.shutdown()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _run_two_ranks(env_extra, cycles, script=None, world=2, timeout=200):
    import socket
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.pop("NVRX_GPU_TIMING", None)
        env.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "NVRX_TEST_INIT": f"tcp://127.0.0.1:{port}",
                    "NVRX_TEST_CYCLES": str(cycles), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, "-c", f"REPO = {REPO!r}\n" + (script or COLLECTIVE_SCRIPT)], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, so[-2000:] + "\n" + se[-3000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):]))
    return outs


@pytest.mark.gpu
def test_a_step_that_ends_in_a_collective_per_kernel_default_sees_the_slow_gpu_region_timing_does_not():
    """tests/test_host_logic.py::test_region_timing_flattens_gpu_scores_when_the_region_holds_a_collective on REAL kernels:
    two processes, rank 1's compute kernel takes 25 % longer, the GPU-timed section ends in an all-reduce.

    * default of a multi-rank job (``WORLD_SIZE=2``, nothing else set): ``ktrace.timing_mode()`` picks ``kernels`` -- the
      reference's data model (CuptiProfiler.cpp:168-207, reporting.py:237-253) -- and rank 1 scores 1/1.25 = 0.8 and is
      flagged at 0.85, whatever the collective's wait looks like;
    * ``NVRX_GPU_TIMING=stamp`` (one row per REGION): both regions last as long as the slow rank's compute plus the
      exchange, both GPUs score ~1.0, nothing is flagged -- the deviation INTEGRATION.md documents.

    The two ranks share the box's one GPU, so the collective is gloo's (RCCL refuses two ranks on one device) and no RCCL
    kernel can show up here; the ``ncclDev`` filter is pinned on the kernel names of the installed librccl.so instead
    (tests/test_host_logic.py::test_rccl_kernel_names_of_the_installed_library_are_filtered)."""
    cycles = 4.0e6
    k0, k1 = _run_two_ranks({}, cycles)
    print("[ktrace collective] per-kernel:", k0["mode"], k0["note"], k0.get("gpu_rel"), sorted(k0["keys"]))
    assert k0["mode"] == k1["mode"] == "kernels", (k0["mode"], k0["note"], k1["mode"], k1["note"])
    spin = [k for k in k0["keys"] if "spin" in k.lower() or "sleep" in k.lower()]
    assert spin, sorted(k0["keys"])
    assert k0["keys"][spin[0]][1] == 40 and k1["keys"][spin[0]][1] == 40       # one record per step under the same key
    ratio = k0["keys"][spin[0]][0] / k1["keys"][spin[0]][0]
    assert abs(ratio - 0.8) < 0.03, (ratio, k0["keys"], k1["keys"])
    assert abs(k0["gpu_rel"]["0"] - 1.0) < 0.03 and abs(k0["gpu_rel"]["1"] - 0.8) < 0.04, k0["gpu_rel"]
    assert k0["flagged"] == [1]
    r0, r1 = _run_two_ranks({"NVRX_GPU_TIMING": "stamp"}, cycles)
    print("[ktrace collective] per-region:", r0["mode"], r0.get("gpu_rel"), sorted(r0["keys"]))
    assert r0["mode"] == r1["mode"] == "stamp"
    assert r0["report_keys"] == ["hipevent::train_step"] and len(r0["keys"]) == 1
    assert min(r0["gpu_rel"].values()) > 0.9, r0["gpu_rel"]                   # the slow GPU is invisible ...
    assert r0["flagged"] == []                                                 # ... and not flagged


AGREE_SCRIPT = r'''
import faulthandler, json, logging, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(170, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
records = []
class _Grab(logging.Handler):
    def emit(self, record):
        records.append(record.getMessage())
logging.getLogger("nvrx_straggler").addHandler(_Grab())
logging.getLogger("nvrx_straggler").setLevel(logging.INFO)
import nvrx_straggler                      # rank 0: WORLD_SIZE=2 -> kernels; rank 1: NVRX_GPU_TIMING=stamp in its environment
from nvrx_straggler import Detector, Statistic, ktrace
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method=os.environ["NVRX_TEST_INIT"], rank=rank, world_size=2)
Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=True, node_name=f"node{rank}")
mode_before = ktrace.timing_mode()
cycles = int(float(os.environ["NVRX_TEST_CYCLES"]) * (1.25 if rank == 1 else 1.0))
reports = []
for window in range(3):
    for i in range(12):
        with Detector.detection_section("train_step", profile_cuda=True):
            torch.cuda._sleep(cycles)       # (no collective inside: a region's time is this rank's own compute)
    rep = Detector.generate_report()
    if rank == 0:
        reports.append({"gpu_rel": {str(k): float(v) for k, v in rep.gpu_relative_perf_scores.items()},
                        "keys": sorted(rep.local_kernel_summaries)})
    dist.barrier()
out = {"mode_before": mode_before, "mode_after": ktrace.timing_mode(), "note": ktrace.mode_note(), "reports": reports,
       "log": [m for m in records if "nvrx straggler" in m]}
Detector.shutdown()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_ranks_that_time_gpu_work_differently_agree_on_one_mode_at_their_first_report():
    """one rank on region stamps (here: ``NVRX_GPU_TIMING=stamp`` in ITS environment, the stand-in for
    "its HIP runtime was up before the import"), the other on per-kernel keys -- no key in common, every relative GPU score
    NaN, not a word.  Now the ranks MIN-reduce their mode at the first collective report: the per-kernel rank swaps its
    profiler for the region one (the tracer's sink comes off the rings), both log it, and from the next window on the
    relative GPU scores are finite and see the 25 % slower rank."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.pop("NVRX_GPU_TIMING", None)
        env.update({"RANK": str(rank), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "NVRX_TEST_INIT": f"tcp://127.0.0.1:{port}",
                    "NVRX_TEST_CYCLES": "4.0e6", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if rank == 1:
            env["NVRX_GPU_TIMING"] = "stamp"
        procs.append(subprocess.Popen([sys.executable, "-c", f"REPO = {REPO!r}\n" + AGREE_SCRIPT], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=200)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, so[-2000:] + "\n" + se[-3000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):]))
    r0, r1 = outs
    print("[ktrace agree]", r0["mode_before"], "->", r0["mode_after"], r0["reports"][-1])
    assert (r0["mode_before"], r1["mode_before"]) == ("kernels", "stamp")
    assert r0["mode_after"] == r1["mode_after"] == "stamp" and "another rank" in r0["note"]
    assert any("measured per kernel" in m for m in r0["log"]) and any("measured per profiled region" in m for m in r1["log"])
    assert sum("all ranks time GPU work per profiled region" in m for m in r0["log"]) == 1
    assert any("they fall back to region timing as well" in m for m in r1["log"])
    import math

    last = r0["reports"][-1]
    assert last["keys"] == ["hipevent::train_step"]
    assert all(math.isfinite(v) for v in last["gpu_rel"].values()), last
    assert abs(last["gpu_rel"]["0"] - 1.0) < 0.03 and abs(last["gpu_rel"]["1"] - 0.8) < 0.04, last


BUDGET_SCRIPT = r'''
import faulthandler, json, logging, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(170, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
import nvrx_straggler                      # WORLD_SIZE=2 and HIP is not up: per-kernel tracing is registered now
from nvrx_straggler import Detector, ktrace
import torch
import torch.distributed as dist

records = []
class _Grab(logging.Handler):
    def emit(self, record):
        records.append(record.getMessage())
log = logging.getLogger("nvrx_straggler.straggler")
log.addHandler(_Grab()); log.setLevel(logging.INFO)
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)                    # both ranks share the one MI355X of the box (gloo group)
dist.init_process_group("gloo", init_method=os.environ["NVRX_TEST_INIT"], rank=rank, world_size=2)
# a tiny budget so that the per-dispatch cost of tracing (about a microsecond each) certainly exceeds it on the rank whose step
# is nothing BUT tiny kernels; the other rank's step is one long kernel plus a few tiny ones
Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=True, node_name=f"node{rank}",
                    report_time_interval=3600.0, kernel_trace_budget_pct=float(os.environ["NVRX_TEST_BUDGET"]))
x = torch.ones(256, device="cuda")
traced_entries = []
c_prev = ktrace.counters()["enqueued"]
for i in range(40):
    with Detector.detection_section("train_step", profile_cuda=True):
        if rank == 0:
            for _ in range(400):
                x.add_(1.0)                 # 400 tiny dispatches
            torch.cuda.synchronize()
        else:
            torch.cuda._sleep(int(os.environ["NVRX_TEST_CYCLES"]))
            for _ in range(5):
                x.add_(1.0)
            torch.cuda.synchronize()
    c_now = ktrace.counters()["enqueued"]
    if c_now != c_prev:
        traced_entries.append(i)
    c_prev = c_now
    assert Detector.generate_report_if_interval_elapsed() is None
out = {"mode": ktrace.timing_mode(), "every": Detector._trace_every, "cost_pct": Detector.kernel_trace_cost_pct,
       "traced_entries": traced_entries, "iter_interval": Detector.report_interval_tracker.iter_interval,
       "log": [m for m in records if "budget" in m]}
rep = Detector.generate_report()
if rank == 0:
    out["gpu_rel_ranks"] = sorted(rep.gpu_relative_perf_scores)
dist.barrier()
Detector.shutdown()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_both_ranks_trace_at_the_same_thinned_interval_when_one_of_them_exceeds_the_tracing_budget():
    '''``Detector.initialize(kernel_trace_budget_pct=...)`` on real kernels, two processes: rank 0's step is 400 tiny kernels
    (tracing costs it a visible share of the step), rank 1's is one long kernel.  The calibration rides on the interval
    tracker's all-reduce: both ranks end on the SAME multiple, the larger one; afterwards kernels are traced on exactly
    every N-th entry on both, and the report still covers both ranks.'''
    outs = _run_two_ranks({"NVRX_TEST_BUDGET": "0.2"}, cycles=2_000_000, script=BUDGET_SCRIPT)
    r0, r1 = outs
    print("[ktrace budget]", {k: (r0[k], r1[k]) for k in ("every", "cost_pct")}, r0["log"], r1["log"])
    assert r0["mode"] == r1["mode"] == "kernels"
    assert r0["every"] == r1["every"] >= 1 and r0["iter_interval"] == r1["iter_interval"]
    assert r0["cost_pct"] is not None and r1["cost_pct"] is not None
    every = r0["every"]
    for r in (r0, r1):
        after = [e for e in r["traced_entries"] if e > 17]
        assert after == [e for e in range(18, 40) if e % every == 0], (every, r["traced_entries"])
        cal = [e for e in r["traced_entries"] if e <= 16]
        assert cal == [0, 1, 3, 5, 7, 9, 11, 13, 15], r["traced_entries"]
        assert len(r["log"]) == 1
    assert r0["gpu_rel_ranks"] == [0, 1]


PROFILER_SCENARIOS_SCRIPT = r'''
import faulthandler, json, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(170, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import nvrx_straggler                      # registers the tracer before HIP starts
import nvrx_cupti_module as cupti_module
from nvrx_straggler.cupti import CuptiManager
import torch

plain = lambda st: {k: int(v.num_calls) for k, v in st.items()}
out = {}
a = torch.randn(1000, 1000, device="cuda")
b = torch.randn(1000, 1000, device="cuda")
torch.matmul(a, b)                          # (the BLAS library's one-time set-up happens outside every scenario)
torch.cuda.synchronize()

def fresh(**kw):
    p = cupti_module.CuptiProfiler(**kw)
    p.initialize()
    return p

def done(p):
    p.shutdown(); p.close()

# start / stop / start: only what ran while started is counted (test_cupti_ext.py:50-76)
p = fresh()
p.start(); torch.matmul(a, b); torch.cuda.synchronize(); p.stop()
torch.matmul(a, b); torch.cuda.synchronize()
p.start(); torch.matmul(a, b); torch.cuda.synchronize(); p.stop()
out["start_stop"] = plain(p.get_stats())
# reset empties the results (:79-95)
p.reset()
out["after_reset"] = plain(p.get_stats())
done(p)
# statsMaxLenPerKernel keeps that many durations per kernel (:98-116)
p = fresh(statsMaxLenPerKernel=7)
p.start()
for _ in range(21):
    torch.matmul(a, b)
torch.cuda.synchronize()
p.stop()
out["max_stats"] = plain(p.get_stats())
# one profiler at a time (:119-122)
try:
    cupti_module.CuptiProfiler()
    out["singleton"] = "no error"
except RuntimeError as e:
    out["singleton"] = str(e)
done(p)
# CuptiManager: nested start / stop, only the outermost pair switches tracing (test_cupti_manager.py:22-47)
m = CuptiManager(); m.initialize()
try:
    m.stop_profiling(); out["stop_without_start"] = "no error"
except Exception as e:
    out["stop_without_start"] = type(e).__name__
m.start_profiling(); m.start_profiling(); m.stop_profiling()
torch.matmul(a, b); torch.cuda.synchronize()
m.stop_profiling()
torch.matmul(a, b); torch.cuda.synchronize()
out["manager_nested"] = plain(m.get_results())
m.shutdown()
# ... and everything launched between the outermost pair is captured, nothing before or after (:50-83)
m = CuptiManager(); m.initialize()
for _ in range(100):
    torch.matmul(a, b)
m.start_profiling()
for _ in range(50):
    torch.matmul(a, b)
m.start_profiling(); m.stop_profiling()
for _ in range(50):
    torch.matmul(a, b)
m.stop_profiling()
for _ in range(100):
    torch.matmul(a, b)
torch.cuda.synchronize()
out["manager_all_started"] = plain(m.get_results())
m.shutdown()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_profiler_and_manager_scenarios_of_the_reference_suite_in_per_kernel_mode():
    """Own-code twins of the rest of the reference's GPU-only profiler tests (tests/straggler/unit/test_cupti_ext.py:50-122,
    test_cupti_manager.py:22-83), in the mode whose data model they describe (``NVRX_GPU_TIMING=kernels``): start / stop /
    start counts two matmuls, reset empties, ``statsMaxLenPerKernel=7`` keeps 7 of 21, a second profiler is refused, nested
    ``start_profiling`` calls trace from the outermost start to the outermost stop -- 1 matmul, then exactly 100 of 300.
    Every assertion below is one of theirs (``len(stats) == 1`` and the ``num_calls``)."""
    out = _run(PROFILER_SCENARIOS_SCRIPT, timeout=200)
    print("[ktrace scenarios]", {k: (list(v.values()) if isinstance(v, dict) else v) for k, v in out.items()})
    assert list(out["start_stop"].values()) == [2], out["start_stop"]
    assert out["after_reset"] == {}
    assert list(out["max_stats"].values()) == [7], out["max_stats"]
    assert "Only one CuptiProfiler instance is allowed" in out["singleton"]
    assert out["stop_without_start"] == "RuntimeError"
    assert list(out["manager_nested"].values()) == [1], out["manager_nested"]
    assert list(out["manager_all_started"].values()) == [100], out["manager_all_started"]
    assert set(out["start_stop"]) == set(out["max_stats"]) == set(out["manager_nested"]) == set(out["manager_all_started"])


EIGHT_RANKS_SCRIPT = r'''
import faulthandler, json, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(280, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
import nvrx_straggler                      # WORLD_SIZE=8, HIP not up: per-kernel tracing registers now
from nvrx_straggler import Detector, Statistic, ktrace
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                    # eight processes share the one MI355X of the box (gloo group)
dist.init_process_group("gloo", init_method=os.environ["NVRX_TEST_INIT"], rank=rank, world_size=world)
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(512, 1024), torch.nn.GELU(), torch.nn.LayerNorm(1024), torch.nn.Linear(1024, 512)).cuda()
opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9)
x = torch.randn(256, 512, device="cuda")
extra = torch.randn(300, 300, device="cuda")

def step():
    loss = model(x).square().mean()
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    if rank % 3 == 0:
        (extra @ extra).sum()               # a kernel only SOME ranks launch: NaN for the others, -1 sentinel in the MIN
    torch.cuda._sleep(int(os.environ["NVRX_TEST_CYCLES"]) * (3 if rank == 5 else 1))   # rank 5: the same kernel, three times as long

for _ in range(3):
    step()
torch.cuda.synchronize()
dist.barrier()
# BASELINE config #4: individual + relative GPU scores from kernel timing, a collective report EVERY step
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name=f"node{rank}", kernel_trace_budget_pct=0.0)
windows = []
plain = lambda d: {k: {"MED": float(v[Statistic.MED]), "AVG": float(v[Statistic.AVG]), "NUM": int(v[Statistic.NUM])} for k, v in d.items()}
for t in range(4):
    with Detector.detection_section("train_step", profile_cuda=True):
        step()
        step()
    torch.cuda.synchronize()
    mine = {"kernels": plain(Detector._get_kernel_summaries()), "sections": plain(Detector._get_section_summaries())}   # (peeks: nothing is reset)
    rep = Detector.generate_report()
    if rank == 0:
        mine["report"] = {"gpu_rel": {str(r): float(v) for r, v in rep.gpu_relative_perf_scores.items()},
                          "gpu_indiv": {str(r): float(v) for r, v in rep.gpu_individual_perf_scores.items()},
                          "sec_rel": {n: {str(r): float(v) for r, v in d.items()} for n, d in rep.section_relative_perf_scores.items()},
                          "flagged": sorted(s.rank for s in rep.identify_stragglers()["straggler_gpus_relative"])}
    else:
        assert rep is None
    windows.append(mine)
out = {"mode": ktrace.timing_mode(), "windows": windows, "counters": ktrace.counters()}
dist.barrier()
Detector.shutdown()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_config4_at_the_eight_rank_table_shape_on_real_kernel_keys_matches_the_oracle():
    '''BASELINE config #4 at the R = 8 table shape in per-kernel mode: eight processes, a small training
    step of real kernels (forward, backward, SGD with momentum: tens of kernel keys by their real mangled names), a kernel
    only three of the ranks launch, one kernel three times as long on rank 5, a collective report EVERY step.  Each rank
    hands back the kernel / section summaries it saw (statistics computed on the device); the oracle's restatement of
    ``ReportGenerator`` (reporting.py:154-554) scores those eight summary sets, and rank 0's report must equal it: every GPU
    and section score within 1e-4 relative (north_star's tolerance), NaN where the oracle has NaN, the same flagged set.'''
    import math

    import numpy as np

    from oracle import oracle

    outs = _run_two_ranks({}, cycles=400_000, script=EIGHT_RANKS_SCRIPT, world=8, timeout=400)
    assert all(o["mode"] == "kernels" for o in outs)
    port = oracle.RefPortReportGenerator(8)
    nkeys = set()
    for t in range(4):
        ks = [outs[r]["windows"][t]["kernels"] for r in range(8)]
        ss = [outs[r]["windows"][t]["sections"] for r in range(8)]
        nkeys |= {k for d in ks for k in d}
        exp = port.generate_reports(ss, ks)
        got = outs[0]["windows"][t]["report"]

        def close(a, b, what):
            if math.isnan(b):
                assert math.isnan(a), (t, what, a, b)
            else:
                assert abs(a - b) <= 1e-4 * abs(b), (t, what, a, b)

        for r in range(8):
            close(got["gpu_rel"][str(r)], exp["gpu_rel"][r], ("gpu_rel", r))
            close(got["gpu_indiv"][str(r)], exp["gpu_indiv"][r], ("gpu_indiv", r))
            close(got["sec_rel"]["train_step"][str(r)], exp["sec_rel"]["train_step"][r], ("sec_rel", r))
        assert got["flagged"] == sorted(oracle.identify_stragglers(exp["gpu_rel"], 0.75)), (t, got["flagged"], exp["gpu_rel"])
    some = [k for k in nkeys if "300" in k or "Cijk" in k]
    print(f"[ktrace 8 ranks] {len(nkeys)} kernel keys over the job; window 3 gpu_rel:", outs[0]["windows"][3]["report"]["gpu_rel"],
          "flagged", outs[0]["windows"][3]["report"]["flagged"])
    assert len(nkeys) >= 12, sorted(nkeys)
    # the extra matmul exists on ranks 0, 3, 6 only
    only_some = [k for k in nkeys if sum(k in outs[r]["windows"][0]["kernels"] for r in range(8)) == 3]
    assert only_some, "the kernel that only three ranks launch did not show up as a key of exactly three ranks"
    for o in outs:
        c = o["counters"]
        assert c["lost_no_row"] == 0 and c["sink_errors"] == 0 and c["forgiven"] == 0, c
