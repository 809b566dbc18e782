"""GPU tests of the per-kernel tracer's native data path (VERDICT r4 "next" item 1).

1. The scenario of tests/test_ktrace_datapath.py -- > 3 x cap launches per key, small batches, a second thread -- fed into
   the DEVICE rings (``nvrx_ring_push_staged`` / ``nvrx_row_alloc`` behind the sink, ``k_scatter`` + ``k_row_stats`` on the
   GPU) and compared with the reference's own ``CuptiProfiler`` (oracle/_ref) on the same launches.
2. REAL kernels in a fresh interpreter (the tool must register before HIP starts): ring capacity 16, 60 launches per key
   in 60 section entries; a tap hands the test the very durations the tracer's thread gave the rings; the reference
   profiler fed with those durations (``statsMaxLenPerKernel`` = 16) must report the same statistics -- the NEWEST 16
   survive (CircularBuffer.h:53-61).  The training thread's part of a report is ``nvrx_ktrace_sync``: timed.
3. ``asynchronous=True`` in per-kernel mode: a report does not wait for the kernels of its window; what is still running
   is counted in the next window, nothing is lost, nothing is counted twice.
"""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import oracle

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fed_launches_reach_the_device_rings_and_match_the_reference_profiler():
    from nvrx_straggler import ktrace

    import test_ktrace_datapath as cpu_twin

    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    cap, K = 32, 40
    rng = np.random.default_rng(5)
    names = [f"_Z{7 + k}dev_kern{k:03d}Pfi" for k in range(K)]
    ids = cpu_twin._fresh_kernel_ids(K)
    for i, n in zip(ids.tolist(), names):
        ktrace.feed_kernel_name(i, n)
    idx, dur, start, blocks = cpu_twin._launches(rng, names, 3 * cap + 5)
    disp = cpu_twin._as_dispatches(ids, idx, dur, start, blocks)
    ktrace.KernelTraceProfiler._live = None
    prof = ktrace.KernelTraceProfiler(statsMaxLenPerKernel=cap, max_keys=K + 8)   # private device rings
    try:
        before = ktrace.counters()

        def feeder():
            for lo in range(0, disp.size, 11):          # small batches: several staging flushes (stage_cap = cap = 32)
                ktrace.feed(disp[lo:lo + 11])

        t = threading.Thread(target=feeder)
        t.start()
        t.join()
        assert prof.harvest(wait=True) == 0
        after = ktrace.counters()
        assert after["delivered"] - before["delivered"] == disp.size and after["sink_errors"] == before["sink_errors"]
        got = prof.get_stats()
        exp = cpu_twin._reference_stats(names, idx, dur, start, blocks, cap)
        assert set(got) == set(exp) and len(got) == K
        for key, (e, n) in exp.items():
            g = got[key]
            assert g.num_calls == n == cap, key
            assert (np.float32(g.min), np.float32(g.max), np.float32(g.median)) == (e[0], e[1], e[2]), key
            assert abs(g.avg - e[3]) <= 2e-4 * abs(e[3]) and abs(g.stddev - e[4]) <= 2e-3 * abs(e[4]), key   # reference sums in f32
        rings = prof._rings
        for k in range(0, K, 7):
            mine = ((dur[idx == k]).astype(np.float32) / np.float32(1000.0))[-cap:]
            row = rings.kernel_row_names[next(n for n in got if n.startswith(names[k] + "_blk_"))]
            assert sorted(rings.read_row(row)[:cap].tolist()) == sorted(mine.tolist()), k
    finally:
        prof.close()
        ktrace.KernelTraceProfiler._live = None


REAL_SCRIPT = r'''
import faulthandler, json, os, sys, time
faulthandler.enable()
faulthandler.dump_traceback_later(170, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import numpy as np
import torch
import nvrx_straggler                      # registers the tracer: no HIP call has happened yet
from nvrx_straggler import Detector, Statistic, ktrace
from nvrx_straggler.straggler import CustomSection
from oracle import oracle

CAP, REPS = 16, 60
CustomSection.max_elapseds_len = CAP        # ring capacity of every row (straggler.py:80), read by Detector.initialize
torch.cuda.set_device(0)
x = torch.randn(512, 512, device="cuda")
y = torch.randn(1 << 18, device="cuda")
(x @ x).sum().item(); torch.relu(y); torch.sigmoid(y)      # warm-up outside sections
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0")
lib = ktrace.load()
lib.nvrx_ktrace_set_max_pending(1 << 16)
_ = Detector.rings                          # device side (and the sink) exist before the tap opens
lib.nvrx_ktrace_tap(1)
c0 = ktrace.counters()
for i in range(REPS):
    with Detector.detection_section("step", profile_cuda=True):
        z = x @ x
        w = torch.relu(y)
        v = torch.sigmoid(y)
t0 = time.perf_counter_ns()
missing = Detector.cupti_manager.harvest(wait=True)          # what generate_report() does first: waits for the window's kernels
t_wait = (time.perf_counter_ns() - t0) / 1e3
t_idle = 1e30
for _ in range(5):                          # (the best of five: one preempted call must not decide a 50 us bound)
    t0 = time.perf_counter_ns()
    again = Detector.cupti_manager.harvest(wait=True)        # everything has arrived: the training thread's steady-state cost
    t_idle = min(t_idle, (time.perf_counter_ns() - t0) / 1e3)
c1 = ktrace.counters()
recs = ktrace.drain_all()                   # the tap: every duration the tracer's thread handed to the rings, in order
lib.nvrx_ktrace_tap(0)
report = Detector.generate_report()
hist = {}
for k in np.unique(recs["key"]):
    hist[ktrace.key_name(int(k))] = recs["us"][recs["key"] == k]
checks = []
R = oracle.ref_lib()
assert R is not None and R.ref_profiler_create(1 << 20, 8, CAP) == 0
R.ref_profiler_initialize(); R.ref_profiler_start()
t = 1000
for k, u in zip(recs["key"].tolist(), recs["us"].tolist()):   # the reference's own profiler, fed with the same durations in the same order
    name = ktrace.key_name(k)
    base, dims = name.rsplit("_blk_", 1)
    b, g = dims.split("_grid_")
    bx, by, bz = (int(v) for v in b.split("_")); gx, gy, gz = (int(v) for v in g.split("_"))
    ns = int(round(float(np.float32(u)) * 1000.0))
    R.ref_profiler_launch(base.encode(), bx, by, bz, gx, gy, gz, t, t + ns)
    t += ns + 10
R.ref_profiler_stop()
buf = np.empty(5, dtype=np.float32)
ref = {}
for i in range(R.ref_profiler_get_stats()):
    n = R.ref_profiler_stats(i, buf.ctypes.data)
    ref[R.ref_profiler_key(i).decode()] = ([float(v) for v in buf], int(n))
R.ref_profiler_destroy()
got = {k: [float(v[s]) for s in (Statistic.MIN, Statistic.MAX, Statistic.MED, Statistic.AVG, Statistic.STD, Statistic.NUM)]
       for k, v in report.local_kernel_summaries.items()}
newest = {k: sorted(float(np.float32(v)) for v in h[-CAP:]) for k, h in hist.items()}
stored = {k: sorted(float(v) for v in Detector.rings.read_row(Detector.rings.kernel_row_names[k])[:CAP]) for k in hist}
out = {"missing": [int(missing), int(again)], "wait_us": t_wait, "idle_us": t_idle, "counters": [c0, c1],
       "hist_len": {k: int(h.size) for k, h in hist.items()}, "ref": ref, "got": got, "newest": newest, "stored": stored,
       "section_num": int(report.local_section_summaries["step"][Statistic.NUM]),
       "gpu_rel": float(report.gpu_relative_perf_scores[0])}
Detector.shutdown()
print("RESULT " + json.dumps(out))
'''


def _run(script, env_extra=None, timeout=200):
    env = dict(os.environ)
    for k in ("NVRX_GPU_TIMING", "WORLD_SIZE", "RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + script], capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.gpu
def test_real_kernels_overflow_their_rings_and_the_newest_survive_like_the_reference():
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    out = _run(REAL_SCRIPT)
    c0, c1 = out["counters"]
    print(f"[ktrace datapath] harvest: {out['wait_us']:.1f} us waiting for the window's kernels, {out['idle_us']:.2f} us when "
          f"everything has arrived; pump flushes {c1['pump_flushes'] - c0['pump_flushes']}, counted dispatches "
          f"{c1['enqueued'] - c0['enqueued']}, own-kernel records left out {c1['own_skipped'] - c0['own_skipped']}")
    assert out["missing"] == [0, 0]
    assert c1["counting"] == 1
    assert c1["enqueued"] - c0["enqueued"] == c1["arrived"] - c0["arrived"] >= 3 * 60
    assert c1["sink_errors"] == c0["sink_errors"] and c1["lost_no_row"] == c0["lost_no_row"]
    assert out["idle_us"] < 50.0, out["idle_us"]                   # a counter comparison + the ctypes call, no per-record work
    assert out["section_num"] == 16                                   # the section's own ring is 16 deep as well
    assert len(out["hist_len"]) >= 3 and all(n >= 60 for n in out["hist_len"].values()), out["hist_len"]   # > 3 x cap per key
    assert set(out["got"]) == set(out["ref"]) == set(out["hist_len"])
    for key, (e, n) in out["ref"].items():
        g = out["got"][key]
        assert g[5] == n == 16, (key, g, n)
        assert g[0] == e[0] and g[1] == e[1] and g[2] == e[2], (key, g, e)          # MIN MAX MED bit-exact
        assert abs(g[3] - e[3]) <= 2e-4 * abs(e[3]) and abs(g[4] - e[4]) <= 2e-3 * max(abs(e[4]), 1e-3), (key, g, e)
        assert out["stored"][key] == out["newest"][key], key                         # the ring holds the NEWEST 16
    assert abs(out["gpu_rel"] - 1.0) < 1e-6


ASYNC_SCRIPT = r'''
import faulthandler, json, os, sys, time
faulthandler.enable()
faulthandler.dump_traceback_later(170, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import torch
import nvrx_straggler
from nvrx_straggler import Detector, Statistic, ktrace

torch.cuda.set_device(0)
torch.cuda._sleep(1000); torch.cuda.synchronize()
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0", asynchronous=True)
N, CYCLES = 12, int(8.0e6)                  # 12 long one-block kernels: the report comes while most of them are still queued
nums, waits, sync_s = [], [], 0.0
for window in range(3):
    if window < 2:
        for i in range(N):
            with Detector.detection_section("step", profile_cuda=True):
                torch.cuda._sleep(CYCLES)
    t0 = time.perf_counter()
    report = Detector.generate_report()     # asynchronous: enqueues, does not wait for the sleeps
    waits.append(time.perf_counter() - t0)
    if window == 1:
        t0 = time.perf_counter()
        torch.cuda.synchronize()            # let the second window's kernels finish before the last report
        sync_s = time.perf_counter() - t0
        ktrace.load().nvrx_ktrace_sync(5.0)
    ks = report.local_kernel_summaries      # first read waits for the report's own kernels only
    key = [k for k in ks if "spin" in k.lower() or "sleep" in k.lower()]
    nums.append(int(ks[key[0]][Statistic.NUM]) if key else 0)
    if window == 0:
        first_gpu = {str(k): float(v) for k, v in report.gpu_relative_perf_scores.items()}
c = ktrace.counters()
Detector.shutdown()
print("RESULT " + json.dumps({"nums": nums, "waits": waits, "counters": c, "first_gpu": first_gpu, "sync_s": sync_s}))
'''


@pytest.mark.gpu
def test_asynchronous_reports_in_per_kernel_mode_do_not_wait_and_lose_nothing():
    out = _run(ASYNC_SCRIPT)
    print("[ktrace async]", out["nums"], [round(w * 1e3, 2) for w in out["waits"]], "ms per generate_report()")
    nums = out["nums"]
    assert sum(nums) == 24, nums                       # every traced kernel is counted exactly once ...
    assert nums[0] < 12 and nums[1] < 24, nums         # ... no report waited for its window's kernels
    assert out["waits"][1] < out["sync_s"] / 4, out    # (the sleeps still queued at the second report took sync_s to drain)
    assert out["counters"]["enqueued"] == out["counters"]["arrived"] + out["counters"]["forgiven"]


MISS_SCRIPT = r'''
import faulthandler, json, os, sys, threading, time
faulthandler.enable()
faulthandler.dump_traceback_later(90, exit=True)   # (the bug this pins was a deadlock: say where)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import torch
import nvrx_straggler
from nvrx_straggler import Detector, Statistic, ktrace

torch.cuda.set_device(0)
x = torch.randn(64, 64, device="cuda")
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0", asynchronous=True)
lib = ktrace.load()
stop = []

def drainer():
    # a second drainer beside the pump: while kernels of a window are still completing it keeps emptying the inbox, so that a
    # batch is consumed INSIDE the few microseconds a report holds the tracer with high probability
    while not stop:
        lib.nvrx_ktrace_sync(0.002)

th = threading.Thread(target=drainer, daemon=True)
th.start()
windows, lanes_missed, total = 160, 0, 0
for w in range(windows):
    with Detector.detection_section("step", profile_cuda=True):
        y = x
        for _ in range(300):                 # ~300 tiny kernels: still completing, one every few microseconds, when the report comes
            y = y + 1.0
    if w % 4 == 3:                           # three steady windows (a lane is built), then one whose occupied rows differ
        with Detector.detection_section("sometimes", profile_cuda=False):
            pass
    had_lane = Detector._lane is not None
    report = Detector.generate_report()      # asynchronous: holds the tracer, finds the rows changed, lifts the hold
    lanes_missed += int(had_lane and w % 4 == 3 and w > 4)
    ks = report.local_kernel_summaries
    total += sum(int(v[Statistic.NUM]) for v in ks.values())
stop.append(1)
th.join()
torch.cuda.synchronize()
lib.nvrx_ktrace_sync(5.0)                    # (a completion callback may run a moment after the stream reports the kernel finished)
last = Detector.generate_report().local_kernel_summaries
total += sum(int(v[Statistic.NUM]) for v in last.values())
c = ktrace.counters()
Detector.shutdown()
print("RESULT " + json.dumps({"windows": windows, "lanes_missed": lanes_missed, "total": total, "counters": c}))
'''


@pytest.mark.gpu
def test_an_asynchronous_window_whose_rows_changed_lifts_the_tracers_hold_without_the_contexts_lock():
    """``nvrx_window_report`` (the lane's one C call) holds the tracer while an asynchronous per-kernel report is enqueued; when
    it finds that the set of occupied rows has changed it says MISS and lifts the hold -- which hands the durations parked
    meanwhile to the sink, and the sink takes the context's mutex.  Up to round 6's last session the hold was lifted INSIDE the
    scope that held that mutex: a self-deadlock whenever a batch had been consumed during those microseconds, met by
    tools/soak.py in per-kernel mode once in ~10 000 reports.  Here a second drainer thread empties the inbox all the time
    while hundreds of tiny kernels of the window are still completing, and every fourth window changes the occupied rows under
    a lane: with the old library the process stops in its first dozens of windows (the control is in
    profiles/r06ae_window_miss_deadlock.txt)."""
    out = _run(MISS_SCRIPT, {"NVRX_DEBUG_WINDOW_HOLD_US": "300"}, timeout=150)
    c = out["counters"]
    print("[ktrace miss]", out["lanes_missed"], "lane windows met changed rows;", out["total"], "kernel samples reported")
    assert out["lanes_missed"] >= 20, out                       # the lane really was in place when the rows changed
    assert c["enqueued"] == c["arrived"] + c["forgiven"] and c["forgiven"] == 0 and c["sink_errors"] == 0 and c["lost_no_row"] == 0
    assert out["total"] == c["delivered"], out                   # every delivered duration was reported exactly once


SOAK_SCRIPT = r'''
import faulthandler, json, os, sys, threading, time
faulthandler.enable()
faulthandler.dump_traceback_later(170, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
import torch
import nvrx_straggler
from nvrx_straggler import Detector, Statistic, ktrace
from nvrx_straggler.straggler import CustomSection

CustomSection.max_elapseds_len = 64          # small rings: the tracer's thread flushes often (a scatter launch from ITS thread)
torch.cuda.set_device(0)
xs = [torch.randn(64 * (i + 1), 64, device="cuda") for i in range(6)]   # six shapes: a few dozen kernel keys
side = torch.cuda.Stream()
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0", asynchronous=(os.environ["SOAK_ASYNC"] == "1"))
t_end = time.time() + float(os.environ["SOAK_SECONDS"])
reports = nums = windows = 0
launched = 0
stop = []
def other_thread():                           # launches of a second thread land in whatever section is open: counted and traced alike
    y = torch.randn(4096, device="cuda")
    while not stop:
        torch.tanh(y)
        time.sleep(0.0005)
th = threading.Thread(target=other_thread, daemon=True)
th.start()
i = 0
while time.time() < t_end:
    with Detector.detection_section("a", profile_cuda=True):
        for x in xs[: 1 + i % 6]:
            (x @ x.t()).relu_()
        with torch.cuda.stream(side):        # a second stream inside the section
            torch.sigmoid(xs[0])
    with Detector.detection_section("b", profile_cuda=(i % 3 == 0)):
        xs[i % 6].mul_(1.0001)
    i += 1
    if i % 37 == 0:
        rep = Detector.generate_report()
        reports += 1
        ks = rep.local_kernel_summaries
        nums += sum(int(v[Statistic.NUM]) for v in ks.values())
        assert all(v[Statistic.MIN] <= v[Statistic.MED] <= v[Statistic.MAX] for v in ks.values())
        assert rep.local_section_summaries["a"][Statistic.NUM] >= 1
stop.append(1); th.join()
torch.cuda.synchronize()
ktrace.load().nvrx_ktrace_sync(5.0)
last = Detector.generate_report()
nums += sum(int(v[Statistic.NUM]) for v in last.local_kernel_summaries.values())
c = ktrace.counters()
keys = len(Detector.rings.kernel_row_names)
Detector.shutdown()
print("RESULT " + json.dumps({"entries": i, "reports": reports, "kernel_samples_reported": nums, "keys": keys, "counters": c}))
'''


@pytest.mark.gpu
@pytest.mark.parametrize("asynchronous", [False, True])
def test_tracer_soak_threads_streams_small_rings(asynchronous):
    """A few seconds of everything at once in per-kernel mode: 64-deep rings (the tracer's thread launches a scatter every few
    dozen records), six launch geometries, a second stream and a second launching thread inside the sections, a report
    every 37 entries, synchronous and asynchronous.  Nothing may be lost on the way (every counted dispatch arrives, the
    sink never fails, no row runs out) and the process must come down cleanly."""
    out = _run(SOAK_SCRIPT, {"SOAK_SECONDS": "4", "SOAK_ASYNC": "1" if asynchronous else "0"})
    c = out["counters"]
    print("[ktrace soak]", "async" if asynchronous else "sync", {k: out[k] for k in ("entries", "reports", "kernel_samples_reported", "keys")},
          {k: c[k] for k in ("enqueued", "arrived", "delivered", "own_skipped", "forgiven", "pump_flushes")})
    assert out["entries"] > 200 and out["reports"] >= 5 and out["keys"] >= 8
    assert c["enqueued"] == c["arrived"] + c["forgiven"] and c["forgiven"] == 0, c
    assert c["sink_errors"] == 0 and c["lost_no_row"] == 0 and c["keys_without_row"] == 0, c
    assert c["delivered"] + c["own_skipped"] + c["blit_skipped"] == c["arrived"], c        # (no record with a zero timestamp on this path; fills / copies the soak issues are ROCclr blit kernels: counted, not keys)
    assert 0 < out["kernel_samples_reported"] <= c["delivered"]         # (rings are 64 deep: windows with more launches keep the newest)
