"""The per-kernel tracer's NATIVE data path (libnvrx_ktrace.so) without a GPU: dispatch records fed through
``nvrx_ktrace_feed`` take the path of the rocprofiler-sdk callback thread from ``consume()`` on -- key cache, sink,
per-key overwrite-oldest rings, counters -- and are compared with the REFERENCE's own ``CuptiProfiler`` (compiled from
/root/reference into ``oracle/_ref``, fake CUPTI feed) on the same launches.  The rings here are the checker's NumPy
rings behind the same two function pointers the device rings offer (``tests/oracle_backend.py``); the ``-m gpu`` twin
(``tests/test_gpu_01_ktrace_datapath.py``) runs the very same scenario into the device rings.

What is pinned (reference: cupti_src/CuptiProfiler.cpp:168-207, CircularBuffer.h:53-61):
  * key = "<name>_blk_x_y_z_grid_x_y_z", grid in workgroups; records with a zero timestamp are skipped;
  * per key the NEWEST statsMaxLenPerKernel durations survive (the round-4 queue dropped the newest);
  * duration = (end - start) / 1000.0f in f32;
  * the training thread does nothing per record: ``harvest()`` is a counter comparison.
"""
import threading

import numpy as np
import pytest

from nvrx_straggler import backend, ktrace
from oracle import oracle
from oracle_backend import OracleBackend

_uid = [1 << 40]


def _fresh_kernel_ids(n):
    """kernel ids nobody in this process has used (the tracer's tables are process-wide)."""
    _uid[0] += n
    return np.arange(_uid[0] - n, _uid[0], dtype=np.uint64)


def _launches(rng, names, per_key, shapes=None):
    """A shuffled launch sequence: (kernel index, block, grid-in-blocks, start_ns, end_ns) per launch."""
    K = len(names)
    idx = rng.permutation(np.repeat(np.arange(K), per_key))
    dur = rng.integers(1_000, 5_000_000, idx.size).astype(np.uint64)
    start = np.cumsum(dur) + np.uint64(10_000)
    blocks = shapes if shapes is not None else [((64 << (k % 3)), 1 + (k % 2), 1, 1 + k, 1 + (k % 5), 1) for k in range(K)]
    return idx, dur, start, blocks


def _as_dispatches(ids, idx, dur, start, blocks):
    d = np.zeros(idx.size, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"] = ids[idx]
    for i, k in enumerate(idx.tolist()):
        bx, by, bz, gx, gy, gz = blocks[k]
        d["workgroup"][i] = (bx, by, bz)
        d["grid"][i] = (gx * bx - (bx // 2 if bx > 1 else 0), gy * by, gz * bz)  # work-items; x is NOT a multiple of the block
    d["start_ns"] = start
    d["end_ns"] = start + dur
    return d


def _reference_stats(names, idx, dur, start, blocks, cap):
    """The same launches through the reference's CuptiProfiler (oracle/_ref): key -> (min max median avg stddev, n)."""
    R = oracle.ref_lib()
    assert R.ref_profiler_create(1 << 20, 8, cap) == 0
    try:
        R.ref_profiler_initialize()
        R.ref_profiler_start()
        for i, k in enumerate(idx.tolist()):
            bx, by, bz, gx, gy, gz = blocks[k]
            R.ref_profiler_launch(names[k].encode(), bx, by, bz, gx, gy, gz, int(start[i]), int(start[i] + dur[i]))
        R.ref_profiler_stop()
        out = {}
        buf = np.empty(5, dtype=np.float32)
        for i in range(R.ref_profiler_get_stats()):
            n = R.ref_profiler_stats(i, buf.ctypes.data)
            out[R.ref_profiler_key(i).decode()] = (buf.copy(), n)
        return out
    finally:
        R.ref_profiler_destroy()


@pytest.fixture
def profiler(monkeypatch):
    """A KernelTraceProfiler over the checker's rings, the SDK not involved (``nvrx_ktrace_ready`` is 0 on this box)."""
    made = []

    def make(rows, cap):
        monkeypatch.setattr(ktrace, "_setup_error", None)
        monkeypatch.setattr(ktrace, "setup", lambda *a, **k: None)
        monkeypatch.setattr(ktrace.KernelTraceProfiler, "_live", None)
        backend.set_backend(OracleBackend())
        rings = backend.get_backend().make_rings(1, rows, cap)
        prof = ktrace.KernelTraceProfiler(statsMaxLenPerKernel=cap, rings=rings)
        made.append(prof)
        return prof, rings

    yield make
    for p in made:
        p.close()
    backend.set_backend(None)


def test_per_key_rings_keep_the_newest_durations_like_the_reference(profiler):
    """> 3 x cap launches per key, fed in SMALL batches from a second thread while the 'training' thread only ever calls
    harvest(): statistics of every key equal the reference profiler's on the same launches."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    cap, K = 32, 12
    rng = np.random.default_rng(11)
    names = [f"_Z{6 + k}kernel{k:03d}Pfi" for k in range(K)]
    ids = _fresh_kernel_ids(K)
    for i, n in zip(ids.tolist(), names):
        ktrace.feed_kernel_name(i, n)
    idx, dur, start, blocks = _launches(rng, names, 3 * cap + 9)
    disp = _as_dispatches(ids, idx, dur, start, blocks)
    # a record CUPTI would skip (CuptiProfiler.cpp:182-184): it is counted as arrived, not recorded
    zero = disp[:1].copy()
    zero["start_ns"] = 0
    prof, rings = profiler(K + 4, cap)
    before = ktrace.counters()

    def feeder():
        ktrace.feed(zero)
        for lo in range(0, disp.size, 7):
            ktrace.feed(disp[lo:lo + 7])

    t = threading.Thread(target=feeder)
    t.start()
    t.join()
    assert prof.harvest(wait=True) == 0
    after = ktrace.counters()
    assert after["enqueued"] - before["enqueued"] == disp.size + 1 == after["arrived"] - before["arrived"]
    assert after["delivered"] - before["delivered"] == disp.size
    got = prof.get_stats()
    exp = _reference_stats(names, idx, dur, start, blocks, cap)
    assert set(got) == set(exp) and len(got) == K
    for key, (e, n) in exp.items():
        g = got[key]
        assert g.num_calls == n == cap, key
        assert (np.float32(g.min), np.float32(g.max), np.float32(g.median)) == (e[0], e[1], e[2]), key
        assert abs(g.avg - e[3]) <= 2e-6 * abs(e[3]) and abs(g.stddev - e[4]) <= 2e-5 * abs(e[4]), key
    # and it is the NEWEST cap durations of each key, in the reference's f32 arithmetic
    for k in range(K):
        mine = ((dur[idx == k]).astype(np.float32) / np.float32(1000.0))[-cap:]
        row = rings.kernel_row_names[next(n for n in got if n.startswith(names[k] + "_blk_"))]
        stored = rings.samples[row]
        assert sorted(stored.tolist()) == sorted(mine.tolist()), k


def test_key_format_and_grid_in_workgroups(profiler):
    prof, rings = profiler(8, 16)
    (kid,) = _fresh_kernel_ids(1).tolist()
    ktrace.feed_kernel_name(kid, "_Z4gemmPKfS0_Pf")
    d = np.zeros(3, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"] = kid
    d["workgroup"] = (256, 2, 1)
    d["grid"] = [(256 * 10, 2 * 3, 1), (256 * 10 - 255, 6, 1), (256 * 11, 6, 1)]  # 10, 10 (rounded up), 11 workgroups in x
    d["start_ns"] = 1000
    d["end_ns"] = [3000, 4500, 7000]
    ktrace.feed(d)
    got = prof.get_stats()
    assert set(got) == {"_Z4gemmPKfS0_Pf_blk_256_2_1_grid_10_3_1", "_Z4gemmPKfS0_Pf_blk_256_2_1_grid_11_3_1"}
    g = got["_Z4gemmPKfS0_Pf_blk_256_2_1_grid_10_3_1"]
    assert (g.num_calls, g.min, g.max, g.median) == (2, 2.0, 3.5, 2.75)  # microseconds; mean of the two middles
    # a kernel nobody named (its code object was loaded before the tool's name context ran)
    (anon,) = _fresh_kernel_ids(1).tolist()
    d2 = d[:1].copy()
    d2["kernel_id"] = anon
    ktrace.feed(d2)
    assert "unknown_kernel_blk_256_2_1_grid_10_3_1" in prof.get_stats()


def test_engine_kernels_are_left_out_and_full_rings_drop_new_keys_not_old_ones(profiler):
    prof, rings = profiler(3, 8)
    ids = _fresh_kernel_ids(5).tolist()
    for i, n in zip(ids, ("a", "b", "c", "d", "k_scatter")):
        ktrace.feed_kernel_name(i, n, own=(n == "k_scatter"))
    d = np.zeros(5, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"] = [ids[4], ids[0], ids[1], ids[2], ids[0]]
    d["workgroup"], d["grid"] = (64, 1, 1), (64, 1, 1)
    d["start_ns"], d["end_ns"] = 1000, 2000
    before = ktrace.counters()
    ktrace.feed(d)
    assert prof.harvest() == 0
    after = ktrace.counters()
    assert after["own_skipped"] - before["own_skipped"] == 1
    assert {k.split("_blk_")[0] for k in prof.get_stats()} == {"a", "b", "c"}
    # a fourth key finds no row: one warning, its durations are counted as lost, the others keep recording
    d4 = d[:2].copy()
    d4["kernel_id"] = [ids[3], ids[0]]
    ktrace.feed(d4)
    with pytest.warns(UserWarning, match="rings are full"):
        prof.harvest()
    assert prof.keys_without_row == 1 and prof.dropped >= 1
    stats = prof.get_stats()
    assert {k.split("_blk_")[0] for k in stats} == {"a", "b", "c"}
    assert stats["a_blk_64_1_1_grid_1_1_1"].num_calls == 3
    prof.reset()
    assert prof.get_stats() == {}


def test_runtime_memset_and_memcpy_kernels_are_not_keys_as_with_cupti(profiler):
    """CUPTI_ACTIVITY_KIND_CONCURRENT_KERNEL is the only record kind the reference accepts (CuptiProfiler.cpp:118,179): a
    memset or memcpy never becomes a key -- its own test expects ONE key after "fill + matmul"
    (tests/straggler/unit/test_cupti_ext.py:22-36).  ROCm's hipMemset / device-to-device hipMemcpy are ROCclr blit KERNELS
    and arrive as kernel dispatches: left out by name prefix, counted, and recorded on request."""
    prof, rings = profiler(8, 16)
    ids = _fresh_kernel_ids(4).tolist()
    names = ("__amd_rocclr_fillBufferAligned", "__amd_rocclr_copyBuffer", "Cijk_Ailk_Bljk_gemm", "my__amd_rocclr_lookalike")
    for i, n in zip(ids, names):
        ktrace.feed_kernel_name(i, n)
    d = np.zeros(6, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"] = [ids[0], ids[2], ids[1], ids[0], ids[3], ids[2]]
    d["workgroup"], d["grid"], d["start_ns"], d["end_ns"] = (256, 1, 1), (512, 1, 1), 1000, 3000
    before = ktrace.counters()
    ktrace.feed(d)
    assert prof.harvest() == 0                       # the blit records count as ARRIVED: nobody waits for them
    after = ktrace.counters()
    assert after["blit_skipped"] - before["blit_skipped"] == 3 and after["own_skipped"] == before["own_skipped"]
    assert after["arrived"] - before["arrived"] == 6 and after["delivered"] - before["delivered"] == 3
    got = prof.get_stats()
    assert set(got) == {"Cijk_Ailk_Bljk_gemm_blk_256_1_1_grid_2_1_1", "my__amd_rocclr_lookalike_blk_256_1_1_grid_2_1_1"}
    assert got["Cijk_Ailk_Bljk_gemm_blk_256_1_1_grid_2_1_1"].num_calls == 2
    # on request they are kernels like any other (same geometry: the cached decision is forgotten) ...
    lib = ktrace.load()
    assert lib.nvrx_ktrace_include_blits(1) == 0
    try:
        ktrace.feed(d[:1])
        assert "__amd_rocclr_fillBufferAligned_blk_256_1_1_grid_2_1_1" in prof.get_stats()
        assert ktrace.counters()["blit_skipped"] == after["blit_skipped"]
    finally:
        assert lib.nvrx_ktrace_include_blits(0) == 0
    # ... and left out again afterwards
    ktrace.feed(d[:1])
    assert prof.get_stats()["__amd_rocclr_fillBufferAligned_blk_256_1_1_grid_2_1_1"].num_calls == 1
    assert ktrace.counters()["blit_skipped"] == after["blit_skipped"] + 1


def test_a_forgiven_dispatch_whose_record_arrives_after_all_is_not_forgiven_twice(profiler):
    """``nvrx_ktrace_forgive`` stops waiting for a dispatch; if its record then arrives, arrived + forgiven would exceed
    enqueued for good and every later sync would under-wait by one.  The forgiveness is taken back."""
    prof, rings = profiler(4, 8)
    lib = ktrace.load()
    assert lib.nvrx_ktrace_sync(5.0) == 0
    (kid,) = _fresh_kernel_ids(1).tolist()
    ktrace.feed_kernel_name(kid, "straggling_record")
    d = np.zeros(1, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"], d["workgroup"], d["grid"], d["start_ns"], d["end_ns"] = kid, (1, 1, 1), (1, 1, 1), 5, 1005
    c0 = ktrace.counters()
    if c0["arrived"] + c0["forgiven"] > c0["enqueued"]:
        pytest.skip("earlier tests of this process fed uncounted records: the counters cannot show the invariant")
    assert lib.nvrx_ktrace_feed(None, 1, 1) == 0     # enqueued, record outstanding
    assert lib.nvrx_ktrace_sync(0.0) == 1
    assert lib.nvrx_ktrace_forgive() == 1 and lib.nvrx_ktrace_sync(0.0) == 0
    ktrace.feed(d, counted=False)                    # ... and there it is after all
    c1 = ktrace.counters()
    assert c1["forgiven"] == c0["forgiven"], (c0, c1)
    assert lib.nvrx_ktrace_feed(None, 1, 1) == 0     # the NEXT outstanding dispatch is waited for again
    assert lib.nvrx_ktrace_sync(0.0) == 1
    ktrace.feed(d, counted=False)
    assert lib.nvrx_ktrace_sync(0.0) == 0


def test_harvest_waits_for_dispatches_that_are_still_running(profiler, monkeypatch):
    """``nvrx_ktrace_sync``: a dispatch that was enqueued (counted) and whose record has not arrived is MISSING; harvest
    (wait=False) says so without blocking, harvest(wait=True) returns once another thread delivers the record."""
    prof, rings = profiler(4, 8)
    (kid,) = _fresh_kernel_ids(1).tolist()
    ktrace.feed_kernel_name(kid, "late")
    d = np.zeros(2, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"], d["workgroup"], d["grid"], d["start_ns"], d["end_ns"] = kid, (1, 1, 1), (1, 1, 1), 5, 2005
    lib = ktrace.load()
    assert lib.nvrx_ktrace_feed(None, 2, 1) == 0      # two kernels enqueued, none finished
    assert prof.harvest(wait=False) == 2
    assert prof.get_stats.__self__ is prof and rings.count(0) == 0
    timer = threading.Timer(0.15, lambda: ktrace.feed(d, counted=False))
    timer.start()
    prof.sync_patience_s = 5.0
    assert prof.harvest(wait=True) == 0               # blocks ~0.15 s in C, GIL released
    timer.join()
    assert prof.get_stats()["late_blk_1_1_1_grid_1_1_1"].num_calls == 2
    # a dispatch whose record never comes: after the patience runs out the device is synchronised (stubbed here), the
    # SDK flushed, and the dispatch is forgiven with one warning -- the next report does not wait for it again
    import torch

    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(ktrace.load(), "nvrx_ktrace_flush", lambda: 0, raising=False)
    prof.sync_patience_s = 0.05
    assert lib.nvrx_ktrace_feed(None, 1, 1) == 0
    assert prof.harvest(wait=True) == 1
    assert ktrace.counters()["forgiven"] >= 1
    assert prof.harvest(wait=True) == 0


def test_records_left_in_the_inbox_by_the_completion_callback_reach_the_rings_at_the_next_wait(profiler):
    """Callback delivery (the default on a GPU): the SDK's completion handler only appends a finished dispatch to the
    tracer's inbox; a look (``harvest(wait=False)``) leaves it there and reports the dispatches as missing, a wait
    (``harvest(wait=True)`` -> ``nvrx_ktrace_sync``) drains it on the calling thread -- in arrival order, per key the newest
    ``cap`` survive -- and so does ``nvrx_ktrace_flush``."""
    cap = 8
    prof, rings = profiler(4, cap)
    ids = _fresh_kernel_ids(2).tolist()
    for i, n in zip(ids, ("inbox_a", "inbox_b")):
        ktrace.feed_kernel_name(i, n)
    d = np.zeros(30, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"] = [ids[i % 2] for i in range(30)]
    d["workgroup"], d["grid"], d["start_ns"] = (64, 1, 1), (128, 1, 1), 1000
    d["end_ns"] = 1000 + 1000 * np.arange(1, 31, dtype=np.uint64)
    before = ktrace.counters()
    ktrace.feed(d[:20], counted=True, through_inbox=True)
    assert prof.harvest(wait=False) == 20                       # enqueued, completed, nobody has brought them in
    assert ktrace.counters()["arrived"] == before["arrived"]
    assert prof.harvest(wait=True) == 0
    got = prof.get_stats()
    assert got["inbox_a_blk_64_1_1_grid_2_1_1"].num_calls == cap and got["inbox_b_blk_64_1_1_grid_2_1_1"].num_calls == cap
    row = rings.kernel_row_names["inbox_b_blk_64_1_1_grid_2_1_1"]
    assert sorted(rings.samples[row].tolist()) == [float(v) for v in range(6, 21, 2)]     # the NEWEST eight of b's ten: 6 .. 20 us
    # the flush entry (the reference's cuptiActivityFlushAll) drains as well
    ktrace.feed(d[20:], counted=True, through_inbox=True)
    assert ktrace.load().nvrx_ktrace_flush() == 0
    assert prof.harvest(wait=False) == 0
    assert ktrace.counters()["arrived"] - before["arrived"] == 30


def test_without_a_sink_the_pending_queue_keeps_the_newest(monkeypatch):
    """No live profiler: records wait in the bounded queue ``nvrx_ktrace_drain`` pops; beyond ``max_pending`` the OLDEST
    are dropped (round 4 dropped the newest -- the tail where a developing straggler shows)."""
    lib = ktrace.load()
    lib.nvrx_ktrace_set_sink(None)
    lib.nvrx_ktrace_reset()
    assert lib.nvrx_ktrace_set_max_pending(16) == 0
    (kid,) = _fresh_kernel_ids(1).tolist()
    ktrace.feed_kernel_name(kid, "queued")
    d = np.zeros(40, dtype=ktrace.DISPATCH_DTYPE)
    d["kernel_id"], d["workgroup"], d["grid"], d["start_ns"] = kid, (1, 1, 1), (1, 1, 1), 1000
    d["end_ns"] = 1000 + 1000 * np.arange(1, 41, dtype=np.uint64)
    dropped0 = int(lib.nvrx_ktrace_dropped())
    ktrace.feed(d)
    assert lib.nvrx_ktrace_pending() == 16 and int(lib.nvrx_ktrace_dropped()) - dropped0 == 24
    buf = (ktrace.Record * 64)()
    n = lib.nvrx_ktrace_drain(buf, 64)
    assert n == 16 and [buf[i].us for i in range(n)] == [float(v) for v in range(25, 41)]
    assert ktrace.key_name(buf[0].key) == "queued_blk_1_1_1_grid_1_1_1"
    lib.nvrx_ktrace_set_max_pending(0)


def test_random_launch_sequences_match_the_reference_profiler(profiler):
    """Property check of the whole native path against the reference's own profiler: random key counts, ring capacities,
    launches per key (from none to > 4 x cap), launch geometries and batch sizes -- the statistics per key are the
    reference's, the newest ``cap`` durations are what the rings hold."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(2024)
    for case in range(25):
        K = int(rng.integers(1, 9))
        cap = int(rng.integers(1, 24))
        names = [f"_Z5case{case:02d}k{k}Pf" for k in range(K)]
        ids = _fresh_kernel_ids(K)
        for i, n in zip(ids.tolist(), names):
            ktrace.feed_kernel_name(i, n)
        counts = rng.integers(0, 4 * cap + 3, K)
        idx = rng.permutation(np.repeat(np.arange(K), counts))
        if idx.size == 0:
            continue
        dur = rng.integers(500, 9_000_000, idx.size).astype(np.uint64)
        start = np.cumsum(dur) + np.uint64(77)
        blocks = [(int(rng.choice([1, 32, 64, 256, 1024])), int(rng.integers(1, 3)), 1, int(rng.integers(1, 5000)), int(rng.integers(1, 4)), 1)
                  for _ in range(K)]
        disp = _as_dispatches(ids, idx, dur, start, blocks)
        prof, rings = profiler(K + 1, cap)
        lo = 0
        while lo < disp.size:
            n = int(rng.integers(1, 40))
            ktrace.feed(disp[lo:lo + n])
            lo += n
        got = prof.get_stats()
        exp = _reference_stats(names, idx, dur, start, blocks, cap)
        assert set(got) == set(exp), (case, sorted(got), sorted(exp))
        for key, (e, n) in exp.items():
            g = got[key]
            assert g.num_calls == n, (case, key, g.num_calls, n)
            assert (np.float32(g.min), np.float32(g.max), np.float32(g.median)) == (e[0], e[1], e[2]), (case, key)
            assert abs(g.avg - e[3]) <= 2e-6 * abs(e[3]) + 1e-12, (case, key)
            assert abs(g.stddev - e[4]) <= 2e-5 * max(abs(e[4]), 1e-3 * abs(e[3])), (case, key)
        prof.shutdown()
        prof.close()
        backend.set_backend(None)


@pytest.mark.parametrize("asynchronous", [False, True])
def test_detector_in_per_kernel_mode_conserves_every_duration_under_bursty_delivery(monkeypatch, asynchronous):
    """The Detector's per-kernel flow on CPU, tracer emulated at the ABI: a section "launches kernels" by counting
    dispatches as enqueued (``nvrx_ktrace_feed(NULL, n, counted)``); a feeder thread delivers their records later, in
    BURSTS of hundreds (what a lazily flushed SDK buffer looks like), while the training thread enters sections and
    reports -- synchronously (each report waits for its window) or asynchronously (reports do not wait, late durations are
    held on the tracer's thread across "statistics launch + ring reset", a report that meets a new kernel name runs on the
    old tables and KEEPS that row for the next one).  New kernel keys keep appearing.  With rings deep enough not to
    overflow, every duration must be reported exactly once: sum of NUM over all reports == records delivered, per key."""
    import queue
    import time

    from nvrx_straggler import Detector, Statistic
    from nvrx_straggler.straggler import CustomSection

    monkeypatch.setattr(ktrace, "_mode", "kernels")
    monkeypatch.setattr(ktrace, "_mode_note", "forced by the test")
    monkeypatch.setattr(ktrace, "_setup_error", None)
    monkeypatch.setattr(ktrace, "setup", lambda *a, **k: None)
    monkeypatch.setattr(ktrace.KernelTraceProfiler, "_live", None)
    monkeypatch.setattr(ktrace.KernelTraceProfiler, "_ensure_ready", lambda self: None)
    monkeypatch.setattr(ktrace.KernelTraceProfiler, "start", lambda self, key="": setattr(self, "_started", True))
    monkeypatch.setattr(ktrace.KernelTraceProfiler, "stop", lambda self, *a: (setattr(self, "_started", False), False)[1])
    monkeypatch.setenv("NVRX_DEBUG_KTRACE_SYNC_PATIENCE_S", "20")
    monkeypatch.setattr(CustomSection, "max_elapseds_len", 4096)
    backend.set_backend(OracleBackend(emulate_fused=True))
    lib = ktrace.load()
    rng = np.random.default_rng(9)
    ids = _fresh_kernel_ids(12)
    for i, kid in enumerate(ids.tolist()):
        ktrace.feed_kernel_name(kid, f"burst_kernel_{i:02d}")
    inbox: "queue.Queue" = queue.Queue()
    stop = []
    delivered = {}

    def tracer_thread():                         # delivers in bursts: waits until a few hundred records are due
        backlog = []
        while not (stop and inbox.empty() and not backlog):
            try:
                backlog.append(inbox.get(timeout=0.002))
            except queue.Empty:
                pass
            n = sum(b.size for b in backlog)
            if n >= 300 or (stop and backlog) or (backlog and rng.random() < 0.02):
                ktrace.feed(np.concatenate(backlog), counted=False)
                backlog = []

    th = threading.Thread(target=tracer_thread)
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0", asynchronous=asynchronous, max_rows=64)
    try:
        th.start()
        reported = {}
        reports = []
        for step in range(400):
            live = 3 + step // 40                # a new kernel key every 40 steps
            with Detector.detection_section("train_step", profile_cuda=True):
                k = rng.integers(0, min(live, 12), int(rng.integers(1, 9)))
                d = np.zeros(k.size, dtype=ktrace.DISPATCH_DTYPE)
                d["kernel_id"] = ids[k]
                d["workgroup"], d["grid"], d["start_ns"] = (64, 1, 1), (640, 1, 1), 1000
                d["end_ns"] = 1000 + rng.integers(1_000, 90_000, k.size).astype(np.uint64)
                assert lib.nvrx_ktrace_feed(None, int(k.size), 1) == 0       # "launched": counted, not finished
                inbox.put(d)                                                 # the tracer's thread will see them finish
                for kk in k.tolist():
                    delivered[kk] = delivered.get(kk, 0) + 1
            if step % 17 == 16:
                reports.append(Detector.generate_report())
        stop.append(1)
        th.join()
        assert lib.nvrx_ktrace_sync(10.0) == 0
        reports.append(Detector.generate_report())
        reports.append(Detector.generate_report())   # (asynchronous: a report that met a new name ran on the old tables; the next one has it)
        for rep in reports:
            for key, v in rep.local_kernel_summaries.items():
                assert v[Statistic.MIN] <= v[Statistic.MED] <= v[Statistic.MAX], (key, v)
                reported[key] = reported.get(key, 0) + int(v[Statistic.NUM])
        want = {f"burst_kernel_{kk:02d}_blk_64_1_1_grid_10_1_1": n for kk, n in delivered.items()}
        assert reported == want, {k: (reported.get(k), want.get(k)) for k in set(reported) | set(want) if reported.get(k) != want.get(k)}
        assert sum(int(r.local_section_summaries["train_step"][Statistic.NUM]) for r in reports if "train_step" in r.local_section_summaries) == 400
    finally:
        stop.append(1)
        if th.is_alive():
            th.join()
        Detector.shutdown()
        backend.set_backend(None)
        ktrace._reset_mode_for_tests()
