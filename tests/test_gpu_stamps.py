"""Device-timestamp region timing (k_stamp_begin / k_stamp_end) through the C ABI: the elapsed GPU time
lands in the ring without any host wait, agrees with a hipEvent pair around the same region, carries the
section's CPU sample along, and is ordered before the report that reads it."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _spin(ms: float):
    torch.cuda._sleep(int(ms * 1e-3 * 2.0e9))  # ~cycles; only the order of magnitude matters


def test_stamp_pair_matches_event_pair_and_appends_cpu_sample():
    from nvrx_straggler import _native
    from nvrx_straggler.backend import get_backend

    be = get_backend()
    rings = be.make_rings(1, 8, 64)
    try:
        gpu_row = rings.row_for(_native.KIND_KERNEL, "region")
        cpu_row = rings.row_for(_native.KIND_SECTION, "section")
        st = be.current_stream_handle()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _spin(0.1)  # first use of the sleep kernel loads its code object
        torch.cuda.synchronize()
        want_us = []
        for i in range(5):
            ev0.record()
            rings.stamp_begin(gpu_row, st)
            _spin(0.2 * (i + 1))
            rings.stamp_end(gpu_row, st, cpu_row, 1.5 + i)
            ev1.record()
            ev1.synchronize()
            want_us.append(ev0.elapsed_time(ev1) * 1e3)
        assert rings.count(gpu_row) == 5 and rings.count(cpu_row) == 5
        got = rings.read_row(gpu_row)[:5]  # flush orders the read after the stamp kernels
        cpu = rings.read_row(cpu_row)[:5]
        assert np.array_equal(cpu, np.array([1.5, 2.5, 3.5, 4.5, 5.5], dtype=np.float32))
        # the stamp pair sits INSIDE the event pair: never longer, and close to it
        for g, w in zip(got, want_us):
            assert 0.0 < g <= w + 5.0, (got, want_us)
            assert g >= 0.6 * w - 20.0, (got, want_us)
    finally:
        rings.close()


def test_report_is_ordered_after_stamps_without_a_host_wait():
    """generate_report right after a long GPU region: the statistics must see that region's time."""
    from nvrx_straggler import Detector, Statistic

    Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=True, node_name="n0")
    try:
        for _ in range(3):
            with Detector.detection_section("step", profile_cuda=True):
                _spin(2.0)
            t0 = time.perf_counter()
            rep = Detector.generate_report()
            dt = time.perf_counter() - t0
            key = next(k for k in rep.local_kernel_summaries if k.endswith("step"))
            gpu_us = rep.local_kernel_summaries[key][Statistic.MED]
            assert rep.local_kernel_summaries[key][Statistic.NUM] == 1
            assert gpu_us > 200.0, gpu_us  # the region really ran ~ms before the report read its ring
            assert rep.local_section_summaries["step"][Statistic.NUM] == 1
            assert dt > 0.0
    finally:
        Detector.shutdown()


def test_nested_regions_and_unmatched_end():
    from nvrx_straggler import _native
    from nvrx_straggler.backend import get_backend

    be = get_backend()
    rings = be.make_rings(1, 4, 16)
    try:
        a = rings.row_for(_native.KIND_KERNEL, "a")
        b = rings.row_for(_native.KIND_KERNEL, "b")
        st = be.current_stream_handle()
        rings.stamp_begin(a, st)
        rings.stamp_begin(b, st)
        _spin(0.1)
        rings.stamp_end(b, st)
        _spin(0.1)
        rings.stamp_end(a, st)
        va, vb = rings.read_row(a)[0], rings.read_row(b)[0]
        assert va > vb > 0.0
        with pytest.raises(_native.NativeError):
            rings.stamp_end(a, st)
    finally:
        rings.close()


@pytest.mark.parametrize("rehome", ["1", "0"])
def test_synchronous_report_follows_the_regions_it_reads_without_a_host_wait(monkeypatch, rehome):
    """A synchronous ``Detector.generate_report()`` issued while the GPU is still busy with the very region it has to
    report: the report must contain THIS window's device-stamped sample (count and magnitude), for the re-homed route
    (the report's kernels enqueued on the stamps' own stream, no events: ``NVRX_DEBUG_REPORT_REHOME=1``, the default once the
    context's own stream is known to be idle) and for the event-ordered route (``=0``).  Alternating regions of very
    different length make a report that ran ahead of its stamp read the PREVIOUS window's value or none at all."""
    from nvrx_straggler import Detector, Statistic

    monkeypatch.setenv("NVRX_DEBUG_REPORT_REHOME", rehome)
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n")
    try:
        _spin(0.1)
        torch.cuda.synchronize()
        for i in range(12):
            ms = 3.0 if i % 2 else 0.6
            with Detector.detection_section("cpu_only", profile_cuda=False):
                pass
            with Detector.detection_section("region", profile_cuda=True):
                _spin(ms)
            rep = Detector.generate_report()          # the host is ~ms ahead of the GPU here
            got = rep.local_kernel_summaries["hipevent::region"]
            assert got[Statistic.NUM] == 1, (i, got)
            assert 0.5 * ms * 1e3 < got[Statistic.MED] < 3.0 * ms * 1e3, (i, ms, got)   # microseconds
            assert rep.local_section_summaries["cpu_only"][Statistic.NUM] == 1
            assert rep.local_section_summaries["region"][Statistic.NUM] == 1
            assert abs(rep.gpu_relative_perf_scores[0] - 1.0) < 1e-6
    finally:
        Detector.shutdown()


def test_a_long_open_region_keeps_its_begin_timestamp_while_other_contexts_open_many_regions():
    """The argument-free stamp slots are the DEVICE's (64 of them, shared by every context of the process).  A region that
    stays open while 200 other regions -- in two other contexts -- open and close must keep its begin timestamp: a slot is
    not handed out again while the region that holds it is open anywhere in the process (it used to be reused
    after 64 further begins, and the long region then reported a far too small GPU time).  And when ALL 64 device slots
    are held by open regions, further regions fall back to slots of their own context and still measure."""
    from nvrx_straggler import _native
    from nvrx_straggler.backend import get_backend

    be = get_backend()
    a, b, c = be.make_rings(1, 8, 256), be.make_rings(1, 80, 256), be.make_rings(1, 8, 256)
    try:
        st = be.current_stream_handle()
        long_row = a.row_for(_native.KIND_KERNEL, "long")
        rows_b = [b.row_for(_native.KIND_KERNEL, f"r{i}") for i in range(70)]
        row_c = c.row_for(_native.KIND_KERNEL, "short")
        _spin(0.1)
        torch.cuda.synchronize()
        assert a.stamp_begin(long_row, st)
        for i in range(100):                       # 200 region entries in other contexts while "long" is open
            b.stamp_begin(rows_b[0], st)
            _spin(0.01)
            b.stamp_end(rows_b[0], st)
            c.stamp_begin(row_c, st)
            c.stamp_end(row_c, st)
        assert a.stamp_end(long_row, st)
        long_us = float(a.read_row(long_row)[0])
        short = b.read_row(rows_b[0])[:100]
        assert long_us >= float(short.sum()) > 0.0, (long_us, float(short.sum()))      # it spans every one of them
        # 30 regions open at once in b (the per-context limit is 32) + 30 in c + 10 in a: more than 64 in the process
        held = []
        for rings, n, names in ((b, 30, rows_b), (c, 30, None), (a, 10, None)):
            rows = names[1:1 + n] if names else [rings.row_for(_native.KIND_KERNEL, f"x{i}") for i in range(n)] if n <= 6 else None
            if rows is None:                       # (a and c have 8 rows: nest on the rows they have, LIFO per row)
                rows = [rings.row_for(_native.KIND_KERNEL, f"x{i}") for i in range(6)] * 5
                rows = rows[:n]
            for r in rows:
                assert rings.stamp_begin(r, st)
                held.append((rings, r))
        _spin(0.5)
        for rings, r in reversed(held):
            assert rings.stamp_end(r, st)
        torch.cuda.synchronize()
        for rings, r in {(id(x), y): (x, y) for x, y in held}.values():
            n = rings.count(r)
            vals = rings.read_row(r)[:n]
            assert n >= 1 and (vals > 100.0).all(), (n, vals)   # every one of the 70 regions spans the 0.5 ms spin
    finally:
        for r in (a, b, c):
            r.close()
