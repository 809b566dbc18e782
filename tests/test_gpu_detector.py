"""End-to-end GPU parity: Detector / FoldedJob on the MI355X vs outputs of the real reference."""
import time

import numpy as np
import pytest
import torch

import synth
from oracle import oracle
from util import close, load_golden

pytestmark = pytest.mark.gpu


def _check_report_vs_golden(rep, exp, S_names, rel=1e-4):
    W = 8
    for r in range(W):
        assert np.isnan(rep.gpu_relative_perf_scores[r]) and np.isnan(rep.gpu_individual_perf_scores[r])
    for n in S_names:
        for r in range(W):
            assert close(rep.section_relative_perf_scores[n][r], exp["section_relative_perf_scores"][n][str(r)], rel=rel)
            assert close(rep.section_individual_perf_scores[n][r], exp["section_individual_perf_scores"][n][str(r)], rel=rel)
    for thr in (0.75, 0.9):
        got = rep.identify_stragglers(thr, thr, thr, thr)
        e = exp["stragglers"][str(thr)]
        assert sorted(x.rank for x in got["straggler_gpus_relative"]) == e["straggler_gpus_relative"]
        assert sorted(x.rank for x in got["straggler_gpus_individual"]) == e["straggler_gpus_individual"]
        assert {k: sorted(x.rank for x in v) for k, v in got["straggler_sections_relative"].items()} == e["straggler_sections_relative"]
        assert {k: sorted(x.rank for x in v) for k, v in got["straggler_sections_individual"].items()} == e["straggler_sections_individual"]


def test_stress_8ranks_64sections_10k_vs_reference_golden():
    """BASELINE configs #3 and #5 folded onto one GPU (8 logical ranks): scores within 1e-4 of the
    reference Detector's, flagged sets identical at thresholds 0.75 and 0.9, local summaries equal.
    Ring capacity 8192 on both sides (the reference keeps the newest 8192 of 10 000)."""
    from nvrx_straggler import Statistic
    from nvrx_straggler.folded import FoldedJob

    g = load_golden("stress.json")
    names = [synth.section_name(s) for s in range(64)]
    job = FoldedJob(total_ranks=8, section_names=names, ring_cap=8192, node_name="node0")
    try:
        for var in g["variants"]:
            for r in range(8):
                job.load(r, synth.stress_samples(r, var["S"], var["n"], var["slow_rank"], var["slow_factor"]))
            rep = job.report()
            exp = g["rank0"][var["name"]]
            _check_report_vs_golden(rep, exp, names)
            for n in names:  # rank 0's local summaries
                e = exp["local_section_summaries"][n]
                s = rep.local_section_summaries[n]
                assert s[Statistic.MED] == np.float32(e["MED"]) and s[Statistic.NUM] == e["NUM"]
                assert s[Statistic.MIN] == np.float32(e["MIN"]) and s[Statistic.MAX] == np.float32(e["MAX"])
                assert close(s[Statistic.AVG], e["AVG"], rel=1e-6) and close(s[Statistic.STD], e["STD"], rel=1e-5)
            assert rep.rank_to_node == {r: "node0" for r in range(8)}
    finally:
        job.close()


def test_stress_capacity_10000_vs_oracle():
    """Same workload with both sides' ring capacity raised to 10 000 (the bench configuration)."""
    from nvrx_straggler.folded import FoldedJob

    names = [synth.section_name(s) for s in range(64)]
    job = FoldedJob(total_ranks=8, section_names=names, ring_cap=10_000)
    try:
        xs = [synth.stress_samples(r, 64, 10_000, 3, 1.5) for r in range(8)]
        for r in range(8):
            job.load(r, xs[r])
        rep = job.report()
        T = np.zeros((8, oracle.table_len(0, 64)), dtype=np.float32)
        for r in range(8):
            st = oracle.rows_stats(xs[r], np.full(64, 10_000, dtype=np.uint32))
            T[r, :64] = st[:, 2]
            T[r, 64:128] = st[:, 2]
        exp = oracle.score_table(T, 0, 64)
        for s, n in enumerate(names):
            for r in range(8):
                assert close(rep.section_relative_perf_scores[n][r], exp[r, 2 + 64 + s], rel=1e-6)
        flagged = rep.identify_stragglers()["straggler_sections_relative"]
        assert all({x.rank for x in v} == {3} for v in flagged.values()) and len(flagged) == 64
    finally:
        job.close()


@pytest.mark.parametrize("R", [65, 96, 512])
def test_one_call_report_beyond_the_single_workgroup_scorer(R):
    """More ranks than `k_score1` takes (64): the one-call report goes through `k_colmin` + `k_score` (one workgroup
    per rank), non-resident.  R logical ranks x 8 sections folded onto one GPU, two reports (the second one's
    individual scores use the minima of both), every score against the oracle, flagged sets against the scores."""
    from nvrx_straggler import Statistic
    from nvrx_straggler.folded import FoldedJob

    S, n = 8, 301
    names = [synth.section_name(s) for s in range(S)]
    job = FoldedJob(total_ranks=R, section_names=names, ring_cap=512, node_name="n")
    try:
        hist = None
        slow = (R - 2, R // 3)
        for t in range(2):
            xs = [synth.stress_samples(r + 1000 * t, S, n, slow_rank=slow[t] + 1000 * t, slow_factor=1.5) for r in range(R)]
            for r in range(R):
                job.load(r, xs[r])
            rep = job.report()
            med = np.stack([oracle.rows_stats(xs[r], np.full(S, n, dtype=np.uint32))[:, 2] for r in range(R)])
            hist = med if hist is None else np.minimum(hist, med)
            T = np.zeros((R, oracle.table_len(0, S)), dtype=np.float32)
            T[:, :S], T[:, S : 2 * S], T[:, -1] = med, hist, 1.0
            exp = oracle.score_table(T, 0, S)
            for s, name in enumerate(names):
                rel, ind = rep.section_relative_perf_scores[name], rep.section_individual_perf_scores[name]
                assert list(rel) == list(range(R))
                for r in range(R):
                    assert close(ind[r], exp[r, 2 + s], rel=1e-6), (t, name, r)
                    assert close(rel[r], exp[r, 2 + S + s], rel=1e-6), (t, name, r)
            got = rep.identify_stragglers()
            assert {k: {x.rank for x in v} for k, v in got["straggler_sections_relative"].items()} == \
                {name: {slow[t]} for name in names}
            exp_ind = {name: {r for r in range(R) if exp[r, 2 + s] < 0.75} for s, name in enumerate(names)}
            assert {k: {x.rank for x in v} for k, v in got["straggler_sections_individual"].items()} == \
                {k: v for k, v in exp_ind.items() if v}
            assert rep.local_section_summaries[names[0]][Statistic.NUM] == n
            assert job.reporter._ring_plan.ws.meta[0] == 1
    finally:
        job.close()


def test_detector_sleep_sections_single_rank():
    """Detector API end to end on one rank: NUM honours profiling_interval, report resets the rings,
    GPU-timed section produces a hipevent:: kernel summary, scores of a lone rank are 1."""
    from nvrx_straggler import Detector, Statistic

    Detector.initialize(scores_to_compute="all", gather_on_rank0=False, profiling_interval=2, node_name="n0")
    try:
        x = torch.randn(256, 256, device="cuda")
        for _ in range(4):
            with Detector.detection_section("cpu_only", profile_cuda=False):
                time.sleep(0.002)
            with Detector.detection_section("with_gpu", profile_cuda=True):
                (x @ x).sum()
        rep = Detector.generate_report()
        assert rep.local_section_summaries["cpu_only"][Statistic.NUM] == 2
        assert rep.local_section_summaries["with_gpu"][Statistic.NUM] == 2
        assert rep.local_section_summaries["cpu_only"][Statistic.MIN] >= 2.0
        ks = rep.local_kernel_summaries
        assert list(ks.keys()) == ["hipevent::with_gpu"] and ks["hipevent::with_gpu"][Statistic.NUM] == 2
        assert ks["hipevent::with_gpu"][Statistic.MIN] > 0.0
        assert rep.section_relative_perf_scores["cpu_only"][0] == pytest.approx(1.0)
        assert rep.gpu_relative_perf_scores[0] == pytest.approx(1.0)
        assert rep.gpu_individual_perf_scores[0] == pytest.approx(1.0)
        # rings were emptied: the next report has nothing and must not crash (test_det_section_api.py:122-133)
        rep2 = Detector.generate_report()
        assert len(rep2.local_section_summaries) == 0 and len(rep2.local_kernel_summaries) == 0
        assert np.isnan(rep2.gpu_relative_perf_scores[0])
        # an exception inside a section records no sample (straggler.py:337-340)
        with pytest.raises(ValueError):
            with Detector.detection_section("raises", profile_cuda=True):
                raise ValueError("boom")
        assert len(Detector.custom_sections["raises"].cpu_elapsed_times) == 0
        with pytest.raises(AssertionError):
            Detector.initialize()
    finally:
        Detector.shutdown()
    with pytest.raises(RuntimeError, match="Detector is not initialized."):
        with Detector.detection_section("x"):
            pass


def test_profiler_module_twin_of_reference_cupti_tests():
    """hipEvent twin of tests/straggler/unit/test_cupti_ext.py + test_cupti_manager.py: one key per
    region, start/stop gating, reset, ring cap, singleton, refcounted manager."""
    import nvrx_cupti_module
    from nvrx_straggler.cupti import CuptiManager

    prof = nvrx_cupti_module.CuptiProfiler(statsMaxLenPerKernel=7)
    with pytest.raises(RuntimeError):
        nvrx_cupti_module.CuptiProfiler()
    prof.initialize()
    x = torch.randn(512, 512, device="cuda")
    (x @ x).sum().item()  # before start: not recorded
    assert prof.get_stats() == {}
    for _ in range(21):
        prof.start("matmul")
        x @ x
        prof.stop()
    st = prof.get_stats()
    assert list(st.keys()) == ["matmul"] and st["matmul"].num_calls == 7  # ring cap (test_cupti_ext.py:98-116)
    assert 0 < st["matmul"].min <= st["matmul"].median <= st["matmul"].max
    assert "num calls: 7" in str(st["matmul"])
    prof.reset()
    assert prof.get_stats() == {}
    prof.shutdown()
    prof.close()

    mgr = CuptiManager(statsMaxLenPerKernel=16)
    with pytest.raises(RuntimeError, match="not initialized"):
        mgr.start_profiling()
    mgr.initialize()
    with pytest.raises(RuntimeError, match="No active profiling run."):
        mgr.stop_profiling()
    mgr.start_profiling("outer")
    mgr.start_profiling("inner")  # nested: only the outermost pair records
    x @ x
    mgr.stop_profiling()
    mgr.stop_profiling()
    res = mgr.get_results()
    assert list(res.keys()) == ["outer"] and res["outer"].num_calls == 1
    mgr.reset_results()
    assert mgr.get_results() == {}
    mgr.shutdown()


@pytest.mark.gpu
def test_gpu_telemetry_reads_rocm_smi():
    """ROCm SMI side of the profiling backend: clocks, temperatures and power of this rank's GPU."""
    from nvrx_straggler import Detector, gpu_telemetry

    assert gpu_telemetry.num_devices() >= 1
    s = gpu_telemetry.sample(0)
    assert 100.0 <= s["sclk_peak_mhz"] <= 4000.0 and 0.0 < s["sclk_frac"] <= 1.0, s
    assert s["sclk_mhz"] <= s["sclk_peak_mhz"]
    assert 500.0 <= s["mclk_peak_mhz"] <= 4000.0, s
    temps = [v for k, v in s.items() if k.startswith("temp_")]
    assert temps and all(5.0 < t < 120.0 for t in temps), s
    if "power_w" in s:
        assert 10.0 < s["power_w"] < 2000.0, s
    line = Detector.gpu_telemetry_line()
    assert line.startswith("gpu telemetry: sclk_mhz="), line


def test_config2_loop_folded_on_one_gpu_matches_reference():
    """BASELINE config #2 folded onto one process (8 logical ranks x 4 sections x 100 samples per report, ten reports
    through one FoldedJob): device history minima across reports, ring reset by every report, cached plan from the
    second report on -- every report vs the real reference's (loop.json)."""
    import workers
    from nvrx_straggler.folded import FoldedJob
    from util import compare_reports

    g = load_golden("loop.json")
    cfg = g["config"]
    names = [synth.section_name(s) for s in range(cfg["S"])]
    job = FoldedJob(total_ranks=8, section_names=names, ring_cap=8192, node_name="node0")
    try:
        for t, exp in enumerate(g["rank0_reports"]):
            slow = cfg["slow_rank"] if t >= cfg["slow_from"] else -1
            for r in range(8):
                job.load(r, synth.loop_samples(r, t, cfg["S"], cfg["n"], slow_rank=slow, slow_factor=cfg["slow_factor"]))
            rep = job.report()
            assert job.rings.count(0) == 0
            got = workers.report_to_plain(rep, (0.75, 0.9))
            got["rank_to_node"] = exp["rank_to_node"]  # one process holds all 8 logical ranks here
            compare_reports(got, exp, ("loop-folded", t), rel=1e-4)
    finally:
        job.close()


@pytest.mark.parametrize("rehome_gap_us", [None, "1"])
def test_asynchronous_reports_do_not_race_with_device_stamps(monkeypatch, rehome_gap_us):
    """(``rehome_gap_us`` "1": every eligible asynchronous report is enqueued on the stamps' own stream, the round-4 route
    of reports at production cadence; None: they stay on the detector's stream, as reports a millisecond apart do.)
    Asynchronous report t is only enqueued when generate_report returns; the stamp kernels of window t+1 (user
    stream) write ring slots that report t's statistics kernel (detector stream) may not have read yet.  The library
    orders them on the device.  Provoked here by parking a long kernel in front of every report: each window's CPU
    samples all carry the window index, so a sample of window t+1 inside report t would show up as MAX > t."""
    from nvrx_straggler import Detector, Statistic
    from nvrx_straggler.backend import get_backend

    be = get_backend()
    if rehome_gap_us is not None:
        monkeypatch.setenv("NVRX_DEBUG_ASYNC_REHOME_GAP_US", rehome_gap_us)  # read when the rings' context is created
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", asynchronous=True)
    try:
        with Detector.detection_section("s", profile_cuda=True):
            pass
        Detector.generate_report()
        rings = Detector.rings
        sec = Detector.custom_sections["s"]
        gpu_row = rings.kernel_row_names["hipevent::s"]
        big = torch.randn(8192, 8192, device="cuda")
        user = torch.cuda.current_stream()
        prev, seen = None, 0
        for t in range(1, 40):
            for _ in range(3):
                rings.stamp_begin(gpu_row, be.current_stream_handle())
                rings.stamp_end(gpu_row, be.current_stream_handle(), sec.row, float(t))
            with torch.cuda.stream(be.stream):
                (big @ big).sum()            # ~3 ms in front of the report on the detector's stream
            rep = Detector.generate_report()  # returns at once
            if prev is not None:
                s = prev.local_section_summaries["s"]
                assert s[Statistic.NUM] == 3 and s[Statistic.MIN] == s[Statistic.MAX] == float(t - 1), (t, s)
                assert prev.local_kernel_summaries["hipevent::s"][Statistic.NUM] == 3
                seen += 1
            prev = rep
        user.synchronize()
        assert seen == 38 and prev.local_section_summaries["s"][Statistic.MAX] == 39.0
    finally:
        Detector.shutdown()


def test_asynchronous_and_synchronous_detector_reports_agree():
    from nvrx_straggler import Detector

    rng = np.random.default_rng(3)
    data = [{f"sec{i}": rng.normal(5.0 * (i + 1), 0.2, 50).astype(np.float32) for i in range(5)} for _ in range(4)]
    results = {}
    for mode in (False, True):
        Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", asynchronous=mode)
        try:
            reps = []
            for window in data:
                for name, vals in window.items():
                    with Detector.detection_section(name, profile_cuda=False):
                        pass
                    Detector.custom_sections[name].cpu_elapsed_times.clear()
                    Detector.custom_sections[name].cpu_elapsed_times.extend(vals)
                reps.append(Detector.generate_report())
            results[mode] = [(dict(r.section_individual_perf_scores), dict(r.local_section_summaries), r.identify_stragglers())
                             for r in reps]
        finally:
            Detector.shutdown()
    assert results[False] == results[True]


@pytest.mark.parametrize("asynchronous", [False, True])
def test_the_steady_state_lane_reports_exactly_what_the_general_path_reports(asynchronous):
    """``Detector._lane`` (one Python function around ``nvrx_window_report``) serves a report only when nothing it was built on
    has changed; everything else -- a section that holds no samples in one window, a section that appears later, GPU-timed
    regions coming and going -- must fall back to the general path and come out the same.  The same 36 windows run twice,
    lanes on and lanes off: every report equal (summaries bit for bit, scores, flagged sets), and with lanes on most
    reports were served by a lane."""
    from nvrx_straggler import Detector, Statistic, straggler

    rng = np.random.default_rng(17)
    windows = []
    for t in range(24):
        w = {f"sec{i}": rng.normal(4.0 * (i + 1), 0.3, 40).astype(np.float32) * (1.6 if (i == 2 and t >= 12) else 1.0) for i in range(4)}
        if t in (7, 8):
            del w["sec1"]                      # holds no samples in these windows: the occupied set changes twice
        if t >= 15:
            w["late_section"] = rng.normal(9.0, 0.1, 40).astype(np.float32)
        windows.append((w, t % 5 != 3))        # (and a GPU-timed region in most windows)
    for t in range(24, 36):                    # ... then a steady stretch: the same sections and region in every window
        w = {f"sec{i}": rng.normal(4.0 * (i + 1), 0.3, 40).astype(np.float32) * (1.6 if i == 2 else 1.0) for i in range(4)}
        w["late_section"] = rng.normal(9.0, 0.1, 40).astype(np.float32)
        windows.append((w, True))
    x = torch.randn(256, 256, device="cuda")

    def run(lanes):
        served = []
        real_run = straggler._Lane.run

        def counting_run(self, det):
            out = real_run(self, det)
            served.append(out is not straggler._MISS)
            return out

        straggler._Lane.run = counting_run
        Detector._lanes_enabled = lanes
        Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", asynchronous=asynchronous)
        try:
            reps = []
            for w, gpu in windows:
                for name, vals in w.items():
                    with Detector.detection_section(name, profile_cuda=False):
                        pass
                    Detector.custom_sections[name].cpu_elapsed_times.clear()
                    Detector.custom_sections[name].cpu_elapsed_times.extend(vals)
                if gpu:
                    for _ in range(3):
                        with Detector.detection_section("gpu_region", profile_cuda=True):
                            (x @ x).sum()
                    torch.cuda.synchronize()
                reps.append(Detector.generate_report())
            out = []
            for r in reps:
                ks = {k: {s: v for s, v in d.items() if s.name in ("NUM",)} for k, d in r.local_kernel_summaries.items()}
                synthetic = lambda d: {k: v for k, v in d.items() if k != "gpu_region"}      # (that one's wall times are real)
                out.append((synthetic(r.section_individual_perf_scores), synthetic(r.section_relative_perf_scores),
                            synthetic(r.local_section_summaries), ks, r.local_section_summaries.get("gpu_region", {}).get(Statistic.NUM),
                            synthetic(r.identify_stragglers(section_indiv_threshold=0.7)["straggler_sections_individual"]),
                            synthetic(r.identify_stragglers()["straggler_sections_relative"])))
            return out, served
        finally:
            Detector.shutdown()
            Detector._lanes_enabled = True
            straggler._Lane.run = real_run

    def plain(v):   # NaN scores (a section that holds no samples in a window) must compare equal to themselves
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple, set, frozenset)):
            return type(v)(plain(x) for x in v)
        return "nan" if isinstance(v, float) and v != v else v

    with_lanes, served = run(True)
    without, never = run(False)
    assert never == []
    for t, (a, b) in enumerate(zip(with_lanes, without)):
        assert plain(a) == plain(b), (t, a, b)
    # the occupied set changes at windows 3|4, 7|9, 8|9, 13|14, 15, 18|19, 23: each change costs a miss, one report on the general
    # path (which leaves a plan) and one on the planned path (which leaves a lane); the steady stretches are served by lanes
    assert served.count(True) >= 5 + 9, served
    assert served.count(False) >= 3, served
    assert all(served[-9:]), served                                   # the steady stretch: every report by a lane
    flagged = [bool(r[5]) for r in with_lanes]
    assert not any(flagged[:12]) and all(flagged[12:]), flagged      # sec2 slows down by 1.6 from window 12 on


def test_report_timeout_raises_and_the_next_report_recovers(monkeypatch):
    """The wait for the completion word is bounded by NVRX_REPORT_TIMEOUT_S (default: 30 minutes, c10d's): when it
    expires the report raises, the workspace whose kernels may still be queued is parked (never reused, never freed
    under them), and the next report runs on a fresh one."""
    from nvrx_straggler import _native
    from nvrx_straggler import backend as backend_mod
    from nvrx_straggler.backend import get_backend
    from nvrx_straggler.folded import FoldedJob

    be = get_backend()
    names = [synth.section_name(s) for s in range(4)]
    job = FoldedJob(total_ranks=2, section_names=names, ring_cap=256, node_name="n")
    try:
        for r in range(2):
            job.load(r, synth.loop_samples(r, 0, 4, 100))
        first = job.report(reset=False)
        job.report(reset=False)  # the cached plan's first run (initialises its exchange rows: a cold, synchronising step)
        ws_before = job.reporter._ring_plan.ws
        big = torch.randn(8192, 8192, device="cuda")
        monkeypatch.setenv("NVRX_REPORT_TIMEOUT_S", "0.02")
        backend_mod.refresh_report_timeout()
        for blk in ws_before.blocks:  # every result block caches a descriptor (and the timeout in it): make both pick the new one up
            blk.desc_key = None
        with torch.cuda.stream(be.stream):
            for _ in range(40):          # > 100 ms of work queued in front of the report
                big = (big @ big) * 1e-4
        with pytest.raises(_native.NativeError, match="not seen after|gave up waiting"):
            job.report(reset=False)
        assert ws_before in be._retired and ws_before not in be._workspaces.values()
        monkeypatch.setenv("NVRX_REPORT_TIMEOUT_S", "60")
        backend_mod.refresh_report_timeout()
        be.synchronize()
        assert job.reporter._ring_plan is None  # the generator dropped the plan that pointed at the parked workspace
        again = job.report(reset=False)
        assert job.reporter._ring_plan.ws is not ws_before
        assert again.section_relative_perf_scores == first.section_relative_perf_scores
    finally:
        job.close()
        monkeypatch.undo()
        backend_mod.refresh_report_timeout()


def test_resident_scorer_never_reads_a_stale_row(monkeypatch):
    """3000 one-call reports on one set of rings whose valid counts cycle through three values: every report's
    medians, scores and statistics must be the ones of ITS counts.  The score kernel is resident on its own stream and
    takes the rows' results from 8-byte {epoch, value} granules (two parities); a granule of an older report, a row
    read before it was published or statistics forwarded under the wrong completion word would all show up here."""
    from nvrx_straggler import Statistic
    from nvrx_straggler.folded import FoldedJob

    monkeypatch.setenv("NVRX_DEBUG_RESIDENT_SCORER", "2")  # resident whatever the library's own choice would be
    S, N, R = 6, 1000, 4
    names = [synth.section_name(s) for s in range(S)]
    job = FoldedJob(total_ranks=R, section_names=names, ring_cap=N, node_name="n")
    try:
        data = [synth.stress_samples(r, S, N, slow_rank=2, slow_factor=1.3) for r in range(R)]
        for r in range(R):
            job.load(r, data[r])
        counts = (N, 617, 333)
        exp = {}
        for n in counts:
            med = np.array([[oracle.section_stats(data[r][s, :n].astype(np.float64))[2] for s in range(S)] for r in range(R)])
            med32 = med.astype(np.float32)
            exp[n] = (med32, med32.min(axis=0).astype(np.float64)[None, :] / med32.astype(np.float64))
        held = []
        busy = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
        for i in range(3000):
            if i % 100 == 0 and i < 2000:
                for _ in range(8):   # the first 2000 reports share the GPU with a stream of matmuls (uneven load)
                    busy = torch.matmul(busy, busy) * 1e-3
            n = counts[i % 3]
            job.rearm(n)
            rep = job.report()
            if i % 7 == 0:
                held.append((n, rep))          # read later: collected by the workspace before the block is reused
                if len(held) > 5:
                    n, rep = held.pop(0)
            med32, rel = exp[n]
            for s, name in enumerate(names):
                got = [rep.section_relative_perf_scores[name][r] for r in range(R)]
                assert np.allclose(got, rel[:, s], rtol=1e-6, atol=0), (i, n, name, got, rel[:, s])
            st = rep.local_section_summaries[names[i % S]]
            assert st[Statistic.NUM] == n and st[Statistic.MED] == med32[0, i % S], (i, n, st)
    finally:
        job.close()


def test_asynchronous_reports_read_late_or_never_keep_their_own_values():
    """An asynchronous report nobody reads is not copied when the next one is enqueued (a poll of its completion word);
    its result block is written again two reports later, and a report that is STILL HELD then must have been copied out
    by the block first.  Sixty reports over rows whose medians change every report: every third one is held and read five
    reports late (its block has been rewritten twice by then), every fourth is read at once, the rest never -- every value
    read must be the one of ITS report."""
    from nvrx_straggler import Statistic
    from nvrx_straggler.folded import FoldedJob

    names = [synth.section_name(s) for s in range(4)]
    job = FoldedJob(total_ranks=2, section_names=names, ring_cap=256, node_name="n")
    job.reporter.asynchronous = True
    rng = np.random.default_rng(11)
    base = [rng.uniform(5.0, 6.0, (4, 200)).astype(np.float32) for _ in range(2)]

    def expected(t):
        x = base[0] * np.float32(1.0 + 0.25 * t)
        return [float(np.sort(x[s])[(200 - 1) // 2]) for s in range(4)]

    held = {}
    checked = 0
    try:
        for t in range(60):
            for r in range(2):
                job.load(r, base[r] * np.float32(1.0 + 0.25 * t))
            rep = job.report()
            assert rep is not None
            if t % 3 == 0:
                held[t] = rep
            elif t % 4 == 0:
                got = [rep.local_section_summaries[n][Statistic.MED] for n in names]
                assert got == expected(t), (t, got)
                checked += 1
            for u in [u for u in held if t - u >= 5]:
                late = held.pop(u)
                got = [late.local_section_summaries[n][Statistic.MED] for n in names]
                assert got == expected(u), (u, t, got, expected(u))
                # scores of a two-rank job with different data per rank: rank 0's relative score is med0/min(med0, med1) <= 1
                assert set(late.section_relative_perf_scores[names[0]]) == {0, 1}
                assert late.identify_stragglers()["straggler_gpus_relative"] == set()
                checked += 1
        assert checked >= 25
    finally:
        job.close()


def test_asynchronous_rehomed_reports_guard_ring_writers_on_other_streams(monkeypatch):
    """An asynchronous report that is enqueued ON the stream its stamps ran on (reports at production cadence, forced here
    by a 1 us gap) records no event of its own; the stamp kernels of the NEXT window, launched on ANOTHER stream, must
    still not overtake its statistics kernel -- the guard records the event lazily, behind the report, when such a writer
    turns up.  Windows alternate between two user streams, a ~3 ms kernel is parked in front of every report on the
    stream it will run on; a sample of window t+1 inside report t would show up as MAX > t."""
    from nvrx_straggler import Detector, Statistic

    monkeypatch.setenv("NVRX_DEBUG_ASYNC_REHOME_GAP_US", "1")
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", asynchronous=True)
    try:
        with Detector.detection_section("s", profile_cuda=True):
            pass
        Detector.generate_report()
        rings = Detector.rings
        sec = Detector.custom_sections["s"]
        gpu_row = rings.kernel_row_names["hipevent::s"]
        big = torch.randn(8192, 8192, device="cuda")
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        prev, seen = None, 0
        for t in range(1, 30):
            st = streams[t % 2]
            with torch.cuda.stream(st):
                (big @ big).sum()            # ~3 ms: the report of this window queues behind it on this stream
                for _ in range(3):
                    rings.stamp_begin(gpu_row, st.cuda_stream)
                    rings.stamp_end(gpu_row, st.cuda_stream, sec.row, float(t))
                rep = Detector.generate_report()  # returns at once; re-homed onto `st`
            if prev is not None:
                s = prev.local_section_summaries["s"]
                assert s[Statistic.NUM] == 3 and s[Statistic.MIN] == s[Statistic.MAX] == float(t - 1), (t, s)
                assert prev.local_kernel_summaries["hipevent::s"][Statistic.NUM] == 3
                seen += 1
            prev = rep
        torch.cuda.synchronize()
        assert seen == 28 and prev.local_section_summaries["s"][Statistic.MAX] == 29.0
    finally:
        Detector.shutdown()


def test_asynchronous_reports_without_stamps_guard_ring_writers_on_other_streams():
    """A context whose rings no stamp kernel has ever written (per-kernel timing, sections without GPU time) records no guard
    event when it enqueues an asynchronous report: every writer it knows is on its own stream.  A device-side append from
    ANOTHER stream -- part of the ABI -- must still not overtake the report's statistics kernel: the guard records the event
    lazily when such a writer turns up.  A ~3 ms kernel is parked on the detector's stream in front of every report; window
    t+1 is appended from a second / third stream right after report t was enqueued (the rings were reset: it lands in the very
    slots report t still has to read); a sample of window t+1 inside report t would show up as MAX > t."""
    from nvrx_straggler import Detector, Statistic
    from nvrx_straggler import backend as backend_mod

    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n", asynchronous=True)
    try:
        with Detector.detection_section("s", profile_cuda=False):
            pass
        Detector.generate_report()
        rings = Detector.rings
        be = backend_mod.get_backend()
        row = Detector.custom_sections["s"].row
        big = torch.randn(8192, 8192, device="cuda")
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        values = [torch.full((3,), float(t), device="cuda") for t in range(0, 31)]
        torch.cuda.synchronize()

        def append(t, st):
            rc = rings.lib.nvrx_ring_push_device(rings.ctx, row, values[t].data_ptr(), 3, st.cuda_stream)
            assert rc == 0, rc

        append(1, streams[1])
        prev, seen = None, 0
        for t in range(1, 30):
            with be.stream_context():
                (big @ big).sum()            # ~3 ms: the report of this window queues behind it on the detector's stream
            rep = Detector.generate_report()  # returns at once
            append(t + 1, streams[t % 2])     # window t+1, from another stream, into the slots report t reads
            if prev is not None:
                s = prev.local_section_summaries["s"]
                assert s[Statistic.NUM] == 3 and s[Statistic.MIN] == s[Statistic.MAX] == float(t - 1), (t, s)
                seen += 1
            prev = rep
        torch.cuda.synchronize()
        assert seen == 28 and prev.local_section_summaries["s"][Statistic.MAX] == 29.0
    finally:
        Detector.shutdown()


def test_initialize_before_the_gpu_is_selected_binds_nothing():
    """The reference's example order (examples/straggler/example.py:60-66): ``Detector.initialize()`` first, the device
    afterwards.  In a fresh interpreter: after ``initialize`` no engine exists and PyTorch's CUDA state is untouched; the
    rings appear with the first section, on the device that is current then."""
    import json
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, os, sys
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO]
import torch
from nvrx_straggler import Detector, backend
Detector.initialize(gather_on_rank0=True)
out = {"engine_after_initialize": backend._backend is not None, "cuda_initialised_after_initialize": bool(torch.cuda.is_initialized()),
       "rings_after_initialize": Detector._rings is not None}
torch.cuda.set_device(torch.cuda.device_count() - 1)      # the LAST visible device: whatever the box has
x = torch.randn(256, 256, device="cuda")
for _ in range(3):
    with Detector.detection_section("fwd", profile_cuda=True):
        y = x @ x
rep = Detector.generate_report()
out["ring_device"] = int(backend.get_backend().device.index)
out["current_device"] = int(torch.cuda.current_device())
out["keys"] = sorted(rep.local_kernel_summaries)
out["num"] = int(list(rep.local_section_summaries["fwd"].values())[-1])
Detector.shutdown()
print("RESULT " + json.dumps(out))
'''
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "NVRX_GPU_TIMING")}
    p = subprocess.run([sys.executable, "-c", f"REPO = {repo!r}\n" + code], capture_output=True, text=True, timeout=180, env=env)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["engine_after_initialize"] is False and out["rings_after_initialize"] is False
    assert out["cuda_initialised_after_initialize"] is False
    assert out["ring_device"] == out["current_device"]
    assert out["keys"] == ["hipevent::fwd"] and out["num"] == 3


@pytest.mark.parametrize("mode", ["stamp", "event"])
def test_a_section_entered_while_its_stream_is_captured_records_no_gpu_sample(monkeypatch, mode):
    """``detection_section(profile_cuda=True)`` inside ``torch.cuda.graph``: nothing runs while a stream is captured, so the
    entry gets no GPU sample (the reference's activity records hold no kernel of a capture either,
    cupti_src/CuptiProfiler.cpp:168-207) -- no timestamp lands in the graph (a replay would write one fixed ring slot, and the
    host would have counted a sample nobody wrote), the section's wall time is kept, the graph replays unharmed, and the
    sections around the replays are timed as usual."""
    from nvrx_straggler import Detector, Statistic

    monkeypatch.setenv("NVRX_GPU_TIMING", mode)
    Detector.initialize(scores_to_compute="all", gather_on_rank0=False, node_name="n0")
    try:
        x = torch.randn(512, 512, device="cuda")
        out = torch.zeros(512, 512, device="cuda")
        for _ in range(3):                                       # eager entries: rows exist, libraries are warm
            with Detector.detection_section("fwd", profile_cuda=True):
                out.copy_(x @ x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            with Detector.detection_section("fwd", profile_cuda=True):
                out.copy_(x @ x)
        assert Detector.rings.regions_skipped == 1
        out.zero_()
        for _ in range(4):
            with Detector.detection_section("replay", profile_cuda=True):
                g.replay()
        torch.cuda.synchronize()
        assert torch.allclose(out, x @ x, rtol=1e-3, atol=1e-3)   # the captured work is the user's, nothing of ours is in it
        rep = Detector.generate_report()
        assert rep.local_section_summaries["fwd"][Statistic.NUM] == 4          # 3 eager + the captured entry's wall time
        assert rep.local_section_summaries["replay"][Statistic.NUM] == 4
        ks = rep.local_kernel_summaries
        assert ks["hipevent::fwd"][Statistic.NUM] == 3                         # the captured entry has no GPU time
        assert ks["hipevent::replay"][Statistic.NUM] == 4 and ks["hipevent::replay"][Statistic.MIN] > 0.0
        assert ks["hipevent::fwd"][Statistic.MIN] > 0.0
        assert rep.gpu_relative_perf_scores[0] == pytest.approx(1.0)
        assert rep.identify_stragglers()["straggler_gpus_relative"] == set()
    finally:
        Detector.shutdown()
