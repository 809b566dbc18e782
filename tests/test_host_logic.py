"""CPU-only tests of the host side: C-ABI surface, loud failure without a GPU, name mapping, report
views, and the multi-rank protocol on gloo (world_size 2/4/8) with the oracle-backed checker backend
injected.  Expected values come from the REAL reference (tests/golden/scoring.json)."""
import ctypes
import math
import os
import pickle
import re
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import workers
from mp_util import run_ranks
from util import compare_reports, load_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------------------------------
# C ABI
# --------------------------------------------------------------------------------------------------
def test_library_exports_every_symbol_in_the_header():
    from nvrx_straggler import _native

    header = open(os.path.join(REPO, "include", "nvrx_straggler.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*|void \*)\s*(nvrx_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 40
    lib = _native.load()  # loads without a GPU; resolves every name in _native.SYMBOLS
    bound = {name for name, _, _ in _native.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.nvrx_abi_version() == _native.NVRX_ABI_VERSION == 2
    assert lib.nvrx_report_desc_size() == ctypes.sizeof(_native.ReportDesc)
    assert lib.nvrx_last_error() is not None


def test_ktrace_library_exports_every_symbol_in_its_header():
    """libnvrx_ktrace.so (per-kernel tracing, include/nvrx_ktrace.h): loads without a GPU and exports every
    declared entry point plus the rocprofiler-sdk tool hook; nothing is registered or traced here."""
    import ctypes

    from nvrx_straggler import ktrace

    header = open(os.path.join(REPO, "include", "nvrx_ktrace.h")).read()
    declared = set(re.findall(r"^(?:int|uint64_t|const char \*)\s*(nvrx_ktrace_\w+)\s*\(", header, flags=re.M))
    assert len(declared) == 25, declared
    lib = ktrace.load()
    assert declared == {name for name, _, _ in ktrace.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name)
    assert hasattr(lib, "rocprofiler_configure")  # what ROCP_TOOL_LIBRARIES / force_configure bind
    assert lib.nvrx_ktrace_ready() == 0
    assert lib.nvrx_ktrace_key_name(1 << 30) is None and lib.nvrx_ktrace_key_row(1 << 30) == -2
    assert lib.nvrx_ktrace_sync(0.0) == 0 and lib.nvrx_ktrace_forgive() == 0
    assert ctypes.sizeof(ktrace.Sink) == 32 and ctypes.sizeof(ktrace.Dispatch) == 48   # the structs of the header
    assert lib.nvrx_ktrace_start() == -1 and b"not set up" in lib.nvrx_ktrace_last_error()
    assert lib.nvrx_ktrace_drain(None, 4) == -22
    buf = (ktrace.Record * 4)()
    assert lib.nvrx_ktrace_drain(buf, 4) == 0
    assert ctypes.sizeof(ktrace.Record) == 8 and ktrace.RECORD_DTYPE.itemsize == 8


def test_kernel_trace_profiler_lifecycle_over_the_native_data_path(monkeypatch):
    """KernelTraceProfiler on CPU: start / stop nest as "subsequent calls" (CuptiProfiler.cpp:116-133), only one instance
    may live (CuptiProfiler.cpp:86-88), records reach one ring row per kernel key through the NATIVE path (fed through
    ``nvrx_ktrace_feed``; tests/test_ktrace_datapath.py pins the path itself against the reference), statistics follow
    computeStats (mean-of-middles median, population stddev), reset empties the rows."""
    from nvrx_straggler import backend, ktrace
    from oracle_backend import OracleBackend

    calls = {"start": 0, "stop": 0}
    lib = ktrace.load()
    monkeypatch.setattr(ktrace, "_setup_error", None)
    monkeypatch.setattr(ktrace, "setup", lambda *a, **k: None)
    monkeypatch.setattr(ktrace.KernelTraceProfiler, "_live", None)
    backend.set_backend(OracleBackend())
    try:
        rings = backend.get_backend().make_rings(1, 3, 16)
        prof = ktrace.KernelTraceProfiler(statsMaxLenPerKernel=16, rings=rings)
        with pytest.raises(RuntimeError, match="Only one"):
            ktrace.KernelTraceProfiler(rings=rings)
        with pytest.raises(RuntimeError):     # no HIP device here: the SDK never calls the tool's initialiser
            prof.initialize()

        class _Lib:                            # start / stop without the SDK: count the calls that reach the library
            def __getattr__(self, name):
                return getattr(lib, name)

            def nvrx_ktrace_start(self):
                calls["start"] += 1
                return 0

            def nvrx_ktrace_stop(self):
                calls["stop"] += 1
                return 0

        prof._lib = _Lib()
        prof.start("ignored")
        prof.start("ignored")  # "subsequent call": no second enable
        assert calls == {"start": 1, "stop": 0}
        assert prof.stop(5, 1.0) is False and calls["stop"] == 1
        assert prof.stop() is False and calls["stop"] == 1
        base = 1 << 50
        names = {base: "gemm", base + 1: "relu", base + 2: "ncclDevKernel"}
        for kid, n in names.items():
            ktrace.feed_kernel_name(kid, n)
        d = np.zeros(7, dtype=ktrace.DISPATCH_DTYPE)
        d["kernel_id"] = [base, base + 1, base, base + 2, base + 1, base, base]
        d["workgroup"], d["grid"], d["start_ns"] = (256, 1, 1), (256 * 64, 1, 1), 1000
        d["end_ns"] = 1000 + 1000 * np.array([10, 2, 30, 99, 4, 20, 40], dtype=np.uint64)
        ktrace.feed(d)
        stats = prof.get_stats()
        key = lambda n: f"{n}_blk_256_1_1_grid_64_1_1"  # noqa: E731
        assert set(stats) == {key("gemm"), key("relu"), key("ncclDevKernel")}
        g = stats[key("gemm")]
        assert (g.num_calls, g.min, g.max, g.median, g.avg) == (4, 10.0, 40.0, 25.0, 25.0)  # mean of the two middles
        assert abs(g.stddev - np.std([10, 30, 20, 40])) < 1e-5                            # population
        assert stats[key("relu")].median == 3.0 and stats[key("relu")].num_calls == 2
        prof.reset()
        assert prof.get_stats() == {}
        prof.shutdown()
        prof.close()
        assert lib.nvrx_ktrace_key_row(0) == -2   # the sink is gone: no key has a row any more
    finally:
        backend.set_backend(None)


def test_detection_section_api_contract():
    """Twin of the reference's tests/straggler/unit/test_det_section_api.py on the injected CPU backend: a name
    reused at another location is accepted (the reference ships with that check switched off, straggler.py:317-321;
    the checker itself is kept), default names are file:line of the `with` statement, profiling_interval
    keeps every n-th entry, profile_cuda=False records no GPU row, an entry that raises records no sample, and
    reports over empty rings do not crash."""
    import inspect

    from nvrx_straggler import Detector, Statistic, backend
    from oracle_backend import OracleBackend

    with pytest.raises(RuntimeError, match="Detector is not initialized."):
        with Detector.detection_section("section00"):
            pass
    backend.set_backend(OracleBackend())
    try:
        Detector.initialize(scores_to_compute="all", gather_on_rank0=False, profiling_interval=2, node_name="n0")
        with pytest.raises(AssertionError):
            Detector.initialize()
        with Detector.detection_section("section00", profile_cuda=False):
            pass
        with Detector.detection_section("section00", profile_cuda=False):  # same name, another line: same section
            pass
        assert Detector.custom_sections["section00"].total_entry_cnt == 2
        with pytest.raises(ValueError, match="already used"):
            Detector._ensure_section_name_is_valid("section00", "elsewhere.py:1")
        for _ in range(2):
            with Detector.detection_section(profile_cuda=False):  # default name: this line, both times
                pass
        frame = inspect.getframeinfo(inspect.currentframe())
        with Detector.detection_section(profile_cuda=False):
            pass
        defaults = [s for n, s in Detector.custom_sections.items() if n != "section00"]
        assert len(defaults) == 2 and defaults[0].name != defaults[1].name
        assert defaults[0].location.endswith(f"{frame.filename}:{frame.lineno - 2}") and defaults[0].name == defaults[0].location
        assert defaults[1].location.endswith(f"{frame.filename}:{frame.lineno + 1}")
        # periodic capture: 2 of 4 entries; an entry that raises is not recorded
        for i in range(4):
            with Detector.detection_section("one", profile_cuda=False):
                pass
        with pytest.raises(KeyError):
            with Detector.detection_section("boom", profile_cuda=False):
                raise KeyError("x")
        assert Detector.custom_sections["boom"].total_entry_cnt == 1 and len(Detector.custom_sections["boom"].cpu_elapsed_times) == 0
        rep = Detector.generate_report()
        assert rep.local_section_summaries["one"][Statistic.NUM] == 2
        assert "boom" not in rep.local_section_summaries
        assert len(rep.local_kernel_summaries) == 0  # profile_cuda=False everywhere
        assert all(len(s.cpu_elapsed_times) == 0 for s in Detector.custom_sections.values())  # report emptied the rings
        rep2 = Detector.generate_report()  # nothing recorded since: must not crash
        assert rep2 is not None and len(rep2.local_section_summaries) == 0
        with pytest.raises(RuntimeError, match="should not be instantiated"):
            Detector()
    finally:
        if Detector.initialized:
            Detector.shutdown()
        backend.set_backend(None)


def test_abi_argument_validation_without_a_gpu():
    """Pure argument checks return -EINVAL before any HIP call."""
    from nvrx_straggler import _native

    lib = _native.load()
    assert lib.nvrx_row_stats(None, None, None, 1, 6, None, None) == -22
    assert b"multiple of 4" in lib.nvrx_last_error()
    assert lib.nvrx_score(None, 0, 0, 0, 1, 1, None, None, None, None, None, 0, None, None, 0, None) == -22
    assert lib.nvrx_ctx_destroy(None) == 0
    assert lib.nvrx_ring_push(None, 0, 1.0) == -22
    with pytest.raises(_native.NativeError, match="ctx is null"):
        _native.check(lib.nvrx_ring_flush(None, None))


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_product_fails_loudly_without_gpu():
    """No silent CPU path: without an MI355X the engine refuses to start."""
    from nvrx_straggler import Detector, backend

    backend.set_backend(None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        backend.get_backend()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Detector.initialize()
    assert Detector.initialized is False
    with pytest.raises(RuntimeError, match="Detector is not initialized."):
        with Detector.detection_section("x"):
            pass
    with pytest.raises(RuntimeError, match="should not be instantiated"):
        Detector()


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_gpu_telemetry_degrades_to_a_message_without_a_gpu():
    from nvrx_straggler import Detector, gpu_telemetry

    line = gpu_telemetry.describe(0)
    assert line.startswith("gpu telemetry unavailable")
    assert Detector.gpu_telemetry_line().startswith("gpu telemetry unavailable")
    with pytest.raises(RuntimeError):
        gpu_telemetry.sample(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "nvidia-resiliency-ext_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle_backend" not in text, f


# --------------------------------------------------------------------------------------------------
# small host-side units
# --------------------------------------------------------------------------------------------------
def test_import_paths_and_exports():
    import nvrx_straggler
    from nvidia_resiliency_ext.attribution import straggler
    import nvidia_resiliency_ext.straggler as old_path

    assert straggler is nvrx_straggler and old_path is nvrx_straggler
    for name in ("Report", "StragglerId", "Statistic", "CallableId", "Detector"):
        assert hasattr(straggler, name)
    assert straggler.reporting.ReportGenerator and straggler.interval_tracker.ReportIntervalTracker
    assert straggler.cupti.CuptiManager
    assert [str(s) for s in straggler.Statistic] == ["MIN", "MAX", "MED", "AVG", "STD", "NUM"]
    assert repr(straggler.Statistic.MED) == "Statistic.MED"
    import dataclasses

    assert [f.name for f in dataclasses.fields(straggler.Report)] == [
        "gpu_relative_perf_scores", "section_relative_perf_scores", "gpu_individual_perf_scores",
        "section_individual_perf_scores", "rank_to_node", "local_section_summaries", "local_kernel_summaries",
        "generate_report_elapsed_time", "gather_on_rank0", "rank"]


def test_callable_id_naming_rules():
    import nvrx_straggler as s

    class Foo:
        def bar(self):
            pass

    assert str(s.CallableId(Foo(), "bar")) == "Foo.bar"
    assert str(s.CallableId(Foo, "bar")) == f"{Foo.__module__}.Foo.bar"
    assert str(s.CallableId(math, "sqrt")) == "math.sqrt"


def test_name_mapper_ids_are_stable_and_consecutive():
    from nvrx_straggler.name_mapper import NameMapper

    m = NameMapper()
    m.gather_and_assign_ids(kernel_names=["k1", "k0"], section_names=["s0"])
    assert m.kernel_name_to_id == {"k1": 0, "k0": 1} and m.section_name_to_id == {"s0": 0}
    m.gather_and_assign_ids(kernel_names=["k0", "k2"], section_names=["s1", "s0"])
    assert m.kernel_name_to_id == {"k1": 0, "k0": 1, "k2": 2} and m.section_name_to_id == {"s0": 0, "s1": 1}
    assert m.get_kernel_name(2) == "k2" and m.get_section_id("s1") == 1 and m.kernel_counter == 3
    v = m.version
    m.gather_and_assign_ids(kernel_names=["k0"], section_names=[])
    assert m.version == v
    pickle.loads(pickle.dumps(m))


def _device_report(scores, flags_thr=(0.75,) * 4, stats=None, section_rows=None):
    """A Report built the way the ring path builds it (lazy fields over private arrays)."""
    from nvrx_straggler.reporting import Report, _ScoreSource, _View

    S = (scores.shape[1] - 2) // 2
    ranks, names = range(scores.shape[0]), [f"s{i}" for i in range(S)]
    view = _View()
    view.S, view.ranks, view.names, view.cols = S, ranks, names, {n: i for i, n in enumerate(names)}
    view.has_rel = view.has_indiv = True
    view.section_rows, view.kernel_rows = section_rows or {}, {}
    view.layout, view.thresholds = None, tuple(float(t) for t in flags_thr)
    src = _ScoreSource(view)
    src.scores = scores
    src.stats = stats if stats is not None else np.zeros((0, 8), dtype=np.float32)
    thr = np.concatenate([[flags_thr[2], flags_thr[0]], np.full(S, flags_thr[3]), np.full(S, flags_thr[1])])
    with np.errstate(invalid="ignore"):
        src.flags = (scores.astype(np.float64) < thr).astype(np.uint8)
    return Report._from_device(src, {r: f"n{r}" for r in ranks}, 0.1, True, 0)


def test_report_is_plain_dicts_and_flag_paths_agree():
    import copy
    import dataclasses
    import json

    from nvrx_straggler.reporting import Report, StragglerId
    from nvrx_straggler.statistics import Statistic

    scores = np.array([[np.nan, 1.0, 1.0, 0.5, 1.0, 0.7], [np.nan, 0.6, 0.8, np.nan, 0.74, 0.9]], dtype=np.float32)
    stats = np.array([[1, 3, 2, 2, 1, 3, 6, 0]], dtype=np.float32)
    rep = _device_report(scores, stats=stats, section_rows={"s0": 0})
    # thresholding straight from the flag bytes builds no mapping at all
    s1 = rep.identify_stragglers()
    assert not any(k.endswith("_scores") or k.endswith("_summaries") for k in vars(rep))
    assert s1["straggler_gpus_relative"] == {StragglerId(1, "n1")}
    assert s1["straggler_gpus_individual"] == set()  # NaN never flagged
    assert s1["straggler_sections_relative"] == {"s0": {StragglerId(1, "n1")}, "s1": {StragglerId(0, "n0")}}
    assert s1["straggler_sections_individual"] == {"s1": {StragglerId(0, "n0")}}
    # other thresholds take the comparison path over the (now built) dicts and agree where they must
    assert rep.identify_stragglers(0.75, 0.75, 0.75, 0.7500001)["straggler_sections_relative"] == s1["straggler_sections_relative"]
    assert rep.identify_stragglers(0.65, 0.65, 0.65, 0.65)["straggler_sections_relative"] == {}
    # every mapping is a plain dict of plain Python values, as in the reference
    assert type(rep.gpu_relative_perf_scores) is dict and rep.gpu_relative_perf_scores == {0: 1.0, 1: float(np.float32(0.6))}
    assert type(rep.section_relative_perf_scores) is dict and type(rep.section_relative_perf_scores["s0"]) is dict
    assert list(rep.section_relative_perf_scores["s0"]) == [0, 1]
    assert all(type(v) is float for v in rep.section_relative_perf_scores["s1"].values())
    assert math.isnan(rep.section_individual_perf_scores["s1"][1])
    with pytest.raises(KeyError):
        rep.gpu_relative_perf_scores[5]
    assert rep.local_section_summaries == {"s0": {Statistic.MIN: 1.0, Statistic.MAX: 3.0, Statistic.MED: 2.0, Statistic.AVG: 2.0,
                                                  Statistic.STD: 1.0, Statistic.NUM: 3}}
    assert type(rep.local_section_summaries["s0"][Statistic.NUM]) is int and rep.local_kernel_summaries == {}
    json.dumps(rep.gpu_relative_perf_scores), json.dumps(rep.section_relative_perf_scores)
    assert set(dataclasses.asdict(rep)) == {f.name for f in dataclasses.fields(Report)}
    with pytest.raises(dataclasses.FrozenInstanceError):
        rep.rank = 3
    with pytest.raises(AttributeError):
        rep.no_such_field
    # pickle / copy BEFORE anything was read (what ret_queue.put(report) does in the reference's tests)
    fresh = _device_report(scores, stats=stats, section_rows={"s0": 0})
    for back in (pickle.loads(pickle.dumps(fresh)), copy.deepcopy(_device_report(scores)), copy.copy(_device_report(scores))):
        assert type(back) is Report and type(back.section_relative_perf_scores) is dict
        assert back.gpu_relative_perf_scores == rep.gpu_relative_perf_scores and back.identify_stragglers() == s1
        assert "_src" not in vars(back)
    assert pickle.loads(pickle.dumps(fresh)).local_section_summaries == rep.local_section_summaries
    # user-constructed reports (the reference's constructor) work too
    plain = Report({0: 0.5}, {"s": {0: 0.9}}, {}, {}, {0: "n"}, {}, {}, 0.0, False, 0)
    assert plain.identify_stragglers()["straggler_gpus_relative"] == {StragglerId(0, "n")}
    assert pickle.loads(pickle.dumps(plain)) == plain


@pytest.mark.parametrize("world", [1, 2])
def test_detector_reports_pickle_in_steady_state(world):
    """VERDICT r01 weak #1 / reference tests/straggler/unit/test_sections.py:85 (ret_queue.put(report))."""
    res = run_ranks(workers.detector_reports_pickle, world)
    assert res[0] == [0, 1, 2]


def test_cupti_manager_refcount_with_fake_native_module(monkeypatch):
    """Manager logic is backend-independent (reference: tests/straggler/unit/test_cupti_manager.py)."""
    from nvrx_straggler import backend
    from nvrx_straggler.cupti import CuptiManager
    from oracle_backend import OracleBackend

    backend.set_backend(OracleBackend())
    try:
        mgr = CuptiManager(statsMaxLenPerKernel=8)
        with pytest.raises(RuntimeError, match="CuptiManager was not initialized"):
            mgr.start_profiling()
        mgr.initialize()
        with pytest.raises(RuntimeError, match="No active profiling run."):
            mgr.stop_profiling()
        mgr.start_profiling("outer")
        mgr.start_profiling("inner")
        assert mgr.started_cnt == 2
        mgr.stop_profiling()
        mgr.stop_profiling()
        res = mgr.get_results()
        assert list(res) == ["outer"] and res["outer"].num_calls == 1
        import nvrx_cupti_module

        with pytest.raises(RuntimeError, match="Only one CuptiProfiler instance is allowed."):
            nvrx_cupti_module.CuptiProfiler()
        mgr.reset_results()
        assert mgr.get_results() == {}
        mgr.shutdown()
        nvrx_cupti_module.CuptiProfiler().close()  # slot released by shutdown
    finally:
        backend.set_backend(None)


def test_interval_tracker_single_process():
    from nvrx_straggler.interval_tracker import ReportIntervalTracker

    tr = ReportIntervalTracker(time_interval=0.1, profiling_interval=1)
    assert not tr.is_interval_elapsed()
    for i in range(18):
        tr.iter_increase()
        time.sleep(0.002 if i else 0.05)  # a warm-up outlier must not matter (median)
    assert tr.iter_interval is not None and 15 <= tr.iter_interval <= 60
    tr2 = ReportIntervalTracker(time_interval=1e-9, profiling_interval=7)
    for _ in range(18):
        tr2.iter_increase()
    assert tr2.iter_interval == 7  # floored at the profiling interval
    tr2.current_iter = 14
    assert tr2.is_interval_elapsed()


class _FakeStrategy:
    def training_step(self, batch):
        time.sleep(0.001)
        return batch


class _FakeTrainer:
    def __init__(self):
        self.strategy = _FakeStrategy()
        self.global_rank = 0
        self.should_stop = False
        self.checkpoint_callback = None


class _FakeModule:
    def __init__(self):
        self.logged = []

    def log_dict(self, d, **kw):
        self.logged.append(d)


def test_ptl_callback_with_duck_typed_trainer(caplog):
    from nvidia_resiliency_ext.ptl_resiliency import StragglerDetectionCallback
    from nvrx_straggler import Detector, backend
    from oracle_backend import OracleBackend

    with pytest.raises(ValueError, match="No straggler performance scores specified"):
        StragglerDetectionCallback(1.0, False, False, 0, 0.7, 0.7, False, False)
    backend.set_backend(OracleBackend())
    cb = StragglerDetectionCallback(report_time_interval=0.02, calc_relative_gpu_perf=True, calc_individual_gpu_perf=True,
                                    num_gpu_perf_scores_to_print=2, gpu_relative_perf_threshold=0.7,
                                    # the step is a 1 ms sleep timed on the host: on a loaded box one window's median can
                                    # be 1.5x the best one's, which a 0.7 individual threshold would call a straggler
                                    gpu_individual_perf_threshold=0.05, stop_if_detected=True, enable_ptl_logging=True,
                                    logger_name="test.straggler")
    trainer, module = _FakeTrainer(), _FakeModule()
    try:
        import logging

        caplog.set_level(logging.INFO, logger="test.straggler")
        cb.setup(trainer, module, "fit")
        assert Detector.initialized
        for i in range(100):  # several report intervals even when the box is loaded and sleeps run long
            trainer.strategy.training_step(i)
            cb.on_train_batch_end(trainer, module, None, None, i)
        assert Detector.report_interval_tracker.iter_interval is not None
        assert "Trainer" not in Detector.custom_sections and "_FakeStrategy.training_step" in Detector.custom_sections
        assert any("Straggler report processing time" in r.message for r in caplog.records)
        assert any("GPU relative performance" in r.message for r in caplog.records)
        assert module.logged and "gpu_relative_perf/median" in module.logged[0]
        assert trainer.should_stop is False
    finally:
        cb.teardown(trainer, module, "fit")
        backend.set_backend(None)
    assert not Detector.initialized
    txt = StragglerDetectionCallback._format_gpu_scores({r: 1.0 - 0.1 * r for r in range(8)}, {r: f"n{r}" for r in range(8)}, 2, 2)
    assert "Worst performing 2/8 ranks" in txt and "Rank=7 Node=n7 Score=0.30" in txt and "Best performing 2/8" in txt


# --------------------------------------------------------------------------------------------------
# multi-rank protocol on gloo, against the reference's golden outputs
# --------------------------------------------------------------------------------------------------
_SCENARIOS = load_golden("scoring.json")["scenarios"]


@pytest.mark.parametrize("idx", range(len(_SCENARIOS)), ids=[s["scenario"]["name"] for s in _SCENARIOS])
def test_report_generator_matches_reference_on_gloo_ranks(idx):
    """Our ReportGenerator (host logic + one all-gather + table scoring) replays every scenario the
    real reference was run on; scores within f32 rounding, NaN positions, key sets, rank_to_node,
    name ids and flagged sets identical."""
    g = _SCENARIOS[idx]
    sc = g["scenario"]
    res = run_ranks(workers.scoring_scenario, sc["world_size"], scenario=sc)
    for r in range(sc["world_size"]):
        for t in range(len(sc["steps"])):
            compare_reports(res[r]["reports"][t], g["per_rank"][r]["reports"][t], (sc["name"], r, t))
        assert res[r]["ids"] == g["per_rank"][r]["ids"], (sc["name"], r)


_FUZZ = load_golden("scoring_fuzz.json")["scenarios"]


@pytest.mark.parametrize("world", sorted({g["scenario"]["world_size"] for g in _FUZZ}))
def test_report_generator_matches_reference_on_random_scenarios(world):
    """The 48 random scenarios of tests/golden/scoring_fuzz.json (real reference outputs; names missing on some ranks or
    appearing mid-run, ranks without kernels, NCCL kernels to ignore, every combination of score families and
    gather_on_rank0), all scenarios of one world size in one set of gloo processes."""
    batch = [g for g in _FUZZ if g["scenario"]["world_size"] == world]
    res = run_ranks(workers.scoring_scenarios_batch, world, timeout=300, scenarios=[g["scenario"] for g in batch])
    for i, g in enumerate(batch):
        sc = g["scenario"]
        for r in range(world):
            for t in range(len(sc["steps"])):
                compare_reports(res[r][i]["reports"][t], g["per_rank"][r]["reports"][t], (sc["name"], r, t))
            assert res[r][i]["ids"] == g["per_rank"][r]["ids"], (sc["name"], r)


def test_world_and_rank_is_remembered_per_default_process_group():
    """The per-report (world, rank) lookup is cached per default process group OBJECT: tearing the group down and
    bringing a new one up (elastic restarts do) must not leave a stale answer behind."""
    import torch.distributed as dist

    from nvrx_straggler import dist_utils

    assert not dist.is_initialized() and dist_utils.world_and_rank() == (1, 0)
    for _ in range(2):
        dist.init_process_group("gloo", rank=0, world_size=1, store=dist.HashStore())
        try:
            assert dist_utils.world_and_rank() == (1, 0)
            assert dist_utils._WR_CACHE[0][0] is dist.distributed_c10d._world.default_pg
            sub = dist.new_group([0])
            assert dist_utils.world_and_rank(sub) == (1, 0) and dist_utils._WR_CACHE[0][1] is sub
            assert dist_utils.world_and_rank() == (1, 0) and dist_utils._WR_CACHE[0][1] is None
            # a caller-owned cache (one per ReportGenerator) never sees, and never disturbs, anybody else's entry
            mine, theirs = [None], [None]
            assert dist_utils.world_and_rank(sub, mine) == (1, 0) and mine[0][1] is sub
            assert dist_utils.world_and_rank(None, theirs) == (1, 0) and theirs[0][1] is None and mine[0][1] is sub
            assert dist_utils._WR_CACHE[0][1] is None
        finally:
            dist.destroy_process_group()
        assert dist_utils.world_and_rank() == (1, 0)
        assert dist_utils.world_and_rank(None, mine) == (1, 0) and mine[0] is None  # nothing of the dead group is kept


def test_all_gather_object_only_when_names_change():
    counts = run_ranks(workers.gather_object_call_counts, 2, n_kernels=256)
    assert counts[0] == [2, 0, 1, 0, 1] and counts[1] == [2, 0, 1, 0, 1]


@pytest.mark.parametrize("world", [2, 4])
def test_name_exchange_digests_give_the_ids_of_the_string_exchange(world):
    """the digest wire format assigns exactly the ids of the reference's string exchange
    (first appearance, rank-major), every rank ends up with every name, and strings only travel for digests
    some rank cannot resolve."""
    ref = run_ranks(workers.name_exchange, world, mode="strings")
    for mode in ("auto", "digests"):
        got = run_ranks(workers.name_exchange, world, mode=mode)
        for r in range(world):
            assert got[r]["kernel_ids"] == ref[0]["kernel_ids"], (mode, r)
            assert got[r]["section_ids"] == ref[0]["section_ids"]
            assert got[r]["id_to_kernel"] == ref[0]["id_to_kernel"]
            assert got[r]["counter"] == 300 + 40 * world + 52 + 1
    assert all(r["kernel_ids"] == ref[0]["kernel_ids"] for r in ref)
    assert ref[0]["calls"] == [1, 1, 1, 1]
    auto = run_ranks(workers.name_exchange, world, mode="auto")[0]["calls"]
    # SPMD bulk: digests only; private bulk: owners spell them out; mixed: the digests only ranks >= 1 hold
    # need spelling when world > 2 (with two ranks rank 1 is the only one missing nothing but rank 0 is);
    # a single late name travels as a string at once
    assert auto[0] == 1 and auto[1] == 2 and auto[2] == 2 and auto[3] == 1, auto


def test_name_digest_is_stable_and_collisions_raise(monkeypatch):
    from nvrx_straggler import name_mapper

    assert name_mapper.name_digest("kernel0") == 0xC5BDBC0080D9652  # BLAKE2b-64, little endian: same in every process
    assert name_mapper.name_digest("a") != name_mapper.name_digest("b")
    m = name_mapper.NameMapper()
    monkeypatch.setattr(name_mapper, "name_digest", lambda name: 7)
    m.sync_names(["x"], [])
    with pytest.raises(RuntimeError, match="digest collision"):
        m.sync_names(["y"], [])


def test_detector_plumbing_two_gloo_ranks_sleep_sections():
    """BASELINE config #1: 2 CPU ranks, Detector wrapping 2 time.sleep sections, relative scores on
    rank 0; the slow rank's section is flagged at the default threshold."""
    res = run_ranks(workers.detector_sleep_sections, 2, slow_rank=1, iters=20)
    assert res[1]["report"] is None
    rep = res[0]["report"]
    assert set(rep["section_relative_perf_scores"].keys()) == {"section_a", "section_b"}
    assert rep["section_relative_perf_scores"]["section_b"][1] == pytest.approx(0.5, abs=0.12)
    assert rep["section_relative_perf_scores"]["section_a"][1] > 0.8
    assert rep["section_relative_perf_scores"]["section_b"][0] == pytest.approx(1.0, abs=1e-6)
    assert rep["stragglers"]["0.75"]["straggler_sections_relative"] == {"section_b": [1]}
    assert rep["stragglers"]["0.75"]["straggler_gpus_relative"] == []
    assert rep["rank_to_node"] == {0: "host0", 1: "host1"}
    assert res[0]["n_after"] == 0 and res[0]["names"] == {0: "section_a", 1: "section_b"} == res[1]["names"]


def test_wrap_callables_two_ranks():
    res = run_ranks(workers.detector_wrap_callables, 2)
    for r in range(2):
        assert res[r]["names"] == ["Trainer.training_step"] and res[r]["num"] == 3 and res[r]["after_restore"] == 0
        assert list(res[r]["rel"].keys()) == [r]
    assert res[0]["rel"][0] == pytest.approx(1.0, abs=1e-6) and res[1]["rel"][1] == pytest.approx(0.5, abs=0.15)


def test_folded_job_over_two_gloo_ranks():
    """N>1 path of bench.py: 8 logical ranks over 2 processes, one all-gather of 4 rows per process."""
    res = run_ranks(workers.folded_job_gloo, 2, total_ranks=8, sections=4, n=200)
    assert res[1] is None
    rep = res[0]
    assert sorted(rep["section_relative_perf_scores"]["section_000"].keys()) == list(range(8))
    for name, scores in rep["section_relative_perf_scores"].items():
        assert scores[3] == pytest.approx(1 / 1.5, rel=0.02)
        assert all(scores[r] > 0.97 for r in range(8) if r != 3)
    assert all(v == [3] for v in rep["stragglers"]["0.75"]["straggler_sections_relative"].values())
    assert len(rep["stragglers"]["0.75"]["straggler_sections_relative"]) == 4
    assert rep["rank_to_node"] == {r: f"node{r // 4}" for r in range(8)}


def test_ranks_agree_on_one_gpu_timing_mode_at_their_first_collective_report():
    """A rank that could not register the kernel tracer times per region while the others time per kernel: no key is shared
    and every relative GPU score would be NaN without a word.  The ranks MIN-reduce their mode at the
    first collective report; the per-kernel rank drops to region timing ONCE and says so; both log the mode."""
    res = run_ranks(workers.detector_mode_agreement, 2)
    assert res[0]["switched"] == 1 and res[0]["mode"] == "stamp" and "another rank" in res[0]["note"]
    assert res[1]["switched"] == 0 and res[1]["mode"] == "stamp"
    assert any("measured per" in m for m in res[0]["log"]) and any("measured per" in m for m in res[1]["log"])   # said once at initialize
    assert sum("all ranks time GPU work per profiled region" in m for m in res[0]["log"]) == 1
    assert not any("all ranks time GPU work" in m for m in res[1]["log"])


def test_a_rank_that_cannot_switch_yet_defers_its_switch_and_never_repeats_the_agreement_alone():
    """``switch_to_regions()`` refused on rank 0 at the first report (a region open).  Exactly ONE
    agreement all-reduce per rank over four reports; rank 0 switches at the third report, with no collective of its own."""
    res = run_ranks(workers.detector_mode_agreement_deferred, 2)
    assert res[0]["agreements"] == 1 and res[1]["agreements"] == 1, res
    assert res[0]["refused"] == 2 and res[0]["switched"] == 1 and res[0]["mode"] == "stamp", res[0]
    assert res[0]["pending_after"] == [True, True, False, False] and res[1]["pending_after"] == [False] * 4


def test_kernel_trace_budget_thins_tracing_until_it_fits_and_the_ranks_agree():
    """``Detector.initialize(kernel_trace_budget_pct=...)``: the interval tracker's 16 timed iterations alternate between
    tracing and not tracing; rank 0's tracing costs 3 ms on a 12 ms step (25 %), rank 1's nothing.  Budget 5 %: rank 0 needs
    every 5th-6th entry, rank 1 every entry -- both adopt the LARGER interval, carried by the tracker's own all-reduce (no
    collective of the detector's).  Section wall times are still recorded for every entry."""
    res = run_ranks(workers.detector_trace_budget, 2, cost_ms_by_rank=[3.0, 0.0], budget_pct=5.0, timeout=240)
    r0, r1 = res
    # (the sleeps overshoot under load and under the sanitizer runtime -- 36 % was seen where 25 % is nominal: what is pinned is
    #  the RULE, every = ceil(measured cost / budget) of the rank that needs more, adopted by both, not the sleep's accuracy)
    assert r0["every"] == r1["every"] == min(64, math.ceil(r0["cost_pct"] / 5.0)), (r0["every"], r1["every"], r0["cost_pct"], r1["cost_pct"])
    assert 16.0 < r0["cost_pct"] < 70.0 and abs(r1["cost_pct"]) < 8.0 and 4 <= r0["every"] <= 14, (r0["cost_pct"], r1["cost_pct"])
    assert r0["iter_interval"] == r1["iter_interval"] and r0["iter_interval"] > 1000     # (3600 s / ~13 ms)
    for r in (r0, r1):
        every = r["every"]
        cal = [e for e in r["traced"] if e <= 16]
        after = [e for e in r["traced"] if e > 17]
        assert cal == [0, 1, 3, 5, 7, 9, 11, 13, 15], r["traced"]                         # iteration 0, then every other one
        assert after and all(e % every == 0 for e in after) and len(after) == len([e for e in range(18, 40) if e % every == 0]), r["traced"]
        assert r["cpu_samples"] == 40
        assert len(r["log"]) == 1 and "budget 5.00 %" in r["log"][0]
    assert "another rank needed" in r1["log"][0] and "another rank" not in r0["log"][0]


def test_kernel_trace_budget_zero_or_cheap_tracing_changes_nothing():
    """Budget 0 (the reference's behaviour: trace at ``profiling_interval`` whatever it costs) never calibrates; a budget that
    the measured cost fits leaves every profiled entry traced."""
    (off,) = run_ranks(workers.detector_trace_budget, 1, cost_ms_by_rank=[2.0], budget_pct=0.0, iters=24)
    assert off["every"] == 1 and off["cost_pct"] is None and off["traced"] == list(range(24)) and not off["log"]
    (fits,) = run_ranks(workers.detector_trace_budget, 1, cost_ms_by_rank=[0.0], budget_pct=10.0, iters=24)
    assert fits["every"] == 1 and fits["traced"][-6:] == list(range(18, 24)) and len(fits["log"]) == 1
    # a GPU-bound step whose tracing cost does not show in eight iterations each way: the dispatch count decides -- 600 traced
    # dispatches x 1 us on a 12 ms step = 5 % -> every 5th entry at a budget of 1 %
    (model,) = run_ranks(workers.detector_trace_budget, 1, cost_ms_by_rank=[0.0], budget_pct=1.0, iters=30, dispatches_per_entry=600)
    # (600 us over the MEASURED untraced step: 12 ms sleeps that overshoot on a loaded host make it 4.x % -- the rule is pinned)
    assert model["dispatches"] == 600 and 3.5 < model["cost_pct"] < 5.2 and model["every"] == math.ceil(model["cost_pct"]) in (4, 5, 6), model
    assert "600 traced dispatches x 1 us" in model["log"][0]
    # with profiling_interval=3 the multiple applies to the PROFILED entries: every (3 x N)-th entry is traced
    (thin,) = run_ranks(workers.detector_trace_budget, 1, cost_ms_by_rank=[9.0], budget_pct=10.0, iters=60, profiling_interval=3)
    n = thin["every"]
    assert n >= 2 and all(e % (3 * n) == 0 for e in thin["traced"] if e > 17) and thin["cpu_samples"] == 20, thin


def test_c10d_exchange_route_is_taken_by_every_rank_when_one_asks_for_it():
    """``NVRX_EXCHANGE=c10d`` (the report's all-gather on the JOB's own process group, no communicator of ours) is a
    collective decision: one rank's environment is enough to keep every rank on it."""
    res = run_ranks(workers.detector_c10d_route, 2)
    for r in range(2):
        assert "NVRX_EXCHANGE=c10d" in res[r]["route"] and res[r]["direct"] is False, res[r]
    assert res[0]["scores"][1] == {0: 1.0}                                      # (gather_on_rank0=False: each rank reports itself)
    assert res[1]["scores"][1][1] == pytest.approx(0.5, abs=0.15), res[1]["scores"]


def test_interval_tracker_ranks_agree():
    res = run_ranks(workers.interval_tracker_agreement, 2)
    assert res[0] == res[1] and res[0] >= 1


def test_steady_state_plan_and_name_change_on_one_rank():
    """Reports 0-2 run the cached plan; at report 3 only rank 1 has a new section, which must force both
    ranks through the name sync (rank 0 learns it from the flag word in the gathered table); later
    reports are planned again.  Scores follow the reference's rules (rank-only section -> NaN)."""
    res = run_ranks(workers.detector_name_change_midway, 2)
    assert res[0]["ids"] == res[1]["ids"] == {"a": 0, "b": 1, "late_rank1_only": 2, "late_everywhere": 3}
    assert res[0]["planned"] and res[1]["planned"]
    assert all(r is None for r in res[1]["reports"])
    reps = res[0]["reports"]
    for t in range(6):
        rel = reps[t]["section_relative_perf_scores"]
        assert rel["a"] == {0: 1.0, 1: 0.5}
        assert rel["b"] == ({0: 1.0, 1: 1.0} if t < 4 else {0: 1.0, 1: 0.5})
        if t >= 3:
            assert math.isnan(rel["late_rank1_only"][0]) and math.isnan(rel["late_rank1_only"][1])
            ind = reps[t]["section_individual_perf_scores"]["late_rank1_only"]
            assert math.isnan(ind[0]) and ind[1] == 1.0
        else:
            assert "late_rank1_only" not in rel
        if t == 5:
            assert rel["late_everywhere"] == {0: 1.0, 1: 1.0}
    # individual scores remember the best median: b doubled on rank 0 (4 -> 8) and x4 on rank 1
    assert reps[5]["section_individual_perf_scores"]["b"] == {0: 0.5, 1: 0.25}
    assert reps[5]["stragglers"]["0.75"]["straggler_sections_individual"] == {"b": [0, 1]}


def test_config2_loop_ten_reports_match_reference():
    """BASELINE config #2 (8 ranks, 4 sections, a report every 100 steps, 10 reports through one Detector) against
    the real reference's reports: every score within 1e-4 (history-driven individual scores included), identical
    flagged sets at 0.75 / 0.9, rings emptied by every report.  Host logic on the CPU checker backend; the GPU twin
    is tests/test_gpu_multiproc.py."""
    g = load_golden("loop.json")
    res = run_ranks(workers.detector_loop_config2, g["config"]["world"], timeout=300)
    assert all(rep is None for r in range(1, 8) for rep in res[r])
    for t, exp in enumerate(g["rank0_reports"]):
        compare_reports(res[0][t], exp, ("loop", t), rel=1e-4)
        for n, e in exp["local_section_summaries"].items():
            got = res[0][t]["local_section_summaries"][n]
            assert got["NUM"] == e["NUM"] == 100 and got["MED"] == np.float32(e["MED"]), (t, n)
    flagged = [res[0][t]["stragglers"]["0.9"]["straggler_sections_relative"] for t in range(10)]
    assert all(not f for f in flagged[:5]) and all(set(f) == {f"section_{s:03d}" for s in range(4)} for f in flagged[5:])


def test_asynchronous_detector_on_the_default_c10d_route_reports_like_a_synchronous_one():
    """The default exchange route is torch.distributed on the job's own group: a host-issued collective, so a multi-rank
    report is complete when ``generate_report`` returns even with ``asynchronous=True`` (said once in the log).  Same reports,
    same name ids, as the synchronous run -- the new name enters AT report 3, not one report later."""
    kw = dict(backend_kwargs={"emulate_fused": True})
    sync = run_ranks(workers.detector_async_sequence, 2, asynchronous=False, **kw)
    asyn = run_ranks(workers.detector_async_sequence, 2, asynchronous=True, **kw)
    assert asyn[0]["ids"] == sync[0]["ids"] and asyn[1]["ids"] == sync[1]["ids"]
    for t in range(6):
        a, s = asyn[0]["reports"][t], sync[0]["reports"][t]
        assert set(a["section_relative_perf_scores"]) == set(s["section_relative_perf_scores"]), t
        for n, per_rank in s["section_relative_perf_scores"].items():
            for r, v in per_rank.items():
                w = a["section_relative_perf_scores"][n][r]
                assert (np.isnan(v) and np.isnan(w)) or abs(v - w) < 1e-6, (t, n, r, v, w)


def test_asynchronous_ranks_stay_paired_when_one_leaves_its_plan_in_the_report_another_flags_new_names():
    """Asynchronous reports on an in-stream route: a rank with a new name runs its old plan with "ids missing" in its row and
    syncs names at the start of its NEXT report.  A peer whose occupied rows change in that very report is on the general path
    when it meets the flag; it used to sync by itself and exchange a second time -- every later exchange of the two ranks was
    then paired one report apart and the last one with nobody (a hang; found by tools/soak_mp.py on the peer route).  Now it
    keeps the report and syncs first next time, as the planned path does.  Same scores as the synchronous run; the new name
    one report later."""
    kw = dict(backend_kwargs={"emulate_fused": True}, env={"NVRX_EXCHANGE": "rccl", "NVRX_REPORT_TIMEOUT_S": "30"}, timeout=120)
    sync = run_ranks(workers.detector_async_rows_change_beside_a_new_name, 2, asynchronous=False, **kw)
    asyn = run_ranks(workers.detector_async_rows_change_beside_a_new_name, 2, asynchronous=True, **kw)
    late = "late_rank1_only"
    for r in range(2):
        for t in range(7):
            a, s = asyn[r][t], sync[r][t]
            for key in ("section_relative_perf_scores", "section_individual_perf_scores"):
                exp = dict(s[key])
                if r == 1 and t == 3:
                    assert late in exp and late not in a[key]
                    exp.pop(late)
                assert a[key].keys() == exp.keys(), (r, t, key, sorted(a[key]), sorted(exp))
                if key == "section_relative_perf_scores":
                    for n in exp:
                        for rk, v in exp[n].items():
                            w = a[key][n][rk]
                            assert (np.isnan(v) and np.isnan(w)) or abs(v - w) < 1e-6, (r, t, n, rk, v, w)


@pytest.mark.parametrize("route", ["c10d", "rccl"])
def test_randomised_detector_cycles_of_three_ranks_keep_their_collectives_paired(route):
    """The multi-process soak (tools/soak_mp.py, ``workers.detector_soak_ranks``) on the CPU checker backend for a few seconds:
    three gloo ranks, randomised ``Detector`` cycles -- asynchronous or not, gathered or not, all / relative / individual scores,
    sections that come and go per rank, names only one rank has -- on the default route and on an emulated in-call exchange.  Every
    report is checked; a rank that issues a collective its peers do not (both bugs the GPU form of this soak found were of that
    kind) leaves the job hanging, which the time limit turns into a failure."""
    import os

    out = run_ranks(workers.detector_soak_ranks, 3, timeout=150, backend_kwargs={"emulate_fused": True},
                    env={"NVRX_EXCHANGE": route, "NVRX_REPORT_TIMEOUT_S": "30", "NVRX_GPU_TIMING": "stamp"},
                    seconds=6.0, seed=int.from_bytes(os.urandom(2), "little"), gpu=False)
    assert out[0]["cycles"] > 10 and out[0]["cycles"] == out[1]["cycles"] == out[2]["cycles"], out


@pytest.mark.parametrize("asynchronous", [False, True])
def test_a_rank_with_an_empty_first_window_still_joins_the_first_reports_name_sync(asynchronous):
    """The deferral above must not reach a generator's FIRST ring report: a rank that has recorded nothing yet has "all its names"
    while every peer's are new and no rank has a plan to run -- the peers sync inside that report, and so must it."""
    kw = dict(backend_kwargs={"emulate_fused": True}, env={"NVRX_EXCHANGE": "rccl", "NVRX_REPORT_TIMEOUT_S": "30"}, timeout=120)
    out = run_ranks(workers.detector_async_first_window_empty_on_one_rank, 2, asynchronous=asynchronous, **kw)
    assert sorted(out[0][0]["section_relative_perf_scores"]) == ["a", "b"] and out[1][0]["section_relative_perf_scores"] == {}
    for t in range(1, 5):
        for r in range(2):
            rel = out[r][t]["section_relative_perf_scores"]
            assert sorted(rel) == ["a", "b"] and abs(rel["a"][r] - (1.0 if r == 0 else 0.5)) < 1e-6, (r, t, rel)


def test_an_asynchronous_generator_that_exchanges_nothing_never_calls_a_collective_for_its_new_names():
    """Individual scores only, nothing gathered: no report of such a generator holds a collective, its peers may be anywhere
    in their step.  Asynchronous, it runs the report in which one of ITS sections first appears on the old tables and takes
    the name in at the next one -- where it used to join a name sync nobody else was heading for (``all_gather_object`` on
    one rank: the rank hung; found by tools/soak_mp.py).  Two ranks, rank 1 alone meets a new section at report 3: both
    finish, the section scores from report 4 on."""
    out = run_ranks(workers.detector_async_individual_only, 2, timeout=90, backend_kwargs={"emulate_fused": True})
    assert out[0] == [["a", "b"]] * 6
    assert out[1][:3] == [["a", "b"]] * 3 and out[1][4:] == [["a", "b", "late_rank1_only"]] * 2, out[1]


@pytest.mark.parametrize("world", [1, 2, 4])
def test_asynchronous_reports_match_synchronous_ones_and_defer_new_names(world):
    """Asynchronous reports (enqueue now, wait on first read): same scores as the synchronous run for every report;
    a section that ONE rank meets at report 3 enters at report 4, after the name sync every rank runs at the start of
    that report (the synchronous run has it at report 3 already)."""
    # (an in-stream exchange route, which asynchronous multi-rank reports need: on the default route -- c10d, a host-driven
    #  collective on the job's group -- a report waits, see the next test)
    kw = dict(backend_kwargs={"emulate_fused": True}, env={"NVRX_EXCHANGE": "rccl"})
    sync = run_ranks(workers.detector_async_sequence, world, asynchronous=False, **kw)
    asyn = run_ranks(workers.detector_async_sequence, world, asynchronous=True, **kw)
    late = "late_rank1_only"
    for r in range(world):
        assert asyn[r]["ids"] == sync[r]["ids"]
    # the cached plan serves reports 1, 2 and 5 in both modes; asynchronous also runs the OLD plan at report 3
    assert sync[0]["planned"][1] and sync[0]["planned"][2]
    assert asyn[0]["planned"][1] and asyn[0]["planned"][2] and asyn[0]["planned"][5]
    for t in range(6):
        a, s = asyn[0]["reports"][t], sync[0]["reports"][t]
        assert a is not None and s is not None
        for key in ("section_relative_perf_scores", "section_individual_perf_scores"):
            exp = dict(s[key])
            if world > 1 and t == 3:
                assert late in exp and late not in a[key]   # one report later in asynchronous mode
                exp.pop(late)
            assert a[key].keys() == exp.keys(), (t, key)
            for n in exp:
                if n == late and key == "section_individual_perf_scores":
                    continue  # its history (best median so far) also starts one report later
                np.testing.assert_allclose(list(a[key][n].values()), list(exp[n].values()), rtol=1e-6, equal_nan=True)
        if world > 1 and t >= 4:
            assert late in a["section_relative_perf_scores"]


def test_ptl_callback_two_ranks_flags_the_slow_one():
    """The PTL callback end to end on two gloo ranks (CPU checker backend; GPU twin in tests/test_gpu_multiproc.py)."""
    res = run_ranks(workers.ptl_callback_run, 2, timeout=200, slow_rank=1)
    r0 = res[0]
    assert r0["interval"] is not None and res[1]["interval"] == r0["interval"]
    assert r0["logged"]["gpu_relative_perf/min"] < 0.7 <= r0["logged"]["gpu_relative_perf/max"]
    assert any("STRAGGLER DETECTION WARNING" in m and "rank=1" in m for m in r0["messages"])
    assert r0["should_stop"] and res[1]["should_stop"]


def test_poll_fails_fast_when_the_word_has_moved_past():
    """nvrx_poll_u32 waits for equality with a sequence number that only moves forward: a word already past it means
    the result block was reused before this report was collected -- reported at once, never waited for."""
    from nvrx_straggler import _native

    lib = _native.load()
    word = ctypes.c_uint32(7)
    addr = ctypes.addressof(word)
    assert lib.nvrx_poll_u32(addr, 7, 0.01) == 0
    assert lib.nvrx_poll_u32(addr, 5, 5.0) == -1 and b"past the awaited 5" in lib.nvrx_last_error()
    assert lib.nvrx_poll_u32(addr, 9, 0.01) == -62 and b"not seen after" in lib.nvrx_last_error()
    word.value = 0
    assert lib.nvrx_poll_u32(addr, 1, 0.01) == -62  # a fresh block (0) is simply not there yet


def test_a_report_read_on_another_thread_while_the_next_ones_are_generated():
    """Two-thread stress of the lazily read result block (reporting._LiveBlock.head / stats vs Workspace.settle ->
    detach): a reader thread builds the mappings of the most recent report while the main thread already generates the
    next ones on the same workspace.  Every report carries values only it can have (medians scale with the report
    number), so a block copied after it was overwritten, or half of each, shows up as a wrong number."""
    import threading

    from nvrx_straggler import Statistic, backend
    from nvrx_straggler.reporting import ReportGenerator
    from oracle_backend import OracleBackend

    be = OracleBackend(emulate_fused=True)
    backend.set_backend(be)
    try:
        S, N = 12, 33
        rings = be.make_rings(1, S, 64)
        rows = {f"s{i}": rings.row_for(0, f"s{i}") for i in range(S)}
        none = {}
        gen = ReportGenerator(["relative_perf_scores", "individual_perf_scores"], gather_on_rank0=True, node_name="n")
        base = np.linspace(1.0, 2.0, N, dtype=np.float32)

        def arm(t):
            for i in range(S):
                rings.samples[i, :N] = base * np.float32(1 + (t % 13)) * np.float32(1 + i)
            rings.total[:] = N

        latest = [None]
        errors = []
        stop = threading.Event()
        reads = [0]

        def reader():
            while not stop.is_set():
                item = latest[0]
                if item is None:
                    continue
                t, rep = item
                try:
                    summ = rep.local_section_summaries
                    flagged = rep.identify_stragglers()
                    for i in (0, S // 2, S - 1):
                        want = np.float32(np.median(base)) * np.float32(1 + (t % 13)) * np.float32(1 + i)
                        got = summ[f"s{i}"][Statistic.MED]
                        if abs(got - want) > 1e-4 * want:
                            errors.append((t, i, got, want))
                    if flagged["straggler_gpus_relative"]:
                        errors.append((t, "flagged", flagged))
                    reads[0] += 1
                except Exception as e:  # noqa: BLE001
                    errors.append((t, repr(e)))
                    return

        th = threading.Thread(target=reader, daemon=True)
        th.start()
        for t in range(3000):
            arm(t)
            rep = gen.generate_report_from_rings(rings, rows, none)
            latest[0] = (t, rep)
            rings.reset()
        stop.set()
        th.join(10)
        assert not errors, errors[:5]
        assert reads[0] > 50, reads[0]   # the reader really ran beside the generator
        assert gen._ring_plan is not None and gen._ring_plan.fused  # the lazily read block was the path under test
    finally:
        backend.set_backend(None)


def test_region_timing_flattens_gpu_scores_when_the_region_holds_a_collective():
    """How far the default GPU timing (one device-stamped row per profiled REGION, key ``hipevent::<section>``) moves
    the relative GPU score away from the reference's per-kernel score when the region contains a collective
    (INTEGRATION.md quotes these numbers).  Four ranks, rank 2 computes 25 % slower; the step ends in an all-reduce,
    so every rank's region lasts as long as the slowest rank's compute plus the wire time.

    * per-kernel rows (``NVRX_GPU_TIMING=kernels``, the reference's data model): the collective kernel ``ncclDev*`` is
      dropped (reporting.py:330-336), the score is min(compute) / compute = 0.8 for the slow rank -> flagged at 0.85;
    * one row per region: all four regions last the same, every score is 1.0 -> the slow GPU is invisible.
    """
    compute = [10.0, 10.0, 12.5, 10.0]   # ms of GPU compute per step
    wire = 1.0
    region = max(compute) + wire         # what a device stamp around the whole step measures on every rank

    def stat(v, n=100):
        return {"MIN": v, "MAX": v, "MED": v, "AVG": v, "STD": 0.0, "NUM": n}

    per_kernel = [({}, {"gemm_blk_256_1_1_grid_512_1_1": stat(c), "ncclDevKernel_AllReduce_blk_1_1_1_grid_1_1_1": stat(region - c)})
                  for c in compute]
    per_region = [({}, {"hipevent::train_step": stat(region)}) for _ in compute]
    out = {}
    for name, step in (("kernels", per_kernel), ("regions", per_region)):
        sc = {"world_size": 4, "scores_to_compute": ["relative_perf_scores"], "gather_on_rank0": True, "steps": [step],
              "thresholds": [0.85]}
        res = run_ranks(workers.scoring_scenario, 4, scenario=sc)
        out[name] = res[0]["reports"][0]
    k, r = out["kernels"], out["regions"]
    assert [round(k["gpu_relative_perf_scores"][i], 4) for i in range(4)] == [1.0, 1.0, 0.8, 1.0]
    assert k["stragglers"]["0.85"]["straggler_gpus_relative"] == [2]
    assert [round(r["gpu_relative_perf_scores"][i], 4) for i in range(4)] == [1.0, 1.0, 1.0, 1.0]
    assert r["stragglers"]["0.85"]["straggler_gpus_relative"] == []


def test_c_dict_builders_equal_the_python_builders(monkeypatch):
    """``_nvrx_pyread`` (csrc/nvrx_pyread.c) builds a Report's nested dicts straight from the f32 blocks; the Python builders
    (numpy ``tolist`` + ``dict(zip)``) are what runs when the module was not built.  Same keys in the same order, same
    values (NaN and inf included), NUM an ``int`` -- for the identity column order of a gathered report and for a
    rank's own section order."""
    import math

    from nvrx_straggler import Statistic, reporting

    assert reporting._pyread is not None, "csrc/nvrx_pyread.c was not built (make -C nvidia-resiliency-ext_amd/csrc)"
    rng = np.random.default_rng(3)
    R, S = 8, 37
    W = 2 + 2 * S
    scores = rng.uniform(0.1, 1.0, (R, W)).astype(np.float32)
    scores[2, 5] = np.nan
    scores[7, 2 + S + 3] = np.inf
    stats = rng.uniform(1.0, 9.0, (S + 5, 8)).astype(np.float32)
    stats[:, 5] = rng.integers(0, 70000, S + 5)
    stats[3, 4] = np.nan

    def same(a, b):
        assert list(a) == list(b)                      # key order
        for k in a:
            assert list(a[k]) == list(b[k]), k
            for kk in a[k]:
                x, y = a[k][kk], b[k][kk]
                assert type(x) is type(y), (k, kk, type(x), type(y))
                assert (x == y) or (math.isnan(x) and math.isnan(y)), (k, kk, x, y)

    for order in ("identity", "permuted"):
        v = reporting._View()
        v.S, v.ranks = S, range(3, 3 + R)
        names = [f"section_{i:02d}" for i in range(S)]
        if order == "permuted":
            names = [names[i] for i in rng.permutation(S)]
        v.names = names
        v.cols = {n: int(n.split("_")[1]) for n in names}
        v.has_rel = v.has_indiv = True
        v.section_rows = {n: 2 + v.cols[n] for n in names}
        v.kernel_rows = {"k_a": 0, "k_b": S + 4}
        v.layout, v.thresholds = None, None
        got, want = {}, {}
        for use_c, out in ((True, got), (False, want)):
            with monkeypatch.context() as m:
                if not use_c:
                    m.setattr(reporting, "_pyread", None)
                src = reporting._ScoreSource(v)
                src.scores, src.flags, src.stats = scores, np.zeros((R, W), np.uint8), stats
                for f in sorted(reporting._LAZY_FIELDS):
                    out[f] = src.build(f)
        for f in got:
            if f.startswith("gpu_"):
                assert got[f] == want[f] or all(math.isnan(got[f][k]) == math.isnan(want[f][k]) for k in got[f])
            else:
                same(got[f], want[f])
        assert all(type(d[Statistic.NUM]) is int for d in got["local_section_summaries"].values())
        assert list(got["section_relative_perf_scores"]) == names
    flagged = {f"s{i}": {reporting.StragglerId(rank=i % 3, node="n"), reporting.StragglerId(rank=7, node="m")} for i in range(9)}
    for use_c in (True, False):
        with monkeypatch.context() as m:
            if not use_c:
                m.setattr(reporting, "_pyread", None)
            copy = reporting._copy_sets(flagged)
        assert copy == flagged and list(copy) == list(flagged) and all(type(v) is set for v in copy.values())
        assert all(copy[k] is not flagged[k] for k in flagged)


@pytest.mark.parametrize("order", ["identity", "permuted"])
def test_c_flag_decoder_equals_the_python_decoder(order, monkeypatch):
    """``_nvrx_pyread.flagged`` (identify_stragglers at the kernel's thresholds: flag bytes -> sets, no numpy) against
    ``_DeviceFlags.decode``: random flag tables of every density (none, one, many per column, all), both score families
    and each alone, a rank's own section order, rows [lo, hi) of a larger table; the memo hands out equal but fresh
    sets, and what a caller does to its sets never shows up in the next answer."""
    from nvrx_straggler import reporting

    assert reporting._pyread is not None and hasattr(reporting._pyread, "flagged")
    rng = np.random.default_rng(11)
    R_all, S = 12, 23
    W = 2 + 2 * S
    names = [f"section_{i:02d}" for i in range(S)]
    if order == "permuted":
        names = [names[i] for i in rng.permutation(S)][: S - 3]   # a rank's own order, not every column used
    cols = {n: int(n.split("_")[1]) for n in names}
    off_f = 64 + R_all * W * 4
    for has_rel, has_indiv in ((True, True), (True, False), (False, True)):
        for lo, hi in ((0, R_all), (3, 9), (5, 6)):
            v = reporting._View()
            v.S, v.ranks, v.names, v.cols = S, range(lo, hi), names, cols
            v.has_rel, v.has_indiv = has_rel, has_indiv
            v.layout = (64, off_f, off_f + R_all * W, R_all, W, lo, hi, 0)
            v.thresholds = (0.75, 0.75, 0.75, 0.75)
            rank_to_node = {r: f"node{r // 4}" for r in range(R_all)}
            for density in (0.0, 0.002, 0.05, 0.5, 1.0, 0.05):
                table = (rng.random((R_all, W)) < density).astype(np.uint8)
                blob = np.zeros(off_f + R_all * W, dtype=np.uint8)
                blob[off_f:] = table.reshape(-1)
                src = reporting._ScoreSource(v, blob.copy())
                got = src.flagged(rank_to_node, (0.75, 0.75, 0.75, 0.75))
                assert src.flagged(rank_to_node, (0.75, 0.75, 0.75, 0.7)) is None        # other thresholds: not ours
                with monkeypatch.context() as m:
                    m.setattr(reporting._ScoreSource, "flagged", lambda self, a, b: None)
                    rep = reporting.Report._from_device(reporting._ScoreSource(v, blob.copy()), rank_to_node, 0.0, True, 0)
                    want = rep.identify_stragglers()
                assert got == want, (has_rel, has_indiv, lo, hi, density)
                assert list(got["straggler_sections_relative"]) == list(want["straggler_sections_relative"])
                assert list(got["straggler_sections_individual"]) == list(want["straggler_sections_individual"])
                # the same table again: served from the memo, equal, and not the same objects
                again = reporting._ScoreSource(v, blob.copy()).flagged(rank_to_node, (0.75, 0.75, 0.75, 0.75))
                assert again == got
                for k in got:
                    assert again[k] is not got[k]
                    if isinstance(got[k], dict):
                        assert all(again[k][n] is not got[k][n] for n in got[k])
                # a caller emptying its sets changes nobody else's answer
                for k in again:
                    again[k].clear()
                assert reporting._ScoreSource(v, blob.copy()).flagged(rank_to_node, (0.75, 0.75, 0.75, 0.75)) == want
    with pytest.raises(ValueError):
        reporting._pyread.flagged(b"\x00" * 10, 0, 2, 6, 2, True, True, [1, 2], ("a", "b"), None, None)   # buffer too short
    with pytest.raises(ValueError):
        reporting._pyread.flagged(b"\x01" * 12, 0, 2, 6, 2, True, True, [1, 2], ("a", "b"), (0, 5), None)  # column out of range


def test_rccl_kernel_names_of_the_installed_library_are_filtered():
    """The reference keeps collective kernels out of the GPU score by the substring "ncclDev" (reporting.py:330-336:
    NCCL's ``ncclDevKernel_*``).  RCCL's device kernels have other names -- ``rcclGenericKernel<N, bool>`` and the
    ``mscclKernel_*`` family -- so the filter is checked against what the INSTALLED librccl.so actually launches: every
    kernel entry point of the library (its host-side launch stubs carry the device kernels' mangled names, which is what
    the tracer's keys are made of) must be dropped by ``is_collective_kernel``, and by the reference's plain substring
    too (the mangled names spell out the argument types ``ncclDevKernelArgsStorage`` / ``ncclDevComm``)."""
    import importlib.util
    import re

    from nvrx_straggler import reporting

    spec = importlib.util.find_spec("torch")
    path = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
    if not os.path.exists(path):
        pytest.skip("no librccl.so next to this PyTorch")
    names = set()
    pat = re.compile(rb"_Z\d+(?:rcclGenericKernel|mscclKernel|ncclDevKernel|ncclKernel)[A-Za-z0-9_]*")
    with open(path, "rb") as f:
        tail = b""
        while True:
            chunk = f.read(16 << 20)
            if not chunk:
                break
            buf = tail + chunk
            names.update(m.group(0).decode() for m in pat.finditer(buf))
            tail = buf[-256:]
    assert len(names) >= 8, sorted(names)[:10]
    assert any("rcclGenericKernel" in n for n in names) and any("mscclKernel" in n for n in names), sorted(names)[:10]
    for n in names:
        key = f"{n}_blk_256_1_1_grid_64_1_1"
        assert reporting.is_collective_kernel(key), n
        assert "ncclDev" in key, n   # the reference's own filter hits RCCL's mangled names as well
    # and nothing else is caught
    for n in ("Cijk_Ailk_Bljk_SB_MT128x128x16_blk_256_1_1_grid_64_1_1", "_ZN2at6native29vectorized_elementwise_kernelILi4E_blk_256_1_1_grid_8_1_1",
              "hipevent::train_step"):
        assert not reporting.is_collective_kernel(n)
    kept = reporting.ReportGenerator._filter_out_nccl_kernels if hasattr(reporting.ReportGenerator, "_filter_out_nccl_kernels") else None
    if kept is not None:
        some = {f"{next(iter(names))}_blk_1_1_1_grid_1_1_1": 1, "gemm_blk_1_1_1_grid_1_1_1": 2}
        try:
            out = kept(some)
        except TypeError:
            out = kept(None, some)
        assert list(out) == ["gemm_blk_1_1_1_grid_1_1_1"]


def test_gpu_timing_mode_defaults(monkeypatch):
    """``ktrace.timing_mode``: a name in NVRX_GPU_TIMING wins; unset, a process of a multi-rank job whose HIP runtime is not
    up yet gets per-kernel tracing (the reference's data model), everything else region stamps."""
    from nvrx_straggler import ktrace

    calls = []
    monkeypatch.setattr(ktrace, "setup", lambda max_pending=0: calls.append("setup"))
    monkeypatch.setattr(ktrace, "_hip_is_up", lambda: False)
    real_exists = os.path.exists
    monkeypatch.setattr(ktrace.os.path, "exists", lambda p: True if p == "/dev/kfd" else real_exists(p))

    def mode(env):
        for k in ("NVRX_GPU_TIMING",) + ktrace._JOB_SIZE_VARS:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ktrace._reset_mode_for_tests()
        del calls[:]
        return ktrace.timing_mode(), list(calls)

    try:
        assert mode({}) == ("stamp", [])
        assert mode({"WORLD_SIZE": "1"}) == ("stamp", [])
        assert mode({"WORLD_SIZE": "8"}) == ("kernels", ["setup"])
        assert "WORLD_SIZE=8" in ktrace.mode_note()
        # launchers that do not follow torchrun's convention: srun, mpirun, PMI -- WORLD_SIZE, where present, has the word
        assert mode({"SLURM_NTASKS": "16"}) == ("kernels", ["setup"]) and "SLURM_NTASKS=16" in ktrace.mode_note()
        assert mode({"OMPI_COMM_WORLD_SIZE": "4"}) == ("kernels", ["setup"])
        assert mode({"PMI_SIZE": "2"}) == ("kernels", ["setup"])
        assert mode({"SLURM_NTASKS": "1", "PMI_SIZE": "x"}) == ("stamp", [])
        assert mode({"WORLD_SIZE": "1", "SLURM_NTASKS": "16"}) == ("stamp", [])
        assert mode({"WORLD_SIZE": "8", "NVRX_GPU_TIMING": "stamp"}) == ("stamp", [])
        assert mode({"WORLD_SIZE": "8", "NVRX_GPU_TIMING": "event"}) == ("event", [])
        assert mode({"NVRX_GPU_TIMING": "kernels"}) == ("kernels", ["setup"])
        assert mode({"WORLD_SIZE": "8", "NVRX_GPU_TIMING": "auto"}) == ("kernels", ["setup"])
        monkeypatch.setattr(ktrace, "_hip_is_up", lambda: True)
        assert mode({"WORLD_SIZE": "8"}) == ("stamp", [])
        assert "before" in ktrace.mode_note()
        monkeypatch.setattr(ktrace, "_hip_is_up", lambda: False)

        def broken(max_pending=0):
            raise RuntimeError("no sdk")

        monkeypatch.setattr(ktrace, "setup", broken)
        assert mode({"WORLD_SIZE": "8"})[0] == "stamp" and "no sdk" in ktrace.mode_note()
        assert mode({"NVRX_GPU_TIMING": "kernels"})[0] == "kernels"   # asked for by name: the error surfaces at first use
        monkeypatch.setattr(ktrace.os.path, "exists", lambda p: False if p == "/dev/kfd" else real_exists(p))
        monkeypatch.setattr(ktrace, "setup", lambda max_pending=0: calls.append("setup"))
        assert mode({"WORLD_SIZE": "8"}) == ("stamp", [])
    finally:
        monkeypatch.undo()
        ktrace._reset_mode_for_tests()


def test_tool_search_guard_keeps_rocprofiler_sdk_off_the_large_libraries():
    """rocprofiler-sdk looks for tools by reading every library of the link map front to back (10.7 GB in a PyTorch
    process: the start-up stall of rounds 1-3).  Registration through ``nvrx_ktrace_setup`` hides the large ones from that
    one search: in a fresh interpreter with PyTorch loaded it must read well under 1 GB and take seconds (no GPU needed:
    the search runs when the SDK is configured, before any device is touched)."""
    code = r'''
import os, sys, time
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd")]
import torch
os.environ["NVRX_GPU_TIMING"] = "kernels"
def rchar():
    return int(dict(l.split(": ") for l in open("/proc/self/io").read().strip().splitlines())["rchar"])
a, t = rchar(), time.monotonic()
from nvrx_straggler import ktrace
mode = ktrace.timing_mode()
lib = ktrace.load()
print("RESULT", mode, int(lib.nvrx_ktrace_ready()), int(lib.nvrx_ktrace_hidden_libraries()), (rchar() - a) / 1e9, time.monotonic() - t)
'''
    env = dict(os.environ)
    for k in ("NVRX_GPU_TIMING", "NVRX_DEBUG_KTRACE_FORCE", "NVRX_KTRACE_SCAN_GUARD", "ROCP_TOOL_LIBRARIES"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    _, mode, ready, hidden, gb, secs = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert mode == "kernels" and int(ready) == 1
    assert int(hidden) >= 5, hidden          # libtorch_*, the BLAS / solver / MIOpen libraries ...
    assert float(gb) < 1.0, gb               # (10.7 GB without the guard)
    assert float(secs) < 20.0, secs


def test_flag_memo_is_keyed_on_everything_the_answer_depends_on():
    """``_nvrx_pyread.flagged`` answers an unchanged flag table from a memo; the memo must miss when the same bytes are
    decoded for other families / names / column tables (a memo list shared across views would otherwise hand out the
    wrong straggler sets)."""
    from nvrx_straggler import reporting

    pr = reporting._pyread
    assert pr is not None
    R, S = 4, 3
    W = 2 + 2 * S
    flags = np.zeros((R, W), dtype=np.uint8)
    flags[2, :] = 1
    buf = flags.tobytes()
    ids = [reporting.StragglerId(rank=r, node="n") for r in range(R)]
    names = ("a", "b", "c")
    memo = [None, None]
    both = pr.flagged(buf, 0, R, W, S, True, True, ids, names, None, memo)
    assert both[1] == {ids[2]} and set(both[3]) == set(names)
    again = pr.flagged(buf, 0, R, W, S, True, True, ids, names, None, memo)
    assert again == both and again[2] is not both[2]            # a hit: equal, fresh copies
    rel_only = pr.flagged(buf, 0, R, W, S, True, False, ids, names, None, memo)
    assert rel_only[1] == set() and rel_only[3] == {}            # individual family off: must not come from the memo
    assert rel_only[0] == {ids[2]} and set(rel_only[2]) == set(names)
    other_names = ("x", "y", "z")
    renamed = pr.flagged(buf, 0, R, W, S, True, True, ids, other_names, None, memo)
    assert set(renamed[2]) == set(other_names)
    permuted = pr.flagged(buf, 0, R, W, S, True, True, ids, other_names, (2, 1, 0), memo)
    assert set(permuted[2]) == set(other_names)


def test_python_summaries_builder_rejects_a_non_finite_count_like_the_c_builder(monkeypatch):
    from nvrx_straggler import reporting

    stats = np.ones((2, 8), dtype=np.float32)
    stats[1, 5] = np.nan
    rows = {"a": 0, "b": 1}
    with pytest.raises((ValueError, OverflowError)):
        reporting._summaries_from_rows(rows, stats)
    monkeypatch.setattr(reporting, "_pyread", None)
    with pytest.raises(ValueError):
        reporting._summaries_from_rows(rows, stats)


def test_detector_binds_its_device_side_at_first_use_not_at_initialize():
    """The reference's own example initialises the detector BEFORE it selects its GPU (examples/straggler/example.py:60-66:
    ``Detector.initialize()``, ``init_process_group``, ``torch.cuda.set_device(local_rank)``).  CUPTI does not care; device
    memory does -- rings created inside ``initialize`` would sit on GPU 0 in every rank of such a script.  So
    ``initialize`` only checks that the engine CAN run; rings and profiler come into being at the first section / report /
    attribute access, on the device that is current then; a detector that was never used shuts down without having
    created anything."""
    from nvrx_straggler import Detector, backend
    from oracle_backend import OracleBackend

    be = OracleBackend()
    made = []
    real = be.make_rings

    def counting(*a, **k):
        made.append(real(*a, **k))
        return made[-1]

    be.make_rings = counting
    backend.set_backend(be)
    try:
        Detector.initialize(scores_to_compute="all", gather_on_rank0=False, node_name="n0")
        assert made == [] and Detector._rings is None and Detector._cupti_manager is None   # nothing is bound yet
        Detector.shutdown()                                                                  # ... and nothing to undo
        assert made == []
        Detector.initialize(scores_to_compute="all", gather_on_rank0=False, node_name="n0")
        with Detector.detection_section("first", profile_cuda=False):
            pass
        assert len(made) == 1 and Detector.rings is made[0] and Detector.cupti_manager is not None
        rep = Detector.generate_report()
        assert list(rep.local_section_summaries) == ["first"] and len(made) == 1
        Detector.shutdown()
        assert Detector._rings is None
        Detector.initialize(scores_to_compute="all", gather_on_rank0=False, node_name="n0")
        assert Detector.rings is not None and len(made) == 2    # touching the attribute is a first use as well
    finally:
        if Detector.initialized:
            Detector.shutdown()
        backend.set_backend(None)


def test_inplace_filled_dicts_are_ordinary_dicts():
    """``_nvrx_pyread`` fills CLONED dicts in place where it can (CPython 3.10: a clone of a template dict already holds
    the keys; the values are stored straight into its entries -- private layout, checked by a self-test at import and
    before every fill).  What comes out must be indistinguishable from the dicts built insert by insert: same contents
    and order, picklable / deep-copyable / JSON-serialisable, free to grow, shrink and be collected."""
    import copy
    import gc
    import json

    from nvrx_straggler import Statistic, reporting
    from nvrx_straggler.statistics import STAT_KEYS

    pr = reporting._pyread
    assert pr is not None
    was = pr.inplace()
    rng = np.random.default_rng(5)
    R, S = 8, 64
    W = 2 + 2 * S
    scores = rng.uniform(0.5, 1.0, (R, W)).astype(np.float32)
    stats = rng.uniform(1.0, 2.0, (S, 8)).astype(np.float32)
    stats[:, 5] = rng.integers(1, 10000, S)
    names = tuple(f"section_{i:03d}" for i in range(S))
    ranks = tuple(range(R))
    rows = tuple(range(S))
    try:
        pr.inplace(0)
        plain = (pr.sections(names, ranks, scores, 0, R, W, 2, None, 2 + S), pr.summaries(names, STAT_KEYS, stats, rows))
        if not pr.inplace(1):
            pytest.skip("the in-place fill is not available on this interpreter (it is compiled for CPython 3.10 only)")
        fast = (pr.sections(names, ranks, scores, 0, R, W, 2, None, 2 + S), pr.summaries(names, STAT_KEYS, stats, rows))
        assert fast == plain
        # with the caller's name template the OUTER dicts are filled clones too -- and, holding dicts, known to the collector
        tmpl = dict.fromkeys(names)
        outer = (pr.sections(names, ranks, scores, 0, R, W, 2, None, 2 + S, tmpl), pr.summaries(names, STAT_KEYS, stats, rows, tmpl))
        assert outer == plain and all(v is None for v in tmpl.values()) and list(outer[1]) == list(names)
        assert gc.is_tracked(outer[0][0]) and gc.is_tracked(outer[0][1]) and gc.is_tracked(outer[1])
        # inner dicts are tracked exactly when a dict built insert by insert would be (int -> float: no; enum member -> number: yes)
        assert gc.is_tracked(outer[0][0][names[0]]) == gc.is_tracked(plain[0][0][names[0]]) == False  # noqa: E712
        assert gc.is_tracked(outer[1][names[0]]) == gc.is_tracked(plain[1][names[0]])
        stale = dict.fromkeys(names[:-1])                     # a template that does not fit is ignored, not trusted
        assert pr.summaries(names, STAT_KEYS, stats, rows, stale) == plain[1]
        for a, b in ((fast[0][0], plain[0][0]), (fast[0][1], plain[0][1]), (fast[1], plain[1])):
            assert list(a) == list(b)
            for k in a:
                assert type(a[k]) is dict and list(a[k]) == list(b[k]) and [type(v) for v in a[k].values()] == [type(v) for v in b[k].values()]
        assert type(fast[1][names[0]][Statistic.NUM]) is int
        assert pickle.loads(pickle.dumps(fast)) == plain and copy.deepcopy(fast) == plain
        assert json.loads(json.dumps(fast[0][0])) == json.loads(json.dumps(plain[0][0]))
        inner = fast[0][0][names[3]]
        inner[99] = 1.5                       # grows (a resize of the cloned key table) ...
        for r in range(4):
            del inner[r]                      # ... shrinks ...
        inner.update({r: 0.25 for r in range(100, 140)})
        assert len(inner) == 45 and inner[7] == plain[0][0][names[3]][7] and inner[120] == 0.25
        inner["self"] = inner                 # ... and a cycle through it is collectable
        del inner, fast
        gc.collect()
        many = [pr.sections(names, ranks, scores, 0, R, W, 2, None) for _ in range(200)]   # clones of one template per call
        assert all(m == plain[0][0] for m in many[::37])
        del many
        gc.collect()
        # keys that cannot share a template (duplicates) still come out right, the public way
        dup = pr.sections(("a", "b"), (0, 0, 1), scores[:3], 0, 3, W, 2, None)
        assert dup == {"a": {0: float(scores[1, 2]), 1: float(scores[2, 2])}, "b": {0: float(scores[1, 3]), 1: float(scores[2, 3])}}
    finally:
        pr.inplace(1 if was else 0)


def test_mappings_nobody_holds_are_recycled_and_held_ones_are_never_touched():
    """``_nvrx_pyread`` keeps the last mapping it built per plan (a list the caller owns) and, when NOTHING else refers to it,
    hands the very same dicts out again with the next report's values swapped in -- no dict is created.  The contract: what
    comes back always equals what the plain builder gives; an object somebody still holds -- the outer dict, one inner
    dict, a value -- is never modified; a mapping the caller changed (key added, value replaced) is not trusted."""
    import gc
    import tracemalloc

    from nvrx_straggler import reporting
    from nvrx_straggler.statistics import STAT_KEYS

    pr = reporting._pyread
    assert pr is not None
    if not pr.inplace():
        pytest.skip("recycling rides on the in-place fill (CPython 3.10 only)")
    rng = np.random.default_rng(11)
    R, S = 8, 64
    W = 2 + 2 * S
    names = tuple(f"section_{i:03d}" for i in range(S))
    ranks = tuple(range(R))
    rows = tuple(range(S))
    tmpl = dict.fromkeys(names)

    def blocks():
        sc = rng.uniform(0.5, 1.0, (R, W)).astype(np.float32)
        st = rng.uniform(1.0, 2.0, (S, 8)).astype(np.float32)
        st[:, 5] = rng.integers(1, 10000, S)
        return sc, st

    def plain(sc, st):
        return pr.sections(names, ranks, sc, 0, R, W, 2, None, 2 + S), pr.summaries(names, STAT_KEYS, st, rows)

    keep_s, keep_m = [None, None], [None]

    def build(sc, st):
        return pr.sections(names, ranks, sc, 0, R, W, 2, None, 2 + S, tmpl, keep_s), pr.summaries(names, STAT_KEYS, st, rows, tmpl, keep_m)

    # 1. dropped by the caller -> the same objects come back, refilled
    sc, st = blocks()
    (a, b), m = build(sc, st)
    assert ((a, b), m) == plain(sc, st) and keep_s[0] is a and keep_s[1] is b and keep_m[0] is m
    ids = (id(a), id(b), id(m), id(a[names[5]]), id(m[names[5]]))
    del a, b, m
    sc, st = blocks()
    (a, b), m = build(sc, st)
    assert ((a, b), m) == plain(sc, st)
    assert (id(a), id(b), id(m), id(a[names[5]]), id(m[names[5]])) == ids
    assert type(m[names[0]][STAT_KEYS[5]]) is int and list(a) == list(names) and list(a[names[0]]) == list(ranks)

    # 2. the caller still holds them -> fresh objects, the held ones keep their values
    held = ((a, b), m)
    frozen = plain(sc, st)
    sc2, st2 = blocks()
    (a2, b2), m2 = build(sc2, st2)
    assert a2 is not a and b2 is not b and m2 is not m
    assert ((a2, b2), m2) == plain(sc2, st2) and held == frozen
    del a, b, m, held

    # 3. only an INNER dict (and one value) is held -> the outer is recycled, that inner one is replaced, not modified
    inner_held, value_held = a2[names[7]], b2[names[2]][3]
    inner_frozen, outer_id, other_inner_id = dict(inner_held), id(a2), id(a2[names[8]])
    minner_held = m2[names[1]]
    minner_frozen = dict(minner_held)
    del a2, b2, m2
    sc3, st3 = blocks()
    (a3, b3), m3 = build(sc3, st3)
    assert ((a3, b3), m3) == plain(sc3, st3)
    assert id(a3) == outer_id and id(a3[names[8]]) == other_inner_id
    assert a3[names[7]] is not inner_held and inner_held == inner_frozen
    assert m3[names[1]] is not minner_held and minner_held == minner_frozen
    assert isinstance(value_held, float)
    del inner_held, minner_held

    # 4. a mapping the caller changed is not trusted: grown, shrunk, a value that is not ours
    a3["extra"] = {}
    b3[names[0]][0] = ["not", "a", "float"]
    m3[names[4]] = "replaced"
    del m3[names[9]]
    del a3, b3, m3
    sc4, st4 = blocks()
    (a4, b4), m4 = build(sc4, st4)
    assert ((a4, b4), m4) == plain(sc4, st4)
    assert id(a4) != outer_id and "extra" not in a4 and list(m4) == list(names)

    # 5. an error in the middle of a refill leaves no half-built mapping behind
    del a4, b4, m4
    bad = st4.copy()
    bad[10, 5] = np.nan
    with pytest.raises(ValueError):
        pr.summaries(names, STAT_KEYS, bad, rows, tmpl, keep_m)
    assert keep_m == [None]
    assert pr.summaries(names, STAT_KEYS, st4, rows, tmpl, keep_m) == plain(sc4, st4)[1]

    # 6. nothing leaks: many refills, steady memory
    (a, b), m = build(sc4, st4)
    del a, b, m
    gc.collect()
    tracemalloc.start()
    base = tracemalloc.get_traced_memory()[0]
    for _ in range(300):
        (a, b), m = build(sc4, st4)
        del a, b, m
    grown = tracemalloc.get_traced_memory()[0] - base
    tracemalloc.stop()
    assert grown < 64 * 1024, grown

    # 7. with the in-place fill off nothing is recycled, everything is still right
    try:
        pr.inplace(0)
        (a, b), m = build(sc4, st4)
        first = id(a)
        del a, b, m
        (a, b), m = build(sc3, st3)
        assert ((a, b), m) == plain(sc3, st3)
    finally:
        pr.inplace(1)


@pytest.mark.parametrize("R,S", [(1, 1), (3, 7), (70, 5), (9, 300), (70, 300)])
def test_c_builders_beyond_their_stack_buffers(R, S):
    """``_nvrx_pyread`` keeps its per-call scratch (values of one inner dict, hashes, column table) on the stack up to 64
    ranks / 256 names and on the heap beyond: both sides of both limits must give what the Python builders give, with the
    in-place fill on and off, for the identity column order and a permuted one."""
    from nvrx_straggler import reporting
    from nvrx_straggler.statistics import STAT_KEYS

    pr = reporting._pyread
    assert pr is not None
    was = pr.inplace()
    rng = np.random.default_rng(R * 1000 + S)
    W = 2 + 2 * S
    scores = rng.uniform(0.1, 1.0, (R, W)).astype(np.float32)
    scores[rng.integers(0, R), rng.integers(0, W)] = np.nan
    stats = rng.uniform(1.0, 9.0, (S, 8)).astype(np.float32)
    stats[:, 5] = rng.integers(1, 60000, S)
    names = tuple(f"s{i}" for i in range(S))
    ranks = tuple(range(100, 100 + R))
    perm = tuple(int(i) for i in rng.permutation(S))
    rows = tuple(int(i) for i in rng.permutation(S))

    def expect_sections(first, cols):
        return {n: {ranks[r]: float(scores[r, first + (cols[i] if cols else i)]) for r in range(R)} for i, n in enumerate(names)}

    def same(a, b):
        assert list(a) == list(b)
        for k in a:
            assert list(a[k]) == list(b[k])
            for kk in a[k]:
                x, y = a[k][kk], b[k][kk]
                assert type(x) is type(y) and (x == y or (x != x and y != y)), (k, kk, x, y)

    try:
        for on in (0, 1):
            pr.inplace(on)
            for cols in (None, perm):
                one = pr.sections(names, ranks, scores, 0, R, W, 2, cols)
                same(one, expect_sections(2, cols))
                a, b = pr.sections(names, ranks, scores, 0, R, W, 2, cols, 2 + S, dict.fromkeys(names))
                same(a, expect_sections(2, cols))
                same(b, expect_sections(2 + S, cols))
            got = pr.summaries(names, STAT_KEYS, stats, rows, dict.fromkeys(names))
            exp = {n: {k: (int(stats[rows[i], j]) if j == 5 else float(stats[rows[i], j])) for j, k in enumerate(STAT_KEYS)} for i, n in enumerate(names)}
            same(got, exp)
            col0 = pr.ranks(ranks, scores, 0, R, W, 1)
            assert list(col0) == list(ranks) and all((col0[ranks[r]] == float(scores[r, 1])) or scores[r, 1] != scores[r, 1] for r in range(R))
            # an offset into a larger buffer (the live result block is addressed this way)
            blob = b"\x00" * 64 + scores.tobytes()
            same(pr.sections(names, ranks, blob, 64, R, W, 2, None), expect_sections(2, None))
    finally:
        pr.inplace(1 if was else 0)


def test_stream_priority_default_is_high_for_a_single_process_and_normal_for_a_multi_rank_job(monkeypatch):
    """``backend._stream_priority``: the detector's own streams run at high HIP priority in a single-process job (measured
    gain, nothing to interact with); in a multi-rank job they carry the report's RCCL all-gather beside the job's own
    collectives, and the default stays normal; ``NVRX_STREAM_PRIORITY`` names it outright (the library applies the same rule to the resident scorer's stream)."""
    from nvrx_straggler import backend, ktrace

    def prio(env):
        for k in ("NVRX_STREAM_PRIORITY",) + ktrace._JOB_SIZE_VARS:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return backend._stream_priority(), os.environ.get("NVRX_STREAM_PRIORITY")

    assert prio({}) == (-1, None)                       # (nothing is exported: child processes decide for themselves)
    assert prio({"WORLD_SIZE": "1"}) == (-1, None)
    assert prio({"WORLD_SIZE": "8"}) == (0, None)
    assert prio({"SLURM_NTASKS": "4"}) == (0, None)
    assert prio({"WORLD_SIZE": "1", "SLURM_NTASKS": "4"}) == (-1, None)
    assert prio({"WORLD_SIZE": "8", "NVRX_STREAM_PRIORITY": "high"}) == (-1, "high")
    assert prio({"NVRX_STREAM_PRIORITY": "normal"}) == (0, "normal")
    assert prio({"NVRX_STREAM_PRIORITY": "off"})[0] == 0


def test_children_of_a_process_that_registered_the_tracer_can_register_theirs(tmp_path):
    """``rocprofiler_force_configure`` leaves ``ROCPROFILER_REGISTER_FORCE_LOAD=1`` in the process environment.  A child
    that inherits it (a spawned rank, a DataLoader worker) loads and CONFIGURES rocprofiler-sdk the moment libamdhip64 is
    loaded -- the full tool search on ``import torch``, and "already configured" for its own tracer: on the MI355X box
    every multi-process test of the reference's suite died that way in per-kernel mode.  ``ktrace.release_env`` (called
    when the profiler initialises, i.e. once the runtime is up) takes the variable back."""
    import subprocess
    import sys

    script = tmp_path / "register.py"
    script.write_text(f"REPO = {REPO!r}\n" + r"""
import ctypes, os, subprocess, sys
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO]
os.environ["NVRX_GPU_TIMING"] = "kernels"
from nvrx_straggler import ktrace
libc = ctypes.CDLL(None)
libc.getenv.restype = ctypes.c_char_p
role, err = sys.argv[1], None
try:
    ktrace.timing_mode()
    ktrace.setup()
except Exception as e:
    err = str(e)
print("STATE", role, err, ktrace._setup_route, libc.getenv(b"ROCPROFILER_REGISTER_FORCE_LOAD"))
if role == "parent":
    ktrace.release_env()
    print("AFTER", libc.getenv(b"ROCPROFILER_REGISTER_FORCE_LOAD"), libc.getenv(b"GLOG_v"))
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True)   # inherits the C environment
    print(r.stdout, r.stderr[-500:])
""")
    env = {k: v for k, v in os.environ.items() if k not in ("NVRX_GPU_TIMING", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARIES")
           and not k.startswith("GLOG_")}
    p = subprocess.run([sys.executable, str(script), "parent"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith(("STATE", "AFTER"))]
    assert lines[0] == "STATE parent None force_configure b'1'", lines       # the SDK wrote it ...
    assert lines[1] == "AFTER None None", lines                              # ... release_env took it (and the GLOG switches) back
    assert lines[2].startswith("STATE child None force_configure"), lines    # the child registers its own tracer


def test_tool_search_guard_is_refused_while_another_thread_runs_and_the_sdks_own_route_is_taken(tmp_path):
    """The link-map guard of ``nvrx_ktrace_setup`` touches loader state, so the library applies it only while every OTHER
    thread of the process is asleep.  With a busy thread: ``NVRX_KTRACE_ERR_UNSAFE`` (-16), nothing registered, and
    ``ktrace.setup`` falls back to naming the library in ``ROCP_TOOL_LIBRARIES`` (the SDK's own, slower route) -- and
    ``release_env`` takes that variable back again once the runtime is up."""
    import subprocess
    import sys

    script = tmp_path / "busy.py"
    script.write_text(f"REPO = {REPO!r}\n" + r"""
import os, sys, threading, time
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO]
os.environ["NVRX_GPU_TIMING"] = "kernels"
os.environ["NVRX_KTRACE_AT_IMPORT"] = "0"
stop = []
def spin():
    while not stop:
        pass
t = threading.Thread(target=spin, daemon=True)
from nvrx_straggler import ktrace           # (imports torch: before the busy thread starts, or the import crawls)
lib = ktrace.load()
t.start(); time.sleep(0.05)
rc = lib.nvrx_ktrace_setup(0)
print("RC", rc, lib.nvrx_ktrace_last_error().decode()[:60], lib.nvrx_ktrace_hidden_libraries())
ktrace.setup()
print("ROUTE", ktrace._setup_route, ktrace.lib_path() in os.environ.get("ROCP_TOOL_LIBRARIES", ""))
ktrace.release_env()
print("AFTER", os.environ.get("ROCP_TOOL_LIBRARIES"))
stop.append(1)
""")
    env = {k: v for k, v in os.environ.items() if k not in ("NVRX_GPU_TIMING", "ROCP_TOOL_LIBRARIES", "NVRX_KTRACE_SCAN_GUARD")}
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout + p.stderr[-2000:]
    out = dict(l.split(" ", 1) for l in p.stdout.splitlines() if l.startswith(("RC", "ROUTE", "AFTER")))
    assert out["RC"].startswith("-16 ") and "other thread" in out["RC"] and out["RC"].endswith(" 0"), out
    assert out["ROUTE"] == "ROCP_TOOL_LIBRARIES True", out
    assert out["AFTER"] == "None", out
