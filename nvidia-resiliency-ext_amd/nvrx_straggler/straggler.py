"""``Detector``: section timing + report orchestration, MI355X edition.

API kept from the reference (straggler.py): ``CallableId`` (:35-62), ``CustomSection`` (:66-83) and
the class-level, never-instantiated ``Detector`` with ``initialize / shutdown / detection_section /
generate_report / generate_report_if_interval_elapsed / is_interval_elapsed / wrap_callables /
restore_original_callables`` (:86-407), including its error behaviour (double initialize ->
AssertionError, use before initialize -> RuntimeError("Detector is not initialized."), instantiation
-> RuntimeError).

What changed underneath:

* a section's elapsed times are appended to a DEVICE ring (pinned staging + one scatter kernel per
  flush) instead of a Python deque, so a report never converts thousands of Python floats into a
  tensor (the reference's dominant cost, straggler.py:185);
* ``profile_cuda=True`` measures GPU time in one of two ways (``ktrace.timing_mode``): per KERNEL, by kernel name,
  through rocprofiler-sdk -- the reference's CUPTI data model, the default of multi-rank jobs, fed into the device rings
  by the tracer's own thread -- or per REGION, with two device-timestamp kernels around the section on the current stream
  (single-process default).  The ranks of a job agree on one of the two at their first collective report;
* ``generate_report`` waits only for what it needs -- the traced kernels of the window (``nvrx_ktrace_sync``) or the
  region stamps -- not ``torch.cuda.synchronize()``, then runs statistics -> all-gather -> scoring on the detector's own
  HIP stream; ``asynchronous=True`` does not even wait for that.
"""
from __future__ import annotations

import ctypes
import dataclasses
import functools
import inspect
import logging
import os
import socket
import sys
import time
from collections.abc import Callable
from contextlib import contextmanager
from typing import Any, Dict, Iterable, List, Optional, Sequence, Union

from . import _native
from . import backend as _backend_mod
from . import dist_utils as _dist_utils
from . import ktrace as _ktrace
from .cupti import CuptiManager
from .interval_tracker import ReportIntervalTracker
from .reporting import Report, ReportGenerator, _LiveBlock, _PendingBlock, _ScoreSource
from .statistics import Statistic  # noqa: F401  (re-exported for callers that poke summaries)

GPU_KEY_PREFIX = "hipevent::"  # kernel-summary key of a section's GPU-time row
_log = logging.getLogger(__name__)


@dataclasses.dataclass(frozen=True)
class CallableId:
    """Names a callable as (owner object, attribute name); ``str()`` gives its section name.

    The extra optional fields exist in the reference's dataclass and are accepted for compatibility.
    """

    obj: object
    name: str
    arg_filter_fn: Optional[Callable[[inspect.BoundArguments], bool]] = None
    extra_args_fn: Optional[Callable[[inspect.BoundArguments], dict]] = None
    ignored_args: Optional[tuple] = None

    def __str__(self) -> str:
        # naming rules of the reference (straggler.py:53-62): module name, qualified class name,
        # or the class name of an instance
        o = self.obj
        if inspect.ismodule(o):
            owner = o.__name__
        elif inspect.isclass(o):
            owner = f"{o.__module__}.{o.__name__}"
        elif hasattr(o, "__class__"):
            owner = getattr(o.__class__, "__name__", o)
        else:
            owner = getattr(o, "__name__", o)
        return f"{owner}.{self.name}"


class _ElapsedRing:
    """deque-like facade (append / extend / clear / len) over one device ring row."""

    __slots__ = ("_rings", "_row")

    def __init__(self, rings, row: int):
        self._rings = rings
        self._row = row

    def append(self, value: float) -> None:
        self._rings.push(self._row, value)

    def extend(self, values: Iterable[float]) -> None:
        self._rings.push_many(self._row, list(values) if not hasattr(values, "__len__") else values)

    def clear(self) -> None:
        self._rings.set_count(self._row, 0)

    def __len__(self) -> int:
        return self._rings.count(self._row)


class CustomSection:
    """A user-defined code section (``Detector.detection_section``).

    ``cpu_elapsed_times`` holds the wall-clock durations [ms] of profiled entries, newest
    ``max_elapseds_len`` only; here it is a view of a device ring row rather than a deque.
    ``CustomSection.max_elapseds_len`` (class attribute, default 8192 as in the reference,
    straggler.py:80) is read by ``Detector.initialize`` as the ring capacity.
    """

    max_elapseds_len: int = 8 * 1024

    __slots__ = ("name", "location", "total_entry_cnt", "row", "gpu_row", "cpu_elapsed_times")

    def __init__(self, name: str, location: str, rings=None):
        self.name = name
        self.location = location
        self.total_entry_cnt = 0
        rings = rings if rings is not None else Detector.rings
        self.row = rings.row_for(_native.KIND_SECTION, name)
        self.gpu_row: Optional[int] = None
        self.cpu_elapsed_times = _ElapsedRing(rings, self.row)


_MISS = object()  # a lane that cannot serve this report says so; the general path runs
_TRACED_DISPATCH_US = 1.0  # GPU time one traced dispatch costs (the SDK's completion signal + timestamps), see _trace_every_needed


class _Lane:
    """The steady-state report as ONE Python function around ONE C call (``nvrx_window_report``).

    ``Detector.generate_report`` is a dozen small methods of four objects -- profiler harvest, occupancy check, the
    generator's plan lookup, the one-call report, ring reset -- each cheap while the interpreter is warm.  At production
    cadence (a report per minute, a report per hundreds of steps) every one of them runs COLD after a second of training
    code: ~100 interpreter-level calls cost 150-190 us around a C call of 45 (profiles/r06a/r06c_kernels_mode_breakdown.txt).
    A lane is built after a report that took the cached plan; it holds everything that report looked up, checks with a few
    attribute reads that nothing it depends on has changed, and otherwise says ``_MISS`` BEFORE anything has happened, so the
    general path -- which stays the specification -- runs as if the lane did not exist."""

    trace = None

    __slots__ = ("rings", "reporter", "manager", "ext", "plan", "ws", "be", "view", "versions", "wr", "group", "wr_cache",
                 "rows_used", "desc_key", "window", "window_ref", "call", "ctx", "stream", "multi", "enqueue_only",
                 "returns_none", "stats_needed", "gather_on_rank0", "rank", "token", "asynchronous")

    @classmethod
    def build(cls, det) -> Optional["_Lane"]:
        rings, reporter, manager = det._rings, det.reporter, det._cupti_manager
        plan = reporter._ring_plan
        call = getattr(getattr(rings, "lib", None), "nvrx_window_report", None)
        if plan is None or not plan.fused or call is None or manager is None or not manager.is_initialized:
            return None
        multi = reporter.world_size > 1 and reporter._exchanged()
        if (multi and reporter._direct is None) or det._pending_region_switch is not None:
            return None  # (a host-driven exchange: the report is three calls with a collective between them)
        if reporter.asynchronous and not reporter.enqueue_only():
            return None
        ext = manager.cupti_ext
        ws = plan.ws
        if ws.block.desc_key is None or not ws.send_initialised:
            return None
        self = cls()
        self.rings, self.reporter, self.manager, self.ext, self.plan, self.ws = rings, reporter, manager, ext, plan, ws
        self.be = _backend_mod.get_backend()
        self.view = plan.view
        self.versions = (reporter.name_mapper.version, reporter._private_mapper.version)
        self.group, self.wr_cache = reporter.group, reporter._wr_cache
        self.wr = _dist_utils.world_and_rank(self.group, self.wr_cache)
        self.token = det._mode_agreed
        self.rows_used = rings.rows_used
        self.desc_key = ws.block.desc_key
        self.asynchronous = reporter.asynchronous
        self.enqueue_only = bool(reporter.asynchronous)
        self.multi = multi
        self.stats_needed = plan.stats_needed
        self.gather_on_rank0, self.rank = reporter.gather_on_rank0, reporter.rank
        self.returns_none = reporter.gather_on_rank0 and reporter.rank != 0
        w = self.window = _native.WindowDesc()
        if manager.per_kernel:
            if not getattr(ext, "_counting", False):
                return None  # (dispatches are not counted: a report has to synchronise the device first -- the general path does)
            cast, klib = ctypes.cast, ext._lib
            w.kt_sync = cast(klib.nvrx_ktrace_sync, ctypes.c_void_p).value
            w.kt_hold = cast(klib.nvrx_ktrace_hold, ctypes.c_void_p).value
            w.kt_counter = cast(klib.nvrx_ktrace_counter, ctypes.c_void_p).value
            w.kt_patience_s = float(ext.sync_patience_s)
            w.kt_rows_known, w.kt_keys_without_row = int(ext._rows_known), int(ext.keys_without_row)
        else:
            w.harvest_regions = 1
        w.rows_used = self.rows_used
        w.asynchronous = 1 if self.enqueue_only else 0
        self.window_ref = ctypes.byref(w)
        self.call, self.ctx, self.stream = call, rings.ctx, self.be._stream_handle
        return self

    def run(self, det):
        """The report, or ``_MISS`` (nothing has happened), or -- names incomplete on some rank -- the general path's report."""
        t0 = time.perf_counter_ns()
        tr = _Lane.trace  # (diagnostics, tools/cadence_kernels_breakdown.py: a list that collects clock marks, normally None)
        reporter, ws, rings = self.reporter, self.ws, self.rings
        # (what else a lane is built on -- the rings, the profiler, the agreed timing mode -- only changes through Detector
        #  methods that drop the lane: initialize, shutdown, _apply_pending_mode_switch)
        if (reporter._ring_plan is not self.plan or rings._rows_used != self.rows_used or reporter._resync_pending
                or reporter.asynchronous is not self.asynchronous
                or (reporter.name_mapper.version, reporter._private_mapper.version) != self.versions
                or _dist_utils.world_and_rank(self.group, self.wr_cache) != self.wr):
            return _MISS
        if self.enqueue_only:
            pend = reporter._inflight
            if pend is not None:
                # the previous asynchronous report: usually published long ago -- one look at its completion word (what
                # ReportGenerator._settle_inflight does behind a lock, a call into the library and numpy)
                blk = pend.blk
                words = getattr(blk, "meta_words", None)
                if words is not None and pend.blob is None and words[4] == pend.seq:
                    reporter._inflight = None
                    reporter._prev_async_settled = True
                    if self.multi:
                        reporter._check_exchange()
                    names_missing = words[0] != 1
                else:
                    names_missing = reporter._settle_inflight()
                if names_missing:
                    reporter._ring_plan = None       # that report's table carried an "ids missing" flag: every rank is
                    reporter._resync_pending = True  # heading for the name sync now, at the start of its general path
                    return _MISS
        if tr is not None:
            tr.append(("checks", time.perf_counter_ns() - t0))
        cur = ws._cur
        nxt = ws.blocks[1 - cur]
        if nxt.desc_key != self.desc_key:
            rings.configure_desc(ws, nxt, self.desc_key)  # (cold: this block has not run this shape yet)
        if nxt._live is not None:
            nxt.settle()  # (its report of two reports ago: copied out if somebody still holds it)
        ws._cur, ws.block = 1 - cur, nxt
        d = nxt.desc
        if self.enqueue_only:
            d.prev_settled = 1 if reporter._prev_async_settled else 0
        if self.multi:
            d.order_after_stream, d.order_after_enabled = self.be.current_stream_handle(), 1
        elif d.order_after_enabled:
            d.order_after_enabled = 0
        d.seq = ws.seq
        if tr is not None:
            tr.append(("settle_flip_desc", time.perf_counter_ns() - t0))
        rc = self.call(self.ctx, nxt.desc_ref, self.stream, self.window_ref)
        if tr is not None:
            tr.append(("c_call", time.perf_counter_ns() - t0))
        seq = ws.seq = d.seq
        if rc == _native.WINDOW_MISS:
            ws._cur, ws.block = cur, ws.blocks[cur]  # nothing ran on the block: it is not the current one
            return _MISS
        if rc < 0:
            reporter._ring_plan = None
            if rc == _native.ERR_TIMEOUT:
                self.be.retire_workspace(ws)
            _native.check(rc)
        if self.enqueue_only:
            reporter._prev_async_settled = False  # until somebody has seen THIS report complete
            pend = reporter._inflight = _PendingBlock(self.be, ws, seq)
            if self.returns_none:
                return None
            return Report._from_device(_ScoreSource(self.view, pend), reporter._shared_rank_to_node(),
                                       (time.perf_counter_ns() - t0) * 1e-6, self.gather_on_rank0, self.rank)
        nxt.mark_live(seq)  # (the statistics rows land under a completion word of their own: the block's next user waits for it)
        if self.multi:
            reporter._check_exchange()
        if rc == _native.WINDOW_NAMES:
            # some rank met a new name during this report's exchange: the rings still hold the window -- sync names, report again
            reporter._ring_plan = None
            reporter._resync_pending = True
            return det._report_and_reset(rings, reporter)
        if self.returns_none:
            return None
        live = _LiveBlock(self.be, ws, seq, self.stats_needed)
        nxt.attach(live)
        if tr is not None:
            tr.append(("live_block", time.perf_counter_ns() - t0))
        rep = Report._from_device(_ScoreSource(self.view, live), reporter._shared_rank_to_node(),
                                  (time.perf_counter_ns() - t0) * 1e-6, self.gather_on_rank0, self.rank)
        if tr is not None:
            tr.append(("report_object", time.perf_counter_ns() - t0))
        return rep


class _DeviceSideOnDemand(type):
    """``Detector.rings`` and ``Detector.cupti_manager`` -- everything that lives on a GPU -- come into being the first
    time they are touched, on the device that is current THEN.  The reference's own example initialises the detector
    before it selects its GPU (examples/straggler/example.py:60-66: ``Detector.initialize()``, then
    ``init_process_group``, then ``torch.cuda.set_device(local_rank)``); CUPTI does not care, device memory does: rings
    created inside ``initialize`` would sit on GPU 0 in every rank of such a script."""

    @property
    def rings(cls):
        if cls._rings is None and cls.initialized:
            cls._create_device_side()
        return cls._rings

    @rings.setter
    def rings(cls, value):
        cls._rings = value

    @property
    def cupti_manager(cls):
        if cls._cupti_manager is None and cls.initialized:
            cls._create_device_side()
        return cls._cupti_manager

    @cupti_manager.setter
    def cupti_manager(cls, value):
        cls._cupti_manager = value


class Detector(metaclass=_DeviceSideOnDemand):
    """Straggler detector; class-level singleton, not meant to be instantiated.

    Class attributes after ``initialize``: ``scores_to_compute``, ``gather_on_rank0``,
    ``profiling_interval``, ``custom_sections`` (name -> CustomSection), ``cupti_manager``,
    ``reporter`` (ReportGenerator), ``report_interval_tracker``, ``original_callables``, and ``rings``
    (the device ring buffers).
    """

    # configuration (set by initialize)
    initialized: bool = False
    scores_to_compute: Sequence[str]
    gather_on_rank0: bool
    profiling_interval: int
    report_time_interval: float
    # state (``rings``: device ring buffers, one row per section / GPU-timed region; ``cupti_manager``: the profiler that
    # feeds the GPU-time rows -- both created on first use, see _DeviceSideOnDemand)
    _rings: Any = None
    _cupti_manager: Optional[CuptiManager] = None
    _device_side_args: Any = None
    custom_sections: Dict[str, CustomSection]
    original_callables: Optional[Dict[CallableId, Any]]
    # collaborators
    reporter: ReportGenerator
    report_interval_tracker: ReportIntervalTracker
    # the name -> row tables of the rows that held samples at the last report (and the rings they were derived from)
    _occupied_key: Any = None
    _active_sections: Dict[str, int] = {}
    _active_kernels: Dict[str, int] = {}
    # the GPU-timing mode has been compared across the ranks of this process group (a token of the group, or None)
    _mode_agreed: Any = None
    # the job's common mode is region timing but this rank could not switch yet (a region was open): the reason, else None
    _pending_region_switch: Optional[str] = None
    _pending_switch_said: bool = False
    # per-kernel tracing budget (kernel_trace_budget_pct, see initialize and _calibration_mark): kernels are traced on every
    # (profiling_interval x _trace_every)-th entry of a profile_cuda section; while the cost is being measured _trace_gate
    # switches tracing on and off iteration by iteration
    kernel_trace_budget_pct: float = 0.0
    kernel_trace_cost_pct: Optional[float] = None   # what the calibration estimated (None: not measured)
    kernel_trace_dispatches: Optional[int] = None    # ... and how many dispatches a traced iteration had
    _trace_every: int = 1
    _trace_gate: bool = True
    _trace_sparse: bool = False                      # fast check in detection_section: gate closed or _trace_every > 1
    _calib: Any = None                               # (marks, flags) while measuring, else None
    # the steady-state report as one function around one C call (_Lane), built after a report that took the cached plan
    _lane: Optional[_Lane] = None
    _lanes_enabled: bool = True

    def __new__(cls):
        raise RuntimeError(f"class {cls.__name__} should not be instantiated")

    # ---- lifecycle -----------------------------------------------------------------------------
    @classmethod
    def initialize(
        cls,
        scores_to_compute: Union[Sequence[str], str] = "all",
        gather_on_rank0: bool = True,
        profiling_interval: int = 1,
        report_time_interval: float = 60,
        node_name: Optional[str] = None,
        max_rows: int = 256,
        asynchronous: Optional[bool] = None,
        kernel_trace_budget_pct: Optional[float] = None,
    ):
        """
        Args:
            scores_to_compute: list with 'relative_perf_scores' and/or 'individual_perf_scores', or "all".
            gather_on_rank0: rank 0's report covers all ranks (others get None); else per-rank reports.
            profiling_interval: profile every N-th entry of a section.
            report_time_interval: seconds between reports for ``generate_report_if_interval_elapsed``.
            node_name: name of this node in reports (default: ``socket.gethostname()``).
            max_rows: timing rows (sections + GPU-timed regions) the device rings can hold.
            asynchronous: ``generate_report`` only enqueues the report on the detector's HIP stream and returns a
                ``Report`` that waits for the device when it is first read, so the training loop never stalls on a
                report (see ``ReportGenerator``).  Default: the ``NVRX_ASYNC_REPORT`` environment variable, else
                False = the reference's synchronous behaviour.
            kernel_trace_budget_pct: per-kernel tracing only (``ktrace.timing_mode() == "kernels"``).  The share of a training
                iteration that tracing every kernel of the profiled sections may cost.  rocprofiler-sdk adds about a
                microsecond to every traced dispatch, which on a step of hundreds of small kernels is more than the
                reference's "< 1 %" (docs/source/straggler_det/usage_guide.rst:169).  With a budget, the first 16 iterations
                seen by ``generate_report_if_interval_elapsed`` -- the ones the interval tracker times anyway
                (interval_tracker.py:35,58-72) -- alternate between tracing and not tracing; if the traced ones are slower
                by more than the budget, kernels are traced on every (profiling_interval x N)-th entry only, N the smallest
                multiple that fits, the LARGEST N over the ranks (one all-reduce), logged once.  Section wall times are
                still recorded at ``profiling_interval``.  Default: ``NVRX_KTRACE_BUDGET_PCT``, else 1.0; 0 = trace at
                ``profiling_interval`` whatever it costs (the reference's behaviour).  Jobs that call ``generate_report``
                themselves never calibrate.
        """
        assert not cls.initialized
        _backend_mod.require_engine()  # no silent CPU path: a box that cannot run the engine says so here
        everything = str(scores_to_compute) == "all"
        cls.scores_to_compute = ["relative_perf_scores", "individual_perf_scores"] if everything else scores_to_compute
        cls.gather_on_rank0, cls.profiling_interval = gather_on_rank0, profiling_interval
        cls.report_time_interval = report_time_interval
        cls.custom_sections, cls.original_callables, cls._occupied_key = {}, {}, None

        # device side -- the rings every section / GPU-timed region records into and the profiler that feeds them --
        # is created on first use, on the device that is current then (_DeviceSideOnDemand)
        capacity = int(CustomSection.max_elapseds_len)
        per_kernel = _ktrace.timing_mode() == "kernels"
        if per_kernel and int(max_rows) == 256:
            max_rows = 4096  # one row per distinct kernel key; 4096 x 8192 f32 = 128 MB of 288 GB
        cls._rings = cls._cupti_manager = cls._lane = None
        cls._device_side_args = (int(max_rows), capacity)
        cls._mode_agreed = None
        cls._pending_region_switch, cls._pending_switch_said = None, False
        if kernel_trace_budget_pct is None:
            try:
                kernel_trace_budget_pct = float(os.environ.get("NVRX_KTRACE_BUDGET_PCT", "1.0"))
            except ValueError:
                kernel_trace_budget_pct = 1.0
        cls.kernel_trace_budget_pct = max(0.0, float(kernel_trace_budget_pct))
        cls.kernel_trace_cost_pct = None
        cls._trace_every, cls._trace_gate, cls._trace_sparse = 1, True, False
        cls._calib = ([], [], []) if (per_kernel and cls.kernel_trace_budget_pct > 0.0) else None
        cls.kernel_trace_dispatches = None
        _log.info("nvrx straggler: GPU time of profile_cuda sections is measured per %s (mode '%s': %s)",
                  "kernel, by kernel name" if per_kernel else "profiled region", _ktrace.timing_mode(), _ktrace.mode_note())

        # host side: who scores, and when
        if asynchronous is None:
            asynchronous = os.environ.get("NVRX_ASYNC_REPORT", "0") not in ("", "0")
        cls.reporter = ReportGenerator(scores_to_compute=cls.scores_to_compute, gather_on_rank0=gather_on_rank0,
                                       node_name=node_name or socket.gethostname(), asynchronous=asynchronous)
        cls.report_interval_tracker = ReportIntervalTracker(time_interval=report_time_interval,
                                                            profiling_interval=profiling_interval)
        cls.report_interval_tracker.also_max = cls._trace_every_needed  # (the tracing budget's number rides on the tracker's all-reduce)
        cls.initialized = True

    @classmethod
    def _create_device_side(cls) -> None:
        if cls._rings is not None:
            return
        max_rows, capacity = cls._device_side_args
        rings = _backend_mod.get_backend().make_rings(1, max_rows, capacity)
        try:
            manager = CuptiManager(statsMaxLenPerKernel=capacity, rings=rings)
            manager.initialize()
        except BaseException:
            rings.close()
            raise
        cls._rings, cls._cupti_manager = rings, manager

    @classmethod
    def _agree_timing_mode(cls, group) -> None:
        """Once per process group, at the first collective report: the ranks compare how they measure GPU time (one MIN
        all-reduce of a mode code).  Per-kernel keys and per-region keys share no names, so a job in which ONE rank fell
        back to region timing (HIP was up before the import, the tracer could not register, ...) would get NaN for every
        relative GPU score without a word: instead every rank drops to the common mode and says so."""
        import torch
        import torch.distributed as dist

        from . import dist_utils

        mine = _ktrace.mode_code()
        t = torch.tensor([mine, -mine], dtype=torch.int32, device=dist_utils.get_device_for_backend(group))
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)  # (min, -max) in one collective
        common, highest = int(t[0].item()), -int(t[1].item())
        if common == highest:
            _log.info("nvrx straggler: every rank measures GPU time in mode '%s'", _ktrace.timing_mode())
            return
        if common == mine:
            _log.info("nvrx straggler: other ranks of this job trace kernels by name, this one times GPU work per profiled region "
                      "(%s): they fall back to region timing as well", _ktrace.mode_note())
            return
        why = ("another rank of this job cannot trace kernels (its HIP runtime was up before nvrx_straggler was imported, or "
               "the tracer could not register): all ranks time GPU work per profiled region")
        # The agreement is ONE collective per group and is never repeated by a single rank (its peers hold the token and would
        # pair a lone all-reduce with whatever collective they issue next).  What may have to wait is only this rank's own
        # switch: a profiled region is open right now (a report called inside a section) -- it is applied, without any
        # collective, as soon as a report finds no region open (_apply_pending_mode_switch).
        cls._pending_region_switch = why
        cls._apply_pending_mode_switch()

    @classmethod
    def _apply_pending_mode_switch(cls) -> None:
        why = cls._pending_region_switch
        if why is None:
            return
        cls._lane = None  # (the profiler underneath is about to change)
        if cls.cupti_manager.switch_to_regions():
            cls._pending_region_switch = None
            cls._calib, cls._trace_every, cls._trace_gate, cls._trace_sparse = None, 1, True, False  # (region stamps cost next to nothing)
            _ktrace.fall_back_to_regions(why)
            _log.warning("nvrx straggler: %s. GPU scores of THIS report window mix kernel keys and region keys and may be NaN; "
                         "later windows are consistent. Collectives inside profile_cuda sections now count into the region's "
                         "time: keep them outside, or fix the slow-starting rank (NVRX_GPU_TIMING=kernels makes it an error).", why)
        elif not cls._pending_switch_said:
            cls._pending_switch_said = True
            _log.warning("nvrx straggler: %s -- but a profiled region is open on this rank right now (generate_report was called "
                         "inside a detection_section): this rank switches at the first report that finds none open; until then its "
                         "GPU scores are NaN on the other ranks' reports.", why)

    @classmethod
    def shutdown(cls):
        """Undo ``initialize``: wrapped callables get their originals back, profiler, rings and exchange route close."""
        cls._lane = None
        manager, cls._cupti_manager = cls._cupti_manager, None
        if manager is not None:
            manager.shutdown()
        cls.restore_original_callables()
        rings, cls._rings = cls._rings, None
        if rings is not None:
            rings.close()
        reporter = getattr(cls, "reporter", None)
        if reporter is not None:
            reporter.close()
        cls.initialized = False

    # ---- system-side context (not in the reference) --------------------------------------------------
    @classmethod
    def gpu_telemetry(cls) -> dict:
        """Clocks / temperatures / power / utilisation of this rank's GPU from ROCm SMI (``gpu_telemetry.sample``):
        what to look at next to a low individual GPU score.  Raises if ROCm SMI is unavailable."""
        from . import gpu_telemetry

        return gpu_telemetry.sample(_backend_mod.get_backend().device.index)

    @classmethod
    def gpu_telemetry_line(cls) -> str:
        """The same as one log line; never raises."""
        from . import gpu_telemetry

        try:
            index = _backend_mod.get_backend().device.index
        except Exception as e:  # noqa: BLE001
            return f"gpu telemetry unavailable: {e}"
        return gpu_telemetry.describe(index)

    # ---- summaries (host-visible form; the report path itself keeps them on the device) -------------
    @classmethod
    def _get_section_summaries(cls):
        """name -> {Statistic: value} for every section holding samples (computed on the device)."""
        stats = cls.rings.peek_stats()
        out = {}
        for name, section in cls.custom_sections.items():
            if len(section.cpu_elapsed_times) == 0:
                continue
            v = stats[section.row]
            out[name] = {
                Statistic.MIN: float(v[0]), Statistic.MAX: float(v[1]), Statistic.MED: float(v[2]),
                Statistic.AVG: float(v[3]), Statistic.STD: float(v[4]), Statistic.NUM: int(v[5]),
            }
        return out

    @classmethod
    def _get_kernel_summaries(cls):
        """key -> {Statistic: value} for every GPU-timed region (hipEvent rows)."""
        out = {}
        for key, ks in cls.cupti_manager.get_results().items():
            out[key] = {
                Statistic.MIN: ks.min, Statistic.MAX: ks.max, Statistic.MED: ks.median,
                Statistic.AVG: ks.avg, Statistic.STD: ks.stddev, Statistic.NUM: ks.num_calls,
            }
        return out

    @classmethod
    def _reset_sections_elapseds(cls):
        cls.rings.reset()

    # ---- reports -----------------------------------------------------------------------------------
    @classmethod
    def generate_report(cls):
        """Score everything recorded since the last report, then empty the rings.  Collective."""
        assert cls.initialized
        lane = cls._lane
        if lane is not None:
            out = lane.run(cls)
            if out is not _MISS:
                return out
            cls._lane = None  # something it was built on has changed: the general path, and a new lane after it
        report = cls._generate_report_general()
        if cls._lane is None and cls._lanes_enabled:
            cls._lane = _Lane.build(cls)
        return report

    @classmethod
    def _generate_report_general(cls):
        rings = cls.rings
        reporter = cls.reporter
        manager = cls.cupti_manager
        if cls._pending_region_switch is not None:
            cls._apply_pending_mode_switch()
        if reporter.world_size > 1 or cls._mode_agreed is None:
            from . import dist_utils

            world, _ = dist_utils.world_and_rank(reporter.group, reporter._wr_cache)
            token = (world, id(reporter.group)) if world > 1 else "single"
            if token != cls._mode_agreed:
                cls._mode_agreed = token
                if world > 1:
                    cls._agree_timing_mode(reporter.group)
        # the recorded GPU regions must have finished; nothing else on the device is waited for.  Per-kernel tracing:
        # every kernel enqueued inside a section so far has finished and its duration is in its ring (one C call; the
        # tracer's thread has been appending them all along) -- an ASYNCHRONOUS report does not wait even for that:
        # durations that arrive later count in the next window
        if manager.per_kernel and reporter.enqueue_only():
            # ... and what arrives from here to the ring reset below stays with the tracer's thread until then
            manager.cupti_ext.hold(True)
            try:
                manager.harvest(wait=False)
                return cls._report_and_reset(rings, reporter)
            finally:
                manager.cupti_ext.hold(False)
        manager.harvest(wait=True)
        return cls._report_and_reset(rings, reporter)

    @classmethod
    def _report_and_reset(cls, rings, reporter):
        # which rows hold samples this window (one C call); the name tables are rebuilt only when
        # that set changes, so a steady-state report does no per-section Python work
        if rings.occupancy_changed() or cls._occupied_key is not rings:
            counts = rings.counts()
            cls._active_sections = {n: sec.row for n, sec in cls.custom_sections.items() if counts[sec.row] > 0}
            cls._active_kernels = {k: row for k, row in rings.kernel_row_names.items() if counts[row] > 0}
            cls._occupied_key = rings  # (the tables belong to these rings: a re-initialised Detector starts over)
        order_after = _backend_mod.get_backend().current_stream_handle() if reporter.world_size > 1 else None
        report = reporter.generate_report_from_rings(rings, cls._active_sections, cls._active_kernels,
                                                     order_after=order_after)
        keep = reporter.take_unreported_rows() if reporter.asynchronous else None  # (empty unless the report was enqueued on old tables)
        if keep:
            # an asynchronous report that met names it has no ids for ran on the old tables (the ranks sync names at the next
            # report): the samples of those rows were in nobody's report and stay for the next window
            counts = rings.counts()
            keep = [(row, int(counts[row])) for row in keep]
        rings.reset()  # both the section rows and the GPU-time rows, like :241-242 of the reference
        if keep:
            for row, n in keep:
                rings.set_count(row, n)
        return report

    @classmethod
    def generate_report_if_interval_elapsed(cls):
        """Call once per training iteration on every rank; reports when the (rank-synchronised)
        iteration interval has elapsed, otherwise returns None."""
        assert cls.initialized
        tracker = cls.report_interval_tracker
        if cls._calib is not None:
            cls._calibration_mark()
        tracker.iter_increase()
        if cls._calib is not None and tracker.iter_interval is not None:
            cls._calibration_done(tracker.agreed_also)
        return cls.generate_report() if tracker.is_interval_elapsed() else None

    # ---- per-kernel tracing budget -----------------------------------------------------------------
    _MAX_TRACE_EVERY = 64

    @classmethod
    def _calibration_mark(cls) -> None:
        """One call per training iteration while the cost of tracing is being measured -- the interval tracker's 16 timed
        iterations: iterations alternate between tracing the kernels of their profiled sections and not tracing, each one's
        wall time is the distance between two of these calls."""
        marks, flags, enqueued = cls._calib
        # The iteration's DEVICE work has to be inside its wall time: a GPU-bound loop enqueues a step in a fraction of the time the
        # GPU takes for it, and tracing doubles what a launch costs the HOST -- without the wait the host-side times say "tracing
        # costs 266 %" of a loop it slows down by 1.2 % (profiles/r06d_budget_calibration.txt).  Seventeen device
        # synchronisations at the start of a job; the reference synchronises at every report (straggler.py:234).
        try:
            import torch

            if torch.cuda.is_initialized() and torch.cuda.is_available():  # (is_available() re-probes a device-less host: 0.6 s per call)
                torch.cuda.synchronize()
        except RuntimeError:  # (e.g. a stream is being captured into a graph right now: no calibration then)
            cls._calib = None
            cls._trace_gate, cls._trace_sparse = True, cls._trace_every > 1
            return
        marks.append(time.monotonic())
        try:
            enqueued.append(int(cls._cupti_manager.cupti_ext._lib.nvrx_ktrace_counter(0)))  # traced dispatches so far (process-wide)
        except Exception:  # noqa: BLE001  (no tracer underneath: the A/B of the wall times alone decides)
            enqueued.append(0)
        on = len(marks) % 2 == 1  # the iteration that starts now: traced after marks 1, 3, 5 ...
        flags.append(on)
        cls._trace_gate = on
        cls._trace_sparse = (not on) or cls._trace_every > 1

    @classmethod
    def _trace_every_needed(cls) -> float:
        """What the interval tracker's all-reduce carries for this rank (``ReportIntervalTracker.also_max``): the multiple of
        ``profiling_interval`` at which tracing fits the budget, from the medians of the traced and the untraced iterations."""
        if cls._calib is None:
            return 1.0
        marks, flags, enqueued = cls._calib
        steps = [b - a for a, b in zip(marks, marks[1:])]
        t_on = sorted(t for t, f in zip(steps, flags) if f)
        t_off = sorted(t for t, f in zip(steps, flags) if not f)
        traced = sorted(b - a for a, b, f in zip(enqueued, enqueued[1:], flags) if f)
        cls.kernel_trace_cost_pct = None
        if not t_on or not t_off:
            return 1.0
        if cls.profiling_interval == 1:
            m_on, m_off = t_on[(len(t_on) - 1) // 2], t_off[(len(t_off) - 1) // 2]  # lower medians, as the tracker's
        else:
            # only every profiling_interval-th entry is traced at all, so only some of the "traced" iterations carry the cost:
            # the average is what a step pays (less robust than a median; the default interval is 1)
            m_on, m_off = sum(t_on) / len(t_on), sum(t_off) / len(t_off)
        if m_off <= 0.0:
            return 1.0
        cost = (m_on - m_off) / m_off * 100.0
        # Second estimate, from what CAN be measured exactly in 16 iterations: the number of dispatches a traced iteration
        # had, times what one traced dispatch costs the GPU.  That cost is rocprofiler-sdk's completion signal and timestamps:
        # 0.3-1.6 us per dispatch on the eleven boxes of rounds 5-6 (docs/MEASUREMENTS.md), 1.0 us assumed.  The A/B above
        # resolves a launch-bound loop's overhead (tracing doubles what a launch costs the host: tens of per cent) but not one
        # per cent of a GPU-bound step from eight iterations each way (a box where the paired A/B of 140 steps says 1.4 %
        # measured 0.03 % here, profiles/r06y_bench_driver.json); the larger estimate decides.
        if traced:
            n = traced[(len(traced) - 1) // 2]
            cls.kernel_trace_dispatches = int(n)
            model = n * _TRACED_DISPATCH_US * 1e-6 / m_off * 100.0
            if cls.profiling_interval > 1:
                model = sum(traced) / len(traced) * _TRACED_DISPATCH_US * 1e-6 / m_off * 100.0
            cost = max(cost, model)
        cls.kernel_trace_cost_pct = cost
        if cost <= cls.kernel_trace_budget_pct:
            return 1.0
        return float(min(cls._MAX_TRACE_EVERY, int(-(-cost // cls.kernel_trace_budget_pct))))

    @classmethod
    def _calibration_done(cls, agreed: Optional[float]) -> None:
        """The ranks take the LARGEST multiple any of them needs (kernel summaries are compared across ranks: they should
        cover the same share of the entries); it came back with the interval tracker's own all-reduce."""
        mine = int(cls._trace_every_needed())
        cost, n = cls.kernel_trace_cost_pct, len(cls._calib[0]) - 1
        disp = cls.kernel_trace_dispatches
        cls._calib = None
        every = max(1, min(cls._MAX_TRACE_EVERY, int(agreed))) if agreed else mine
        cls._trace_every, cls._trace_gate = every, True
        cls._trace_sparse = every > 1
        total = cls.profiling_interval * every
        _log.info("nvrx straggler: tracing the kernels of the profiled sections costs this rank %s of an iteration at "
                  "profiling_interval=%d (the larger of: %s of %d iterations with and %d without; %s traced dispatches x 1 us); budget %.2f %%: kernels are traced on every "
                  "%d%s entry of a profile_cuda section%s", "%.2f %%" % cost if cost is not None else "an unmeasured share",
                  cls.profiling_interval, "medians" if cls.profiling_interval == 1 else "means", (n + 1) // 2, n // 2,
                  "?" if disp is None else disp, cls.kernel_trace_budget_pct, total,
                  {1: "st", 2: "nd", 3: "rd"}.get(total if total < 20 else total % 10, "th"),
                  "" if every == mine else " (another rank needed the larger interval)")

    @classmethod
    def is_interval_elapsed(cls) -> bool:
        return cls.report_interval_tracker.is_interval_elapsed()

    # ---- sections ----------------------------------------------------------------------------------
    @staticmethod
    def _get_this_context_block_location() -> str:
        # frames: this function <- detection_section generator <- contextmanager.__enter__ <- user code
        frame = sys._getframe(3)
        return f"{frame.f_code.co_filename}:{frame.f_lineno}"

    @classmethod
    def _ensure_section_name_is_valid(cls, name, location):
        known = cls.custom_sections.get(name)
        if known is not None and known.location != location:
            raise ValueError(f"Section name '{name}' is already used at: {known.location}")

    @classmethod
    @contextmanager
    def detection_section(cls, name: Optional[str] = None, profile_cuda: bool = True):
        """Time the enclosed block.

        Args:
            name: section name for reports (default: ``file:line`` of the ``with`` statement).
            profile_cuda: also measure the block's GPU time with a hipEvent pair on the current
                stream; it feeds the rank's GPU performance score.
        """
        if not cls.initialized:
            raise RuntimeError("Detector is not initialized.")

        section = cls.custom_sections.get(name) if name is not None else None
        if section is None:
            location = cls._get_this_context_block_location()
            if name is None:
                name = location
            section = cls.custom_sections.get(name)
            if section is None:
                section = CustomSection(name=name, location=location, rings=cls.rings)
                cls.custom_sections[name] = section

        profiled = (section.total_entry_cnt % cls.profiling_interval) == 0
        section.total_entry_cnt += 1
        if not profiled:
            yield
            return

        if profile_cuda and cls._trace_sparse:
            # the tracing budget (initialize: kernel_trace_budget_pct): this entry's wall time is recorded, its kernels are traced
            # only on every (profiling_interval x _trace_every)-th entry -- and not at all in the untraced calibration iterations
            profile_cuda = cls._trace_gate and ((section.total_entry_cnt - 1) // cls.profiling_interval) % cls._trace_every == 0
        if profile_cuda:
            cls.cupti_manager.start_profiling(GPU_KEY_PREFIX + name)
        t0 = time.perf_counter_ns()
        try:
            yield
        except BaseException:
            # no sample for an entry that raised; just close the GPU region
            if profile_cuda:
                cls.cupti_manager.stop_profiling()
            raise
        elapsed_ms = (time.perf_counter_ns() - t0) * 1e-6
        # with device timestamps the kernel that closes the GPU region also appends this sample
        if not (profile_cuda and cls.cupti_manager.stop_profiling(section.row, elapsed_ms)):
            cls.rings.push(section.row, elapsed_ms)

    # ---- callable wrapping -------------------------------------------------------------------------
    @classmethod
    def _build_wrapper(cls, fn, callable_id, profile_cuda: bool = True):
        """``fn`` run inside the section named after ``callable_id``."""
        section = functools.partial(cls.detection_section, name=str(callable_id), profile_cuda=profile_cuda)

        @functools.wraps(fn)
        def timed(*args, **kwargs):
            with section():
                return fn(*args, **kwargs)

        return timed

    @classmethod
    def wrap_callables(cls, callable_ids: List[CallableId], profile_cuda: bool = True):
        """Replace each ``getattr(cid.obj, cid.name)`` by a version that runs inside
        ``detection_section(str(cid))``."""
        cls.original_callables = {cid: getattr(cid.obj, cid.name) for cid in callable_ids}
        for cid, original in cls.original_callables.items():
            setattr(cid.obj, cid.name, cls._build_wrapper(original, cid, profile_cuda=profile_cuda))

    @classmethod
    def restore_original_callables(cls):
        for cid, original in (cls.original_callables or {}).items():
            setattr(cid.obj, cid.name, original)
