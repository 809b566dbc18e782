"""Turns "report every T seconds" into "report every N iterations", the same N on every rank.

Behaviour of the reference's ``ReportIntervalTracker`` (interval_tracker.py:24-81): the first
``INTERVAL_ESTIMATION_ITERS`` (16) step times give a median (torch's LOWER median, f32); ``time_interval / median``
is MAX-all-reduced once so that all ranks agree (a second value of the detector's rides along, ``also_max``); the
result is floored at ``profiling_interval`` and truncated to int.  ``is_interval_elapsed`` is then ``current_iter % iter_interval == 0``.

Kept here as the monotonic-clock marks of the calls (one clock read per iteration, the step times are their
differences, taken once when the 17th mark is in); the one-off all-reduce travels on the group's own device (cuda
for RCCL, cpu for gloo) instead of the reference's unconditional ``torch.cuda.current_device()``, so the tracker
also works in CPU plumbing runs.
"""
from __future__ import annotations

import time
from typing import List, Optional

import torch

from . import dist_utils


class ReportIntervalTracker:
    #: step times the estimate is made of (class attribute in the reference too; its tests read it off the instance)
    INTERVAL_ESTIMATION_ITERS = 16

    def __init__(self, time_interval: float = 60.0, profiling_interval: int = 1):
        self.time_interval = time_interval
        self.profiling_interval = profiling_interval
        self.current_iter = 0
        self.iter_interval: Optional[int] = None
        self._marks: List[float] = []  # time.monotonic() of every iter_increase() until the estimate is made
        # A second number that rides on the one all-reduce of the estimate (MAX over the ranks): ``also_max()`` gives this
        # rank's value when the estimate is made, ``agreed_also`` holds the result.  ``Detector`` agrees its per-kernel
        # tracing interval this way -- EVERY rank takes part in this collective whatever its configuration, which a
        # collective of the detector's own, issued only by the ranks that calibrate, could not guarantee.
        self.also_max = None
        self.agreed_also: Optional[float] = None

    @property
    def step_times(self) -> List[float]:
        """Seconds between consecutive iterations seen so far; empty once the interval is known."""
        m = self._marks
        return [b - a for a, b in zip(m, m[1:])]

    def iter_increase(self) -> None:
        self.current_iter += 1
        if self.iter_interval is not None:
            return
        self._marks.append(time.monotonic())
        if len(self._marks) > self.INTERVAL_ESTIMATION_ITERS:
            self.iter_interval = self._agreed_interval(self.step_times)
            self._marks.clear()

    def _agreed_interval(self, steps: List[float]) -> int:
        """Iterations per ``time_interval`` by this rank's median step, the LARGEST such count over the ranks."""
        per_interval = (self.time_interval / torch.tensor(steps, dtype=torch.float32).median()).reshape(1)
        also = float(self.also_max()) if self.also_max is not None else 1.0
        both = torch.cat([per_interval, torch.tensor([also], dtype=torch.float32)])
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            both = both.to(dist_utils.get_device_for_backend(None))
            torch.distributed.all_reduce(both, op=torch.distributed.ReduceOp.MAX)
        per_interval, self.agreed_also = both.tolist()
        # reporting more often than sections are profiled makes no sense
        return int(max(per_interval, self.profiling_interval))

    def is_interval_elapsed(self) -> bool:
        n = self.iter_interval
        return n is not None and self.current_iter % n == 0
