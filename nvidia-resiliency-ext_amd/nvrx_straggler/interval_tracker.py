"""Turns "report every T seconds" into "report every N iterations", the same N on every rank.

Contract from the reference's ``ReportIntervalTracker`` (interval_tracker.py:24-81): the first
``INTERVAL_ESTIMATION_ITERS`` (16) measured step times give a median step time; ``time_interval /
median`` is MAX-all-reduced once so all ranks agree; the result is floored at ``profiling_interval``
and truncated to int.  ``is_interval_elapsed`` is then ``current_iter % iter_interval == 0``.
The one-off all-reduce travels on the group's own device (cuda for RCCL, cpu for gloo) instead of the
reference's unconditional ``torch.cuda.current_device()``, so the tracker also works in CPU plumbing runs.
"""
from __future__ import annotations

import dataclasses
import time
from typing import List, Optional

import torch

from . import dist_utils


@dataclasses.dataclass
class ReportIntervalTracker:
    INTERVAL_ESTIMATION_ITERS: int = 16
    time_interval: float = 60.0
    current_iter: int = 0
    iter_interval: Optional[int] = None
    prev_iter_start_time: Optional[float] = None
    step_times: List[float] = dataclasses.field(default_factory=list)
    profiling_interval: int = 1

    def _gather_report_interval(self) -> None:
        assert self.iter_interval is None, "Report iteration interval has already been gathered."
        median_step = torch.median(torch.tensor(self.step_times, dtype=torch.float32))
        wanted = (self.time_interval / median_step).reshape(1)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            wanted = wanted.to(dist_utils.get_device_for_backend(None))
            torch.distributed.all_reduce(wanted, op=torch.distributed.ReduceOp.MAX)
        # reporting more often than sections are profiled makes no sense
        self.iter_interval = int(max(wanted.item(), self.profiling_interval))

    def iter_increase(self) -> None:
        self.current_iter += 1
        if self.iter_interval is not None:
            return
        now = time.monotonic()
        if self.prev_iter_start_time is not None:
            self.step_times.append(now - self.prev_iter_start_time)
            if len(self.step_times) == self.INTERVAL_ESTIMATION_ITERS:
                self._gather_report_interval()
                self.step_times.clear()
        self.prev_iter_start_time = time.monotonic()

    def is_interval_elapsed(self) -> bool:
        return self.iter_interval is not None and self.current_iter % self.iter_interval == 0
