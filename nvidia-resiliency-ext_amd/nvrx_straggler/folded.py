"""Several logical ranks per GPU: the harness used to run a whole R-rank job on fewer GPUs.

BASELINE.json quotes the metric on 8 ranks x 64 sections x 10 000 samples and asks for it at 1, 2, 4
and 8 GPUs.  With fewer than 8 GPUs each process holds ``R / world_size`` logical ranks' timing
matrices in one set of device rings (``local_ranks`` of ``nvrx_ctx_create``); the statistics kernel
then covers ``local_ranks * S`` rows per launch, the all-gather carries ``local_ranks`` exchange rows
per process, and the score kernel sees the same ``[R, L]`` table as an 8-GPU run.  Production use is
always one logical rank per GPU (``Detector``); this class exists for ``bench.py`` and the
full-size parity tests.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _native, dist_utils
from .backend import get_backend
from .reporting import ReportGenerator


class FoldedJob:
    def __init__(self, total_ranks: int = 8, section_names: Optional[Sequence[str]] = None, sections: int = 64,
                 ring_cap: int = 8192, scores_to_compute=("relative_perf_scores", "individual_perf_scores"),
                 gather_on_rank0: bool = True, pg=None, node_name: str = "node"):
        world = dist_utils.get_world_size(pg)
        if total_ranks % world:
            raise ValueError(f"total_ranks {total_ranks} must be a multiple of the world size {world}")
        self.total_ranks = total_ranks
        self.local_ranks = total_ranks // world
        self.world_rank = dist_utils.get_rank(pg)
        self.section_names = list(section_names) if section_names is not None else [f"section_{s:03d}" for s in range(sections)]
        self.backend = get_backend()
        self.rings = self.backend.make_rings(self.local_ranks, len(self.section_names), ring_cap)
        self.ring_cap = ring_cap
        self.reporter = ReportGenerator(list(scores_to_compute), gather_on_rank0=gather_on_rank0, pg=pg, node_name=node_name)
        self.rows = {name: self.rings.row_for(_native.KIND_SECTION, name) for name in self.section_names}
        self._no_kernel_rows = {}  # same object every report so the reporter's cached plan stays valid

    def logical_ranks(self):
        """Global logical ranks held by this process."""
        return range(self.world_rank * self.local_ranks, (self.world_rank + 1) * self.local_ranks)

    def load(self, lr: int, samples) -> None:
        """Append ``samples`` ([S, n] f32, host array or device tensor) to logical rank ``lr``'s rings."""
        if isinstance(samples, np.ndarray):
            samples = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float32)).to(self.backend.device)
        self.backend.synchronize()
        torch.cuda.current_stream().synchronize()
        rows = [self.rows[name] for name in self.section_names]
        if rows == list(range(rows[0], rows[0] + len(rows))) and samples.dim() == 2 and samples.stride(1) == 1:
            self.rings.push_device_rows(rows[0], samples, lr=lr)     # the whole matrix in one call
        else:
            for s, row in enumerate(rows):
                self.rings.push_device(row, samples[s].contiguous(), lr=lr)
        self.backend.synchronize()

    def rearm(self, n: int) -> None:
        """Declare every row holds ``n`` resident samples again (after a report emptied the rings)."""
        self.rings.set_count_all(n)

    def report(self, reset: bool = True):
        rep = self.reporter.generate_report_from_rings(self.rings, self.rows, self._no_kernel_rows,
                                                       local_ranks=self.local_ranks)
        if reset:
            self.rings.reset()
        return rep

    def close(self) -> None:
        self.reporter.close()
        self.rings.close()
