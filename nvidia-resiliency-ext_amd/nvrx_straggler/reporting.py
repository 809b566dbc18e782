"""Performance report of the straggler detector: cross-rank exchange, on-device scoring, thresholds.

Public surface kept from the reference (reporting.py): ``StragglerId`` (:31-38), ``Report`` with its
10 fields (:41-82) and ``identify_stragglers`` (:84-151), and ``ReportGenerator(scores_to_compute,
gather_on_rank0=True, pg=None, node_name='<notset>').generate_report(section_summaries,
kernel_summaries)`` (:154-554).

What is different underneath (MI355X-first):

* a report is ONE fixed-length all-gather of every rank's exchange row -- medians, running minima,
  kernel weights and a "names complete" flag -- instead of the reference's all_reduce(MIN) flag +
  all_reduce(MIN) medians + gather of scores (three collectives with host<->device copies around
  each);
* every score for every rank, and the below-threshold flags, come from one HIP kernel over the
  gathered table (``nvrx_score``); the host never loops over kernels or sections;
* one small D2H copy brings back scores + flags + local statistics; ``Report`` exposes them through
  lazy mapping views so nothing is converted to Python objects until it is read.

Scores therefore pass through f32 (as the reference's gathered scores do, reporting.py:352).
"""
from __future__ import annotations

import collections
import dataclasses
import time
from collections.abc import Mapping as _MappingABC
from typing import Any, Dict, Iterator, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import backend as _backend_mod
from . import dist_utils
from .name_mapper import NameMapper
from .statistics import STAT_COLUMNS, Statistic

_SummaryType = Mapping[Statistic, float]

_NCCL_MARKER = "ncclDev"  # RCCL's device kernels carry the same prefix (reporting.py:336)


@dataclasses.dataclass(frozen=True)
class StragglerId:
    """Identity of a flagged rank: global rank and the node it runs on."""

    rank: int
    node: str


# --------------------------------------------------------------------------------------------------
# lazy views over the score arrays
# --------------------------------------------------------------------------------------------------
def _resolve(x):
    """Arrays may be handed to the views as zero-argument callables and are materialised on first use."""
    return x() if callable(x) else x


class RankScores(_MappingABC):
    """``rank -> score`` view over one column of the score array."""

    __slots__ = ("_ranks", "_src")

    def __init__(self, ranks: Sequence[int], values):
        self._ranks = ranks
        self._src = values  # ndarray, or callable returning it

    @property
    def _values(self) -> np.ndarray:
        v = self._src
        if callable(v):
            v = self._src = v()
        return v

    def __getitem__(self, rank: int) -> float:
        try:
            return float(self._values[self._ranks.index(rank)])
        except ValueError:
            raise KeyError(rank) from None

    def __iter__(self) -> Iterator[int]:
        return iter(self._ranks)

    def __len__(self) -> int:
        return len(self._ranks)

    def __repr__(self) -> str:
        return repr(dict(self.items()))

    def __reduce__(self):
        return (dict, (dict(self.items()),))

    def below(self, threshold: float) -> List[int]:
        """Ranks whose score is strictly below ``threshold`` (NaN never qualifies)."""
        with np.errstate(invalid="ignore"):
            hit = np.nonzero(self._values.astype(np.float64) < threshold)[0]
        return [self._ranks[int(i)] for i in hit]


class SectionScores(_MappingABC):
    """``section name -> (rank -> score)`` view over a [ranks, sections] block of the score array."""

    __slots__ = ("_names", "_cols", "_ranks", "_src")

    def __init__(self, names: Sequence[str], cols, ranks: Sequence[int], block):
        self._names = names
        self._cols = cols if isinstance(cols, dict) else {n: c for n, c in zip(names, cols)}
        self._ranks = ranks
        self._src = block  # ndarray [ranks, S], or callable returning it

    @property
    def _block(self) -> np.ndarray:
        b = self._src
        if callable(b):
            b = self._src = b()
        return b

    def __getitem__(self, name: str) -> RankScores:
        return RankScores(self._ranks, self._block[:, self._cols[name]])

    def __iter__(self) -> Iterator[str]:
        return iter(self._names)

    def __len__(self) -> int:
        return len(self._names)

    def __repr__(self) -> str:
        return repr({k: dict(v.items()) for k, v in self.items()})

    def __reduce__(self):
        return (dict, ({k: dict(v.items()) for k, v in self.items()},))


class StatSummaries(_MappingABC):
    """``name -> {Statistic: value}`` built from device statistics rows on first access."""

    __slots__ = ("_rows", "_src", "_cache")

    def __init__(self, rows: Mapping[str, int], stats):
        self._rows = rows  # name -> row index, only rows that hold samples (treated as immutable)
        self._src = stats  # ndarray [rows, 8], or callable returning it
        self._cache: Optional[Dict[str, Dict[Statistic, Any]]] = None

    def _materialise(self) -> Dict[str, Dict[Statistic, Any]]:
        if self._cache is None:
            stats = _resolve(self._src)
            out = {}
            for name, row in self._rows.items():
                vals = stats[row]
                d = {stat: float(vals[col]) for stat, col in STAT_COLUMNS}
                d[Statistic.NUM] = int(vals[5])
                out[name] = d
            self._cache = out
        return self._cache

    def __getitem__(self, name: str):
        return self._materialise()[name]

    def __iter__(self):
        return iter(self._rows)

    def __len__(self) -> int:
        return len(self._rows)

    def __repr__(self) -> str:
        return repr(self._materialise())

    def __reduce__(self):
        return (dict, (self._materialise(),))


@dataclasses.dataclass(frozen=True)
class Report:
    """Result of one ``generate_report`` call.

    Two score families, both "current performance / reference performance" in (0, 1]:

    * relative -- reference is the fastest rank's median for the same section / kernel
      (needs the cross-rank exchange);
    * individual -- reference is this rank's own best median so far.

    With ``gather_on_rank0=True`` the score mappings cover every rank and exist on rank 0 only;
    otherwise each rank's report covers just that rank.  Mappings may be empty.

    Fields (same names and order as the reference, reporting.py:73-82):
    ``gpu_relative_perf_scores`` rank -> score; ``section_relative_perf_scores`` section -> rank ->
    score; ``gpu_individual_perf_scores``; ``section_individual_perf_scores``; ``rank_to_node``;
    ``local_section_summaries`` / ``local_kernel_summaries`` this rank's timing statistics;
    ``generate_report_elapsed_time`` [ms]; ``gather_on_rank0``; ``rank``.
    """

    gpu_relative_perf_scores: Mapping[int, float]
    section_relative_perf_scores: Mapping[str, Mapping[int, float]]
    gpu_individual_perf_scores: Mapping[int, float]
    section_individual_perf_scores: Mapping[str, Mapping[int, float]]
    rank_to_node: Mapping[int, str]
    local_section_summaries: Mapping[str, Any]
    local_kernel_summaries: Mapping[str, Any]
    generate_report_elapsed_time: float
    gather_on_rank0: bool
    rank: Optional[int]

    def _ids(self, ranks) -> set:
        return {StragglerId(rank=r, node=self.rank_to_node[r]) for r in ranks}

    @staticmethod
    def _below(scores: Mapping[int, float], threshold: float):
        if isinstance(scores, RankScores):
            return scores.below(threshold)
        return [r for r, s in scores.items() if s < threshold]

    def identify_stragglers(
        self,
        gpu_rel_threshold: float = 0.75,
        section_rel_threshold: float = 0.75,
        gpu_indiv_threshold: float = 0.75,
        section_indiv_threshold: float = 0.75,
    ) -> Dict[str, Any]:
        """Ranks whose scores fall strictly below the thresholds (NaN scores are never flagged).

        Returns ``{'straggler_gpus_relative': set[StragglerId], 'straggler_gpus_individual': set,
        'straggler_sections_relative': {section: set}, 'straggler_sections_individual': {section:
        set}}``; a section appears only if at least one rank is flagged for it.
        """
        flags = getattr(self, "_device_flags", None)
        if flags is not None and flags.matches(
            gpu_rel_threshold, section_rel_threshold, gpu_indiv_threshold, section_indiv_threshold
        ):
            # thresholds equal the ones the score kernel was launched with: use its flag bytes
            gr, gi, sr, si = flags.decode()
        else:
            gr = self._below(self.gpu_relative_perf_scores, gpu_rel_threshold)
            gi = self._below(self.gpu_individual_perf_scores, gpu_indiv_threshold)
            sr = {n: self._below(v, section_rel_threshold) for n, v in self.section_relative_perf_scores.items()}
            si = {n: self._below(v, section_indiv_threshold) for n, v in self.section_individual_perf_scores.items()}
        return {
            "straggler_gpus_relative": self._ids(gr),
            "straggler_gpus_individual": self._ids(gi),
            "straggler_sections_relative": {n: self._ids(r) for n, r in sr.items() if r},
            "straggler_sections_individual": {n: self._ids(r) for n, r in si.items() if r},
        }


class _DeviceFlags:
    """Below-threshold bytes written by the score kernel, with the thresholds they were computed for."""

    def __init__(self, thresholds, flags: np.ndarray, ranks, names, cols, S, has_rel, has_indiv):
        self.thresholds = thresholds  # (gpu_rel, sec_rel, gpu_indiv, sec_indiv) as floats
        self._flags = flags  # ndarray [ranks, 2+2S] u8, or callable returning it
        self.ranks = ranks
        self.names = names
        self.cols = cols if isinstance(cols, dict) else dict(zip(names, cols))
        self.S = S
        self.has_rel = has_rel
        self.has_indiv = has_indiv

    def matches(self, gpu_rel, sec_rel, gpu_indiv, sec_indiv) -> bool:
        return (float(gpu_rel), float(sec_rel), float(gpu_indiv), float(sec_indiv)) == self.thresholds

    def _ranks_of(self, column: np.ndarray) -> List[int]:
        return [self.ranks[int(i)] for i in np.nonzero(column)[0]]

    def decode(self):
        f, S = _resolve(self._flags), self.S
        gi = self._ranks_of(f[:, 0]) if self.has_indiv else []
        gr = self._ranks_of(f[:, 1]) if self.has_rel else []
        cols = self.cols
        si = {n: self._ranks_of(f[:, 2 + cols[n]]) for n in self.names} if self.has_indiv else {}
        sr = {n: self._ranks_of(f[:, 2 + S + cols[n]]) for n in self.names} if self.has_rel else {}
        return gr, gi, sr, si


class ReportGenerator:
    """Builds :class:`Report` objects; every rank of the group must call ``generate_report`` together.

    Args:
        scores_to_compute: any of ``'relative_perf_scores'``, ``'individual_perf_scores'``.
        gather_on_rank0: rank 0's report covers all ranks and the other ranks return ``None``;
            otherwise every rank reports only itself.
        pg: process group (default: WORLD).
        node_name: name of this node in ``rank_to_node``.
        thresholds: (gpu_rel, section_rel, gpu_indiv, section_indiv) the score kernel pre-computes
            straggler flags for (default 0.75 each, the ``identify_stragglers`` defaults).
    """

    def __init__(self, scores_to_compute, gather_on_rank0=True, pg=None, node_name="<notset>",
                 thresholds: Sequence[float] = _backend_mod.DEFAULT_THRESHOLDS) -> None:
        self.is_computing_rel_scores = "relative_perf_scores" in scores_to_compute
        self.is_computing_indiv_scores = "individual_perf_scores" in scores_to_compute
        self.gather_on_rank0 = gather_on_rank0
        self.group = pg
        self.world_size = dist_utils.get_world_size(self.group)
        self.rank = dist_utils.get_rank(self.group)
        self.node_name = node_name
        self.thresholds = tuple(float(t) for t in thresholds)

        # best medians seen on this rank (individual-score reference); host copy used by the
        # dict-input path, the ring path keeps its own copy in HBM next to the rings
        self.min_local_kernel_times: Dict[str, float] = collections.defaultdict(lambda: float("inf"))
        self.min_local_section_times: Dict[str, float] = collections.defaultdict(lambda: float("inf"))

        self.name_mapper = NameMapper(pg=pg)
        # ids for the "nothing is exchanged" mode (individual scores, no gather): stays local so
        # that ``name_mapper`` is untouched, as the reference guarantees
        self._private_mapper = NameMapper(pg=pg)
        self.rank_to_node: Dict[int, str] = collections.defaultdict(lambda: "<unk>")
        self._ring_gid_state = None
        self._ring_plan = None
        # the per-report all-gather straight into RCCL on our own stream (rccl_direct.py); created
        # collectively on the first multi-rank report, None = stay on torch.distributed
        self._direct = None
        self._direct_tried = False

    # ---- pieces kept from the reference's host logic ----------------------------------------------
    @staticmethod
    def _filter_out_nccl_kernels(kernel_summaries):
        # collective kernels wait for peers, so their duration says nothing about this GPU
        return {k: v for k, v in kernel_summaries.items() if _NCCL_MARKER not in k}

    def _maybe_gather_rank_to_node(self) -> None:
        if self.rank_to_node:
            return
        if self.gather_on_rank0:
            pairs = dist_utils.all_gather_object((self.rank, self.node_name), self.group)
            self.rank_to_node = dict(pairs)
        else:
            self.rank_to_node[self.rank] = self.node_name

    def _update_local_min_times(self, kernel_summaries, section_summaries) -> None:
        for name, summ in kernel_summaries.items():
            if summ[Statistic.MED] < self.min_local_kernel_times[name]:
                self.min_local_kernel_times[name] = summ[Statistic.MED]
        for name, summ in section_summaries.items():
            if summ[Statistic.MED] < self.min_local_section_times[name]:
                self.min_local_section_times[name] = summ[Statistic.MED]

    # ---- the shared score round -------------------------------------------------------------------
    def _exchanged(self) -> bool:
        return self.is_computing_rel_scores or self.gather_on_rank0

    def _maybe_create_direct_exchange(self) -> None:
        """Collective, once: every rank calls this at the top of its first exchanging report."""
        if self._direct_tried or self.world_size == 1 or not self._exchanged():
            return
        self._direct_tried = True
        from . import rccl_direct

        self._direct = rccl_direct.create(self.group)

    def _exchange(self, be, ws):
        """The report's one collective: this rank's rows -> the [R, L] table, on the backend's stream."""
        if self._direct is not None:
            self._direct.all_gather(ws.send_ptr, ws.table_ptr, ws.local_ranks * ws.L, be.stream_handle)
            return ws.table
        with be.stream_context():  # the collective must queue behind the statistics kernel
            return dist_utils.all_gather_rows(ws.send, ws.table, self.group)

    def close(self) -> None:
        """Release the direct-exchange communicator (collective-free; safe to call more than once)."""
        if self._direct is not None:
            self._direct.close()
            self._direct = None

    def _score_round(self, kernel_names: List[str], section_names: List[str], fill_send, local_ranks: int = 1,
                     stats_rows: int = 0, stats_rows_used: Optional[int] = None, resync_first: bool = False):
        """pack -> all-gather -> score, repeated once after a name sync if any rank met a new name.

        ``fill_send(ws, mapper, names_ok)`` must leave this rank's exchange rows in ``ws.send``.
        Returns ``(workspace, mapper)`` with the results in ``ws.scores / flags / meta / stats``.
        """
        be = _backend_mod.get_backend()
        exchanged = self._exchanged()
        mapper = self.name_mapper if exchanged else self._private_mapper
        if resync_first:
            # the planned path already ran this report's first exchange and saw an incomplete flag:
            # every rank is now heading for the name sync, so join it before exchanging rows again
            mapper.sync_names(kernel_names, section_names)
        while True:
            names_ok = mapper.has_all_names(kernel_names, section_names)
            if not names_ok and (not exchanged or self.world_size == 1):
                # nobody to agree with: extend the tables locally (same order as a 1-rank gather)
                for s in section_names:
                    mapper._assign_section_id(s)
                for k in kernel_names:
                    mapper._assign_kernel_id(k)
                names_ok = True
            K, S = mapper.kernel_counter, mapper.section_counter
            world = self.world_size if exchanged else 1
            ws = be.workspace(world * local_ranks, K, S, local_ranks, stats_rows)
            if world > 1:
                with be.stream_context():  # host-packed rows are copied on the stream the report runs on
                    fill_send(ws, mapper, names_ok)
                table = self._exchange(be, ws)
            else:
                fill_send(ws, mapper, names_ok)
                table = ws.send
            be.score(ws, table, self.is_computing_indiv_scores, self.is_computing_rel_scores, self.thresholds,
                     wait=True, stats_rows=stats_rows_used)
            if ws.meta[0] == 1:
                return ws, mapper
            # some rank (maybe this one) has names without ids: cold path, then go again
            mapper.sync_names(kernel_names, section_names)

    # ---- report assembly --------------------------------------------------------------------------
    def _assemble(self, ws, mapper, local_section_names: Sequence[str], section_summaries, kernel_summaries,
                  t_start_ns: int, local_ranks: int = 1):
        S = ws.S
        scores = ws.scores.copy()
        flags = ws.flags.copy()
        has_rel, has_indiv = self.is_computing_rel_scores, self.is_computing_indiv_scores
        empty: Dict = {}
        if self.gather_on_rank0:
            if self.rank != 0:
                return None
            ranks = range(ws.R)
            names = [mapper.get_section_name(i) for i in range(S)]
            cols = {n: i for i, n in enumerate(names)}
        else:
            me = self.rank * local_ranks if self._exchanged() else 0
            scores = scores[me : me + local_ranks]
            flags = flags[me : me + local_ranks]
            ranks = range(self.rank * local_ranks, (self.rank + 1) * local_ranks)
            names = list(local_section_names)
            cols = {n: mapper.get_section_id(n) for n in names}
        gpu_i = RankScores(ranks, scores[:, 0]) if has_indiv else empty
        gpu_r = RankScores(ranks, scores[:, 1]) if has_rel else empty
        sec_i = SectionScores(names, cols, ranks, scores[:, 2 : 2 + S]) if (has_indiv and names) else empty
        sec_r = SectionScores(names, cols, ranks, scores[:, 2 + S : 2 + 2 * S]) if (has_rel and names) else empty
        elapsed_ms = (time.perf_counter_ns() - t_start_ns) * 1e-6
        report = Report(
            gpu_relative_perf_scores=gpu_r,
            section_relative_perf_scores=sec_r,
            gpu_individual_perf_scores=gpu_i,
            section_individual_perf_scores=sec_i,
            rank_to_node=dict(self.rank_to_node),
            local_section_summaries=section_summaries,
            local_kernel_summaries=kernel_summaries,
            generate_report_elapsed_time=elapsed_ms,
            gather_on_rank0=self.gather_on_rank0,
            rank=self.rank,
        )
        object.__setattr__(
            report, "_device_flags", _DeviceFlags(self.thresholds, flags, ranks, names, cols, S, has_rel, has_indiv)
        )
        return report

    # ---- steady-state plan of the ring path ---------------------------------------------------------
    class _RingPlan:
        """Everything about a ring report that only changes when names, ids or topology change."""

        __slots__ = ("key", "ws", "mapper", "snames", "knames", "names", "cols", "ranks", "row_lo", "row_hi",
                     "rows_used", "stats_needed", "section_rows", "kernel_rows")

    def _build_ring_plan(self, key, rings, section_rows, kernel_rows, local_ranks):
        be = _backend_mod.get_backend()
        exchanged = self._exchanged()
        mapper = self.name_mapper if exchanged else self._private_mapper
        knames, snames = list(kernel_rows.keys()), list(section_rows.keys())
        if not mapper.has_all_names(knames, snames):
            if exchanged and self.world_size > 1:
                return None  # ids must be agreed with the other ranks first: general path
            for s in snames:
                mapper._assign_section_id(s)
            for k in knames:
                mapper._assign_kernel_id(k)
        K, S = mapper.kernel_counter, mapper.section_counter
        world = self.world_size if exchanged else 1
        total_rows = rings.local_ranks * rings.rows_per_rank
        plan = self._RingPlan()
        plan.key = key
        plan.mapper = mapper
        plan.ws = be.workspace(world * local_ranks, K, S, local_ranks, total_rows)
        plan.snames, plan.knames = snames, knames
        plan.rows_used = rings.rows_used
        plan.stats_needed = rings.rows_used if rings.local_ranks == 1 else total_rows
        plan.section_rows, plan.kernel_rows = section_rows, kernel_rows
        if self.gather_on_rank0:
            plan.ranks = range(plan.ws.R)
            plan.names = [mapper.get_section_name(i) for i in range(S)]
            plan.row_lo, plan.row_hi = 0, plan.ws.R
        else:
            me = self.rank * local_ranks if exchanged else 0
            plan.ranks = range(self.rank * local_ranks, (self.rank + 1) * local_ranks)
            plan.names = snames
            plan.row_lo, plan.row_hi = me, me + local_ranks
        plan.cols = {n: mapper.get_section_id(n) for n in plan.names}
        # point every ring row at its slot of the exchange row (cold)
        for name, row in rings.kernel_row_names.items():
            g = mapper.kernel_name_to_id.get(name, -1) if _NCCL_MARKER not in name else -1
            rings.configure(row, 1, g)
        for name, row in rings.section_row_names.items():
            g = mapper.section_name_to_id.get(name)
            rings.configure(row, 0, K + g if g is not None else -1)
        plan.ws.send_initialised = False
        self._ring_gid_state = None  # the general path must re-derive its own view if it runs next
        return plan

    def _report_from_plan(self, plan, rings, t0):
        """The steady-state report: three C calls (+ one collective), one host copy, lazy views."""
        be = _backend_mod.get_backend()
        ws = plan.ws
        if self.world_size > 1 and self._exchanged():
            rings.report_local(ws, True, rows_active=plan.rows_used)
            table = self._exchange(be, ws)
            be.score(ws, table, self.is_computing_indiv_scores, self.is_computing_rel_scores, self.thresholds,
                     wait=True, stats_rows=plan.stats_needed)
        else:
            rings.report_local(ws, True, rows_active=plan.rows_used)
            be.score(ws, ws.send, self.is_computing_indiv_scores, self.is_computing_rel_scores, self.thresholds,
                     wait=True, stats_rows=plan.stats_needed)
        if ws.meta[0] != 1:
            return False  # another rank met a new name: fall back to the general (name-syncing) path
        if self.gather_on_rank0 and self.rank != 0:
            return None
        S, W, R = ws.S, ws.W, ws.R
        blob = ws.host_block()  # one memcpy out of the pinned block; everything below views into it
        lo, hi = plan.row_lo, plan.row_hi
        off_s, off_f, off_t = ws._off_scores, ws._off_flags, ws._off_stats
        cache = {}

        def scores():
            a = cache.get("s")
            if a is None:
                a = cache["s"] = blob[off_s : off_s + R * W * 4].view(np.float32).reshape(R, W)[lo:hi]
            return a

        def flags():
            return blob[off_f : off_f + R * W].reshape(R, W)[lo:hi]

        def stats():
            return blob[off_t : off_t + plan.stats_needed * 32].view(np.float32).reshape(plan.stats_needed, 8)

        has_rel, has_indiv = self.is_computing_rel_scores, self.is_computing_indiv_scores
        ranks, names, cols = plan.ranks, plan.names, plan.cols
        empty: Dict = {}
        report = Report(
            gpu_relative_perf_scores=RankScores(ranks, lambda: scores()[:, 1]) if has_rel else empty,
            section_relative_perf_scores=(SectionScores(names, cols, ranks, lambda: scores()[:, 2 + S : 2 + 2 * S])
                                          if (has_rel and names) else empty),
            gpu_individual_perf_scores=RankScores(ranks, lambda: scores()[:, 0]) if has_indiv else empty,
            section_individual_perf_scores=(SectionScores(names, cols, ranks, lambda: scores()[:, 2 : 2 + S])
                                            if (has_indiv and names) else empty),
            rank_to_node=self.rank_to_node if type(self.rank_to_node) is dict else dict(self.rank_to_node),
            local_section_summaries=StatSummaries(plan.section_rows, stats),
            local_kernel_summaries=StatSummaries(plan.kernel_rows, stats),
            generate_report_elapsed_time=(time.perf_counter_ns() - t0) * 1e-6,
            gather_on_rank0=self.gather_on_rank0,
            rank=self.rank,
        )
        object.__setattr__(
            report, "_device_flags", _DeviceFlags(self.thresholds, flags, ranks, names, cols, S, has_rel, has_indiv)
        )
        return report

    # ---- public: summaries given as dicts (reference signature) -------------------------------------
    def generate_report(self, section_summaries: Mapping[str, _SummaryType],
                        kernel_summaries: Mapping[str, _SummaryType]):
        """Score the given per-rank summaries (name -> {Statistic: value}).

        Collective.  Returns a :class:`Report`, or ``None`` on ranks other than 0 when
        ``gather_on_rank0`` is set.  The summaries are packed into this rank's exchange row on the
        host; exchange and scoring run on the device exactly as in the ring path.
        """
        t0 = time.perf_counter_ns()
        self.world_size = dist_utils.get_world_size(self.group)
        self.rank = dist_utils.get_rank(self.group)
        if not self._direct_tried:
            self._maybe_create_direct_exchange()
        kernel_summaries = self._filter_out_nccl_kernels(kernel_summaries)
        self._maybe_gather_rank_to_node()
        if self.is_computing_indiv_scores:
            self._update_local_min_times(kernel_summaries, section_summaries)
        knames, snames = list(kernel_summaries.keys()), list(section_summaries.keys())

        def fill_send(ws, mapper, names_ok):
            K, KS, L = ws.K, ws.K + ws.S, ws.L
            row = np.zeros(L, dtype=np.float32)
            row[:KS] = -1.0
            row[KS : 2 * KS] = np.nan
            for name, summ in kernel_summaries.items():
                g = mapper.kernel_name_to_id.get(name)
                if g is None:
                    continue  # no id yet: this round only carries the "names incomplete" flag
                row[g] = summ[Statistic.MED]
                row[KS + g] = self.min_local_kernel_times[name] if self.is_computing_indiv_scores else np.nan
                row[2 * KS + g] = summ[Statistic.NUM] * summ[Statistic.AVG]
            for name, summ in section_summaries.items():
                g = mapper.section_name_to_id.get(name)
                if g is None:
                    continue
                row[K + g] = summ[Statistic.MED]
                row[KS + K + g] = self.min_local_section_times[name] if self.is_computing_indiv_scores else np.nan
            row[L - 1] = 1.0 if names_ok else 0.0
            ws.set_send_row(0, row)

        ws, mapper = self._score_round(knames, snames, fill_send)
        return self._assemble(ws, mapper, snames, section_summaries, kernel_summaries, t0)

    # ---- internal: summaries never leave the device (Detector path) -----------------------------------
    def generate_report_from_rings(self, rings, section_rows: Mapping[str, int], kernel_rows: Mapping[str, int],
                                   local_ranks: int = 1):
        """Report straight from the device rings: statistics kernel -> exchange -> score kernel.

        ``section_rows`` / ``kernel_rows`` map the names that hold samples this window to their ring
        rows.  Replaces ``_get_section_summaries`` + ``_get_kernel_summaries`` + ``generate_report``
        of the reference (straggler.py:236-239) without materialising per-section Python objects.
        """
        t0 = time.perf_counter_ns()
        self.world_size = dist_utils.get_world_size(self.group)
        self.rank = dist_utils.get_rank(self.group)
        if not self._direct_tried:
            self._maybe_create_direct_exchange()
        # steady state: same name tables as last time -> run the cached plan
        key = (id(section_rows), len(section_rows), id(kernel_rows), len(kernel_rows), self.name_mapper.version,
               self._private_mapper.version, self.world_size, self.rank, rings.rows_used, local_ranks, id(rings))
        plan = self._ring_plan
        resync_first = False
        if plan is not None and plan.key == key:
            out = self._report_from_plan(plan, rings, t0)
            if out is not False:
                return out
            self._ring_plan = None
            resync_first = True  # some OTHER rank met a new name during this report's exchange
        kernel_rows = {k: r for k, r in kernel_rows.items() if _NCCL_MARKER not in k} if any(
            _NCCL_MARKER in k for k in kernel_rows) else kernel_rows
        self._maybe_gather_rank_to_node()
        if local_ranks > 1 and not getattr(self, "_rank_to_node_folded", False):
            # folded runs: every logical rank inherits the node of the process that holds it
            per_proc = dict(self.rank_to_node)
            self.rank_to_node = {p * local_ranks + q: node for p, node in per_proc.items() for q in range(local_ranks)}
            self._rank_to_node_folded = True
        knames, snames = list(kernel_rows.keys()), list(section_rows.keys())
        rows_used = rings.rows_used
        total_rows = rings.local_ranks * rings.rows_per_rank
        stats_needed = rows_used if rings.local_ranks == 1 else total_rows

        def fill_send(ws, mapper, names_ok):
            state = (id(mapper), mapper.version, rows_used, id(ws))
            if self._ring_gid_state != state:
                # ids changed (cold): re-point every ring row at its slot of the exchange row
                K = ws.K
                for name, row in rings.kernel_row_names.items():
                    g = mapper.kernel_name_to_id.get(name, -1) if _NCCL_MARKER not in name else -1
                    rings.configure(row, 1, g)
                for name, row in rings.section_row_names.items():
                    g = mapper.section_name_to_id.get(name)
                    rings.configure(row, 0, K + g if g is not None else -1)
                ws.send_initialised = False
                self._ring_gid_state = state
            rings.report_local(ws, names_ok, rows_active=rows_used)

        ws, mapper = self._score_round(knames, snames, fill_send, local_ranks=local_ranks, stats_rows=total_rows,
                                       stats_rows_used=stats_needed, resync_first=resync_first)
        stats = ws.stats[:stats_needed].copy()
        sec_summ = StatSummaries(dict(section_rows), stats)
        ker_summ = StatSummaries(dict(kernel_rows), stats)
        report = self._assemble(ws, mapper, snames, sec_summ, ker_summ, t0, local_ranks=local_ranks)
        # names are settled now: the next report with the same tables takes the planned path
        key = (id(section_rows), len(section_rows), id(kernel_rows), len(kernel_rows), self.name_mapper.version,
               self._private_mapper.version, self.world_size, self.rank, rings.rows_used, local_ranks, id(rings))
        self._ring_plan = self._build_ring_plan(key, rings, section_rows, kernel_rows, local_ranks)
        return report
