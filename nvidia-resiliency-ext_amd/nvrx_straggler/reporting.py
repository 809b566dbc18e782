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
import itertools
import logging
import os
import threading
import time
from typing import Any, Dict, List, Mapping, Optional, Sequence

import numpy as np

from . import backend as _backend_mod
from . import dist_utils
from .name_mapper import NameMapper
from .statistics import NUM_COLUMN, STAT_KEYS, Statistic

_SummaryType = Mapping[Statistic, float]
_LOG = logging.getLogger(__name__)



def _load_pyread():
    """csrc/nvrx_pyread.c: the same dicts without numpy's intermediate lists (host-side formatting only; same results).
    ``NVRX_DEBUG_LIB_DIR`` (the sanitizer builds, tools/run_sanitized.sh) may hold its own build of it, which then wins."""
    alt = os.environ.get("NVRX_DEBUG_LIB_DIR", "")
    if alt:
        import glob
        import importlib.util

        for path in sorted(glob.glob(os.path.join(alt, "_nvrx_pyread*.so"))):
            spec = importlib.util.spec_from_file_location(f"{__package__}._nvrx_pyread", path)
            if spec is not None and spec.loader is not None:
                try:
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                except (ImportError, OSError) as e:  # a stale / ABI-mismatched file there: the packaged module, or Python
                    _LOG.warning("ignoring %s: %s", path, e)
                    continue
                return mod
    try:
        from . import _nvrx_pyread
    except ImportError:  # not built: the Python builders below
        return None
    return _nvrx_pyread


_pyread = _load_pyread()
# (said once, at INFO: which of the three formatting paths -- all with the same results -- this interpreter gets)
if _pyread is None:
    _LOG.info("nvrx straggler: csrc/nvrx_pyread.c is not built for this interpreter; a report's dicts are built by the Python "
              "builders (same results, slower reads)")
else:
    try:
        if not _pyread.inplace():
            import sys as _sys

            _LOG.info("nvrx straggler: the in-place fill of a report's cloned dicts follows CPython 3.10's dict layout; Python %d.%d "
                      "uses the public dict API instead (same results, ~10 us more per fully read 8 x 64 report)", *_sys.version_info[:2])
    except Exception:  # noqa: BLE001  (an older build of the helper without the query)
        pass


def _copy_sets(d: Dict[str, set]) -> Dict[str, set]:
    """``{name: set}`` copied one level deep (callers of ``identify_stragglers`` may do what they like with their sets)."""
    if _pyread is not None:
        return _pyread.copy_sets(d)
    return {n: v.copy() for n, v in d.items()}

# Kernels of the collective library are kept out of the GPU score: their duration is peer-wait time.  The reference
# drops every key containing "ncclDev" (reporting.py:330-336), which on CUDA matches NCCL's ncclDevKernel_* family.
# RCCL's device kernels are named differently -- rcclGenericKernel<N, bool>(ncclDevKernelArgsStorage<4096>) and
# mscclKernel_<op>_<type>_<proto>_<bool>(ncclDevComm*, mscclAlgo*, mscclWork*) -- and the keys here carry MANGLED
# names, where the argument types appear, so the reference's substring still hits every one of them
# (tests/test_host_logic.py::test_rccl_kernel_names_of_the_installed_library_are_filtered reads the names out of the
# installed librccl.so).  The two RCCL family names are matched as well, so the filter does not depend on an argument
# type staying in the signature.
_NCCL_MARKER = "ncclDev"
_COLLECTIVE_MARKERS = (_NCCL_MARKER, "rcclGenericKernel", "mscclKernel")


def is_collective_kernel(name: str) -> bool:
    return _NCCL_MARKER in name or "rcclGenericKernel" in name or "mscclKernel" in name


@dataclasses.dataclass(frozen=True)
class StragglerId:
    """Identity of a flagged rank: global rank and the node it runs on."""

    rank: int
    node: str


# --------------------------------------------------------------------------------------------------
# Report: a frozen record of plain dicts, some of them built on first read
# --------------------------------------------------------------------------------------------------
def _row_selector(rows: Mapping[str, int]):
    """How to cut ``rows``' statistics out of the block: a slice when the rows are consecutive (ring rows are handed
    out in order of first use, so they nearly always are), else an index array."""
    idx = list(rows.values())
    if idx and idx == list(range(idx[0], idx[0] + len(idx))):
        return slice(idx[0], idx[0] + len(idx))
    return np.asarray(idx, dtype=np.intp)


def _summaries_from_rows(rows: Mapping[str, int], stats: np.ndarray, selector=None, name_template=None,
                         recycle=None, tuples=None) -> Dict[str, Dict[Statistic, Any]]:
    """``name -> {Statistic: value}`` from device statistics rows (``_get_section_summaries``' result shape,
    straggler.py:185-195), NUM as an integer as in the reference (straggler.py:194).  Built by ``_nvrx_pyread`` when it
    is there; else one C-level conversion of the whole block and one ``dict(zip(keys, row))`` per name (``Statistic``
    hashes by identity)."""
    if not rows:
        return {}
    if _pyread is not None and stats.dtype == np.float32 and stats.flags.c_contiguous and stats.shape[1] == 8:
        names, idx = tuples if tuples is not None else (tuple(rows), tuple(rows.values()))
        return _pyread.summaries(names, STAT_KEYS, stats, idx, name_template, recycle)
    block = stats[_row_selector(rows) if selector is None else selector]
    vals = block[:, : NUM_COLUMN + 1].tolist()
    out = dict(zip(rows, map(dict, map(zip, itertools.repeat(STAT_KEYS), vals))))
    num = Statistic.NUM
    counts = block[:, NUM_COLUMN]
    if not np.isfinite(counts).all():  # cannot come out of the statistics kernel; the C builder raises here too
        raise ValueError("cannot convert a non-finite NUM statistic to an integer")
    for d, n in zip(out.values(), counts.astype(np.int64).tolist()):
        d[num] = n
    return out


class _PendingBlock:
    """Result block of a report the host has not waited for (asynchronous reports): the kernels are enqueued, the
    pinned block fills in when they run.  ``complete()`` polls the completion word and leaves the data where it is;
    ``wait()`` also takes the private copy -- when the report is first read, or (``detach``, called by the block) before
    the block is written again, two reports later, if somebody still holds the report.  An asynchronous report that is
    never read therefore costs its successor one poll of a word that has long been written, not a copy of the block: at
    production cadence that copy ran cold and was half of what enqueueing the next report cost (47 of 95 us,
    profiles/r04f_cadence_breakdown.txt)."""

    __slots__ = ("backend", "ws", "blk", "seq", "blob", "lock", "__weakref__")

    def __init__(self, backend, ws, seq: int):
        self.backend, self.ws, self.seq = backend, ws, seq
        self.blk = getattr(ws, "block", None)  # the one of the workspace's result blocks this report was enqueued into
        self.blob: Optional[np.ndarray] = None
        self.lock = threading.Lock()
        if self.blk is not None:
            self.blk.attach(self)  # the block collects this report (if still held) before anything writes it again

    def complete(self) -> int:
        """Wait until the report has published its results; returns its ``names complete`` word (meta[0])."""
        with self.lock:
            blk = self.blk
            if self.blob is None and blk is not None:
                words = getattr(blk, "meta_words", None)
                if words is None:  # (the CPU checker backend's block)
                    self.backend.wait_seq(self.ws, self.seq, block=blk)
                    return int(blk.meta[0])
                if words[4] != self.seq:  # (usually published long ago: one look at the word, no call)
                    self.backend.wait_seq(self.ws, self.seq, block=blk)
                return words[0]
        return int(self.wait()[0:4].view(np.uint32)[0])

    def wait(self) -> np.ndarray:
        with self.lock:
            if self.blob is None:
                if self.blk is not None:
                    self.backend.wait_seq(self.ws, self.seq, block=self.blk)
                    self.blob = self.blk.host_block()
                else:  # (the CPU checker backend of the tests: one block, computed at enqueue time)
                    self.backend.wait_seq(self.ws, self.seq)
                    self.blob = self.ws.host_block()
                self.backend = self.ws = self.blk = None  # nothing of the live workspace is referenced any more
        return self.blob

    def detach(self) -> None:
        self.wait()


_LIVE_LOCK = threading.Lock()  # a report may be read on another thread while the generator collects it


class _LiveBlock:
    """The result block of a synchronous one-call report, still in place.  Nothing is copied when the report returns:
    ``head()`` / ``stats()`` take private copies when the report is first read, and the block calls ``detach()`` before
    anything writes it again -- two reports later (a workspace alternates between two blocks), and it copies only if
    somebody still holds the report by then.  The statistics rows have a completion word of their own (a resident
    score kernel forwards them after the scores)."""

    __slots__ = ("backend", "ws", "blk", "seq", "rows", "_head", "_head_bytes", "_stats", "__weakref__")

    def __init__(self, backend, ws, seq: int, rows: int):
        self.backend, self.ws, self.seq, self.rows = backend, ws, seq, rows
        self.blk = getattr(ws, "block", ws)  # (the CPU checker backend's workspace is its own single block)
        self._head: Optional[np.ndarray] = None
        self._head_bytes: Optional[bytes] = None
        self._stats: Optional[np.ndarray] = None

    def head_bytes(self) -> bytes:
        """meta | scores | flags as ONE ``bytes`` copy of the block (a single memcpy; the flag path of
        ``identify_stragglers`` needs nothing else of the block and nothing of numpy)."""
        if self._head_bytes is None:
            with _LIVE_LOCK:
                if self._head_bytes is None:
                    grab = getattr(self.blk, "host_head_bytes", None)
                    self._head_bytes = grab() if grab is not None else self.blk.host_head().tobytes()
                    self._release()
        return self._head_bytes

    def head(self) -> np.ndarray:
        if self._head is None:
            self._head = np.frombuffer(self.head_bytes(), dtype=np.uint8)  # read-only view of the private copy
        return self._head

    def stats(self) -> np.ndarray:
        if self._stats is None:
            with _LIVE_LOCK:
                if self._stats is None:
                    if self.blk is self.ws:
                        self.backend.wait_seq(self.ws, self.seq, stats=True)
                    else:
                        self.backend.wait_seq(self.ws, self.seq, stats=True, block=self.blk)
                    self._stats = self.blk.host_stats(self.rows)
                    self._release()
        return self._stats

    def in_place(self, fn, stats: bool):
        """``fn(the block's pinned bytes)`` while the block is still this report's -- nothing is copied, not even for the
        mapping builders: under the lock the generator cannot collect this report (``detach`` takes it too), and until it
        has, nothing is enqueued that writes the block again.  ``stats``: the statistics rows are what is read (they have a
        completion word of their own).  None when a private copy exists already or the block exports no flat buffer."""
        with _LIVE_LOCK:
            blk = self.blk
            if blk is None or (self._stats if stats else self._head_bytes) is not None:
                return None
            live = getattr(blk, "_host", None)
            if live is None:
                return None
            if stats:
                if blk is self.ws:
                    self.backend.wait_seq(self.ws, self.seq, stats=True)
                else:
                    self.backend.wait_seq(self.ws, self.seq, stats=True, block=blk)
            try:
                return fn(live)
            except (BufferError, TypeError):
                return None

    def detach(self) -> None:
        self.head_bytes()
        self.stats()

    def _release(self) -> None:
        if self._head_bytes is not None and self._stats is not None:
            self.backend = self.ws = self.blk = None  # nothing of the live workspace is referenced any more


class _View:
    """What all reports of one plan share (immutable once built): table shape, rank / name tables, which score families
    exist, where this report's rows sit in the result block, the thresholds the score kernel flagged with."""

    __slots__ = ("S", "ranks", "names", "cols", "has_rel", "has_indiv", "section_rows", "kernel_rows", "layout", "thresholds",
                 "_rank_list", "_col_index", "_selectors", "_ids", "_memo", "_tuples", "_flag_args")

    def memo(self) -> dict:
        try:
            return self._memo
        except AttributeError:
            self._memo = {}
            return self._memo

    def flag_memo(self) -> list:
        """[flag bytes, (ids, gpu_rel, gpu_indiv, sec_rel, sec_indiv)] of the last flag table ``_nvrx_pyread.flagged``
        decoded for this plan (it hands an unchanged table's sets out as copies)."""
        m = self.memo()
        try:
            return m["c"]
        except KeyError:
            m["c"] = [None, None]
            return m["c"]

    def rank_list(self) -> list:
        try:
            return self._rank_list
        except AttributeError:
            self._rank_list = list(self.ranks)
            return self._rank_list

    def rank_tuple(self) -> tuple:
        try:
            return self._tuples[0]
        except AttributeError:
            idx = self.col_index()
            self._tuples = (tuple(self.ranks), tuple(self.names), None if idx is None else tuple(int(i) for i in idx))
            return self._tuples[0]

    def names_tuple(self) -> tuple:
        self.rank_tuple()
        return self._tuples[1]

    def col_tuple(self):
        self.rank_tuple()
        return self._tuples[2]

    def name_template(self, which: str = "names"):
        """``{name: None}`` over the view's section names (``names``) or over the names of its section / kernel rows:
        ``_nvrx_pyread`` clones it for the outer dict of a mapping instead of inserting the names one by one (the template
        itself is never handed out and never modified)."""
        try:
            return self._memo[("tmpl", which)]
        except (AttributeError, KeyError):
            src = self.names if which == "names" else getattr(self, which)
            t = self.memo()[("tmpl", which)] = dict.fromkeys(src)
            return t

    def row_tuples(self, which: str):
        """``(names, rows)`` of ``section_rows`` / ``kernel_rows`` as the two tuples ``_nvrx_pyread.summaries`` takes."""
        try:
            return self._memo[("rows", which)]
        except (AttributeError, KeyError):
            rows = getattr(self, which)
            t = self.memo()[("rows", which)] = (tuple(rows), tuple(rows.values()))
            return t

    def recycle_list(self, which: str, slots: int = 1):
        """The last mapping(s) ``_nvrx_pyread`` built for this plan: when nothing else refers to one any more (the report it
        was built for is gone and nobody kept the dict) the next report's mapping is that very object with its values
        swapped -- see csrc/nvrx_pyread.c, "recycling".  A list per kind of mapping, owned by the plan's memo."""
        try:
            return self._memo[("recycle", which)]
        except (AttributeError, KeyError):
            lst = self.memo()[("recycle", which)] = [None] * slots
            return lst

    def col_index(self):
        """Score-table column of every name of ``names``, in that order (None: the identity)."""
        try:
            return self._col_index
        except AttributeError:
            idx = [self.cols[n] for n in self.names]
            self._col_index = None if idx == list(range(self.S)) else np.asarray(idx, dtype=np.intp)
            return self._col_index

    def selector(self, which: str):
        try:
            sel = self._selectors
        except AttributeError:
            sel = self._selectors = {}
        if which not in sel:
            sel[which] = _row_selector(getattr(self, which))
        return sel[which]

    def straggler_ids(self, rank_to_node) -> list:
        """``StragglerId`` of every row of this view's score table (built once per ``rank_to_node`` object: the
        generator hands the same dict to every report of a plan)."""
        try:
            owner, ids = self._ids
            if owner is rank_to_node:
                return ids
        except AttributeError:
            pass
        ids = [StragglerId(rank=r, node=rank_to_node[r]) for r in self.ranks]
        self._ids = (rank_to_node, ids)
        return ids


class _ScoreSource:
    """What a steady-state report keeps of the result block: the (shared, immutable) view of its plan and the block
    itself -- live, in flight or a private copy.  The six mapping fields of ``Report`` are built from it as PLAIN
    dicts the first time each one is read; a report whose scores are only thresholded (``identify_stragglers`` with
    the kernel's thresholds) never builds any.  The arrays are cut out of the block on first use."""

    __slots__ = ("view", "pending", "scores", "stats", "flags")

    def __init__(self, view: _View, pending=None):
        self.view = view
        self.pending = pending  # ndarray (private copy), _LiveBlock (synchronous report) or _PendingBlock (asynchronous)
        self.scores = self.stats = self.flags = None

    def cut(self, blob: np.ndarray) -> None:
        """Views of this report's scores / flags / statistics inside a private copy of the result block."""
        off_s, off_f, off_t, R, W, lo, hi, stats_rows = self.view.layout
        self.scores = blob[off_s : off_s + R * W * 4].view(np.float32).reshape(R, W)[lo:hi]
        self.flags = blob[off_f : off_f + R * W].reshape(R, W)[lo:hi]
        if blob.size >= off_t + stats_rows * 32:  # a copy of the whole block; else the statistics come later
            self.stats = blob[off_t : off_t + stats_rows * 32].view(np.float32).reshape(stats_rows, 8)

    def statistics(self) -> np.ndarray:
        st = self.stats
        if st is None:
            pend = self.pending
            if self.scores is None and type(pend) is _LiveBlock:
                st = pend  # (a live block hands its statistics rows out on their own: no cut of the head needed for them)
            else:
                self.ensure()
                st = self.stats
        if type(st) is _LiveBlock:
            st = self.stats = st.stats()
        return st

    def raw_scores(self):
        """``(buffer, byte offset, rows, width)`` of this report's score rows for ``_nvrx_pyread`` -- for a live block its
        ``bytes`` copy of the head, addressed by offset: no numpy view is cut for a report whose mappings are built in C."""
        if self.scores is None and type(self.pending) is _LiveBlock:
            off_s, _, _, _, W, lo, hi, _ = self.view.layout
            return self.pending.head_bytes(), off_s + lo * W * 4, hi - lo, W
        sc = self.ensure().scores
        if sc.dtype == np.float32 and sc.flags.c_contiguous:
            return sc, 0, sc.shape[0], sc.shape[1]
        return None

    def ensure(self) -> "_ScoreSource":
        if self.scores is None:
            pend = self.pending
            if pend is not None:
                t = type(pend)
                if t is _LiveBlock:
                    self.stats = pend  # resolved by statistics()
                    self.cut(pend.head())
                else:
                    self.cut(pend if t is np.ndarray else pend.wait())
                self.pending = None
        return self

    def flag_bytes(self) -> bytes:
        """This report's flag rows ([ranks, 2+2S] u8, row-major) as ``bytes``."""
        if self.scores is None and type(self.pending) is _LiveBlock:
            _, off_f, _, _, W, lo, hi, _ = self.view.layout
            return self.pending.head_bytes()[off_f + lo * W : off_f + hi * W]
        return self.ensure().flags.tobytes()

    def flagged(self, rank_to_node, thresholds):
        """``identify_stragglers`` at the thresholds the score kernel flagged with, decoded by ``_nvrx_pyread.flagged``
        straight from the flag bytes (None: other thresholds, or the helper was not built -- the caller's general path).
        One Python frame and one C call: a report read once a minute runs all of this cold."""
        v = self.view
        if _pyread is None or thresholds != v.thresholds:
            return None
        try:
            owner, off, args = v._flag_args
        except AttributeError:
            owner = None
        if owner is not rank_to_node:  # once per plan: the generator hands the same rank_to_node to all its reports
            _, off_f, _, _, W, lo, hi, _ = v.layout
            off = off_f + lo * W
            args = (hi - lo, W, v.S, v.has_rel, v.has_indiv, v.straggler_ids(rank_to_node), v.names_tuple(), v.col_tuple(),
                    v.flag_memo())
            v._flag_args = (rank_to_node, off, args)
        pend = self.pending
        out = None
        if self.scores is None and type(pend) is _PendingBlock:
            # an asynchronous report read late: its flags are decoded in place as well -- the report's own lock keeps the
            # block from collecting it (``detach``), and until it has, nothing is enqueued that writes the block again
            with pend.lock:
                blk = pend.blk
                live = getattr(blk, "_host", None) if (pend.blob is None and blk is not None) else None
                if live is not None:
                    pend.backend.wait_seq(pend.ws, pend.seq, block=blk)
                    try:
                        out = _pyread.flagged(live, off, *args)
                    except (BufferError, TypeError):
                        out = None
        if out is not None:
            pass
        elif self.scores is None and type(pend) is _LiveBlock:
            with _LIVE_LOCK:
                # the result block itself, nothing copied: under the lock the generator cannot collect this report
                # (``detach`` takes it too), and until it has, nothing is enqueued that writes the block again
                blk = pend.blk
                live = getattr(blk, "_host", None) if pend._head_bytes is None else None
                if live is not None:
                    try:
                        out = _pyread.flagged(live, off, *args)
                    except (BufferError, TypeError):  # a block that does not export a flat byte buffer: its private copy
                        out = None
            if out is None:
                out = _pyread.flagged(pend.head_bytes(), off, *args)
        else:
            buf = self.ensure().flags
            out = _pyread.flagged(buf if buf.flags.c_contiguous else np.ascontiguousarray(buf), 0, *args)
        return {"straggler_gpus_relative": out[0], "straggler_gpus_individual": out[1],
                "straggler_sections_relative": out[2], "straggler_sections_individual": out[3]}

    def device_flags(self) -> "_DeviceFlags":
        v = self.view
        return _DeviceFlags(v.thresholds, self, v.ranks, v.names, v.cols, v.S, v.has_rel, v.has_indiv)

    def build(self, field: str, stash: Optional[dict] = None):
        """One of the six mappings as a plain dict.  ``stash``: the report's ``__dict__`` -- a call that can build a
        sibling in the same pass (both section-score families share names, ranks and hashes) leaves it there."""
        v = self.view
        if field == "local_section_summaries" or field == "local_kernel_summaries":  # (nothing of the score rows is needed)
            which = "section_rows" if field == "local_section_summaries" else "kernel_rows"
            rows = getattr(v, which)
            if not rows:
                return {}
            live = self.stats if self.stats is not None else self.pending
            if _pyread is not None and type(live) is _LiveBlock and (self.scores is None or self.stats is live):
                # the statistics rows straight out of the result block: no numpy copy of them for a report that only builds dicts
                names, idx = v.row_tuples(which)
                off_t = v.layout[2]
                out = live.in_place(lambda buf: _pyread.summaries(names, STAT_KEYS, buf, idx, v.name_template(which),
                                                                  v.recycle_list(which), off_t), True)
                if out is not None:
                    return out
            return _summaries_from_rows(rows, self.statistics(), v.selector(which), v.name_template(which), v.recycle_list(which),
                                        v.row_tuples(which))
        if _pyread is not None:
            # (the score rows are NOT read in place from a live block: the builders walk them column by column, and strided
            #  reads of the pinned, device-mapped block cost the CPU twice what one sequential copy of the head + reads of
            #  the copy cost -- 12.0 against 6.7 us for both section families, tools/report_read_fields.py)
            raw = self.raw_scores()
            if raw is not None:
                return self._score_field(field, stash, *raw)
        sc = self.ensure().scores
        if field == "gpu_relative_perf_scores" or field == "gpu_individual_perf_scores":
            col = 1 if field == "gpu_relative_perf_scores" else 0
            if not (v.has_rel if col else v.has_indiv):
                return {}
            return dict(zip(v.ranks, sc[:, col].tolist()))
        if field == "section_relative_perf_scores" or field == "section_individual_perf_scores":
            rel = field == "section_relative_perf_scores"
            if not ((v.has_rel if rel else v.has_indiv) and v.names):
                return {}
            return self._sections(2 + v.S if rel else 2)
        raise AttributeError(field)

    def _score_field(self, field: str, stash: Optional[dict], buf, off: int, nrows: int, W: int):
        """A score mapping built by ``_nvrx_pyread`` from the [nrows, W] f32 rows at ``buf + off``; never None."""
        v = self.view
        S = v.S
        if field == "gpu_relative_perf_scores" or field == "gpu_individual_perf_scores":
            col = 1 if field == "gpu_relative_perf_scores" else 0
            if not (v.has_rel if col else v.has_indiv):
                return {}
            rt = v.rank_tuple()
            other = "gpu_individual_perf_scores" if col else "gpu_relative_perf_scores"
            if stash is not None and other not in stash and (v.has_indiv if col else v.has_rel):
                stash[other] = _pyread.ranks(rt, buf, off, nrows, W, 1 - col)  # the sibling costs one more C call now, no frame later
            return _pyread.ranks(rt, buf, off, nrows, W, col)
        if field == "section_relative_perf_scores" or field == "section_individual_perf_scores":
            rel = field == "section_relative_perf_scores"
            if not ((v.has_rel if rel else v.has_indiv) and v.names):
                return {}
            other = "section_individual_perf_scores" if rel else "section_relative_perf_scores"
            if stash is not None and other not in stash and (v.has_indiv if rel else v.has_rel):
                mine, sibling = _pyread.sections(v.names_tuple(), v.rank_tuple(), buf, off, nrows, W,
                                                 2 + S if rel else 2, v.col_tuple(), 2 if rel else 2 + S, v.name_template(),
                                                 v.recycle_list("sections", 2))
                stash[other] = sibling
                return mine
            return _pyread.sections(v.names_tuple(), v.rank_tuple(), buf, off, nrows, W, 2 + S if rel else 2, v.col_tuple(), -1,
                                    v.name_template(), v.recycle_list("sections", 2))
        raise AttributeError(field)

    def _sections(self, first_col: int) -> Dict[str, Dict[int, float]]:
        """``section -> {rank: score}`` (reporting.py:196-217 builds the same shape score by score): ``_nvrx_pyread``
        walks the [ranks, 2+2S] block directly; without it, one C-level conversion of the [S, ranks] block and one
        ``dict(zip(...))`` per section."""
        v = self.view
        sc = self.scores
        idx = v.col_index()
        if _pyread is not None and sc.dtype == np.float32 and sc.flags.c_contiguous:
            return _pyread.sections(v.names_tuple(), v.rank_tuple(), sc, 0, sc.shape[0], sc.shape[1], first_col, v.col_tuple(), -1,
                                    v.name_template())
        block = sc[:, first_col : first_col + v.S].T
        if idx is not None:
            block = block[idx]
        return dict(zip(v.names, map(dict, map(zip, itertools.repeat(v.rank_list()), block.tolist()))))


_LAZY_FIELDS = frozenset((
    "gpu_relative_perf_scores", "section_relative_perf_scores", "gpu_individual_perf_scores",
    "section_individual_perf_scores", "local_section_summaries", "local_kernel_summaries",
))


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

    Every mapping is a plain ``dict`` (of plain dicts / floats), as in the reference, so reports can be
    pickled, sent through ``multiprocessing`` queues, deep-copied and ``json.dumps``-ed.  Reports that come
    straight from the device build each mapping on first read (``_from_device``); nothing about that is
    visible from outside except that an unread mapping costs nothing.
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

    # ---- construction from the device result block -------------------------------------------------
    @classmethod
    def _from_device(cls, source: _ScoreSource, rank_to_node, elapsed_ms: float, gather_on_rank0: bool,
                     rank: Optional[int]) -> "Report":
        self = object.__new__(cls)
        d = self.__dict__
        d["rank_to_node"] = rank_to_node
        d["generate_report_elapsed_time"] = elapsed_ms
        d["gather_on_rank0"] = gather_on_rank0
        d["rank"] = rank
        d["_src"] = source
        return self

    def _flags(self) -> "Optional[_DeviceFlags]":
        """The score kernel's below-threshold bytes (built from the source on first use), or None."""
        d = self.__dict__
        flags = d.get("_device_flags")
        if flags is None:
            src = d.get("_src")
            if src is not None and src.view.thresholds is not None:
                flags = d["_device_flags"] = src.device_flags()
        return flags

    def __getattr__(self, name: str):
        # reached only when normal lookup fails, i.e. for a mapping field that has not been built yet
        if name in _LAZY_FIELDS:
            src = self.__dict__.get("_src")
            if src is not None:
                d = self.__dict__
                value = d[name] = src.build(name, d)
                return value
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    def _materialise(self) -> None:
        for name in _LAZY_FIELDS:
            getattr(self, name)

    def __getstate__(self):
        """Plain fields only (every mapping built): what pickle / copy / multiprocessing queues carry."""
        self._materialise()
        state = {f.name: self.__dict__[f.name] for f in dataclasses.fields(self)}
        flags = self._flags()
        if flags is not None:
            state["_device_flags"] = flags
        return state

    def __setstate__(self, state) -> None:
        self.__dict__.update(state)

    def _ids(self, ranks) -> set:
        return {StragglerId(rank=r, node=self.rank_to_node[r]) for r in ranks}

    @staticmethod
    def _below(scores: Mapping[int, float], threshold: float):
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
        src = self.__dict__.get("_src")
        if src is not None and src.view.layout is not None:
            fast = src.flagged(self.rank_to_node, (gpu_rel_threshold, section_rel_threshold, gpu_indiv_threshold,
                                                   section_indiv_threshold))
            if fast is not None:
                return fast
        flags = self._flags()
        if flags is not None and flags.matches(
            gpu_rel_threshold, section_rel_threshold, gpu_indiv_threshold, section_indiv_threshold
        ):
            # thresholds equal the ones the score kernel was launched with: use its flag bytes
            src = self.__dict__.get("_src")
            ids = src.view.straggler_ids(self.rank_to_node) if src is not None else None
            gr, gi, sr, si = flags.decode(ids, src.view.memo() if src is not None else None)
            if ids is not None:
                return {"straggler_gpus_relative": gr, "straggler_gpus_individual": gi,
                        "straggler_sections_relative": sr, "straggler_sections_individual": si}
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
    """Below-threshold bytes written by the score kernel, with the thresholds they were computed for.
    Holds a private ndarray (never a view of the live result block, never a callable): picklable.  ``flags`` may
    be given as a ``_ScoreSource`` whose block is still in flight; it is resolved on first use."""

    __slots__ = ("thresholds", "flags", "ranks", "names", "cols", "S", "has_rel", "has_indiv")

    def __init__(self, thresholds, flags, ranks, names, cols, S, has_rel, has_indiv):
        self.thresholds = thresholds  # (gpu_rel, sec_rel, gpu_indiv, sec_indiv), a tuple of floats
        self.flags = flags  # ndarray [ranks, 2+2S] u8, or the _ScoreSource that will hold it
        self.ranks = ranks
        self.names = names
        self.cols = cols if isinstance(cols, dict) else dict(zip(names, cols))
        self.S = S
        self.has_rel = has_rel
        self.has_indiv = has_indiv

    def _array(self) -> np.ndarray:
        f = self.flags
        if isinstance(f, _ScoreSource):
            f = self.flags = f.ensure().flags
        return f

    def __getstate__(self):
        self._array()
        return {k: getattr(self, k) for k in self.__slots__}

    def __setstate__(self, state) -> None:
        for k, v in state.items():
            setattr(self, k, v)

    def matches(self, gpu_rel, sec_rel, gpu_indiv, sec_indiv) -> bool:
        return (float(gpu_rel), float(sec_rel), float(gpu_indiv), float(sec_indiv)) == self.thresholds

    def decode(self, ids=None, memo=None):
        """Flagged rows per score family: ``(gpu_rel, gpu_indiv, section_rel, section_indiv)``.  With ``ids`` (one
        ``StragglerId`` per row of the table) the results are the sets ``identify_stragglers`` returns; without, lists
        of ranks.  One ``any()`` in the common case (nothing flagged).  Otherwise per-column counts and first flagged
        rows come from two reductions over the table and only flagged columns are visited.  ``memo`` (a dict shared by
        the reports of one plan): a straggler usually stays one for many reports, so the result for an unchanged flag
        table is kept and handed out as fresh copies (a set copy does not re-hash its members)."""
        S = self.S
        wrap = set if ids is not None else list
        src = self.flags
        fb = src.flag_bytes() if isinstance(src, _ScoreSource) else src.tobytes()
        if fb.count(0) == len(fb):  # (bytes.count: one C loop, no numpy call on the common path)
            return wrap(), wrap(), {}, {}
        key = None
        if memo is not None and ids is not None:
            key = fb
            hit = memo.get("flags")
            if hit is not None and hit[0] == key and hit[1] is ids:
                gr, gi, sr, si = hit[2]
                return gr.copy(), gi.copy(), _copy_sets(sr), _copy_sets(si)
        f = self._array()
        who = ids if ids is not None else self.ranks
        cnt = f.sum(axis=0, dtype=np.int32).tolist()
        first = f.argmax(axis=0).tolist()

        def members(c):
            if cnt[c] == 1:
                return wrap((who[first[c]],))
            return wrap(who[int(r)] for r in np.flatnonzero(f[:, c]))

        gi = members(0) if (self.has_indiv and cnt[0]) else wrap()
        gr = members(1) if (self.has_rel and cnt[1]) else wrap()
        si: Dict[str, Any] = {}
        sr: Dict[str, Any] = {}
        if any(cnt[2:]):
            cols = self.cols
            # the report's own section order, as the reference's walk over its score dicts gives it
            if self.has_indiv:
                si = {n: members(2 + cols[n]) for n in self.names if cnt[2 + cols[n]]}
            if self.has_rel:
                sr = {n: members(2 + S + cols[n]) for n in self.names if cnt[2 + S + cols[n]]}
        if key is not None:
            memo["flags"] = (key, ids, (gr.copy(), gi.copy(), _copy_sets(sr), _copy_sets(si)))
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
        asynchronous: steady-state ring reports are only ENQUEUED (statistics kernel, collective, score kernel on
            the detector's stream) and the returned ``Report`` waits for them when it is first read, so the training
            loop never stalls on a report.  Not in the reference, whose ``generate_report`` is synchronous; two
            visible differences: ``generate_report_elapsed_time`` is the enqueue time, and a name that first appears
            on some rank enters the reports one report later (the name exchange is a host collective and runs at the
            start of the next ``generate_report``, when every rank is there).
    """

    def __init__(self, scores_to_compute, gather_on_rank0=True, pg=None, node_name="<notset>",
                 thresholds: Sequence[float] = _backend_mod.DEFAULT_THRESHOLDS, asynchronous: bool = False) -> None:
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
        # asynchronous reports (ring path only): generate_report_from_rings enqueues the report and returns a Report
        # that waits for the device on first read; the block in flight is settled before the next report starts
        self.asynchronous = bool(asynchronous)
        self._inflight: Optional[_PendingBlock] = None
        self.exchange_info: Dict[str, Any] = {}
        self._unreported_rows: list = []  # see take_unreported_rows
        self._prev_async_settled = True  # no asynchronous report of this generator is unaccounted for (see _settle_inflight)
        self._resync_pending = False  # the next ring report starts with the name sync (see Detector's lane)
        self._ring_reports = 0        # ring reports of this generator so far (collective: the same on every rank)
        self._may_defer_sync = False  # inside a ring report that is not the first (see _score_round)
        self._wr_cache: list = [None]  # this generator's remembered (default group, group, (world, rank)): dist_utils.world_and_rank

    # ---- pieces kept from the reference's host logic ----------------------------------------------
    @staticmethod
    def _filter_out_nccl_kernels(kernel_summaries):
        # collective kernels wait for peers, so their duration says nothing about this GPU
        return {k: v for k, v in kernel_summaries.items() if not is_collective_kernel(k)}

    def _shared_rank_to_node(self) -> Dict[int, str]:
        """The mapping handed to reports: the generator's own plain dict (see ``_maybe_gather_rank_to_node``); a generator
        whose mapping is still the initial defaultdict (nothing gathered yet) gets it converted once."""
        r2n = self.rank_to_node
        if type(r2n) is not dict:
            r2n = self.rank_to_node = dict(r2n)
        return r2n

    def _maybe_gather_rank_to_node(self) -> None:
        if self.rank_to_node:
            return
        # a PLAIN dict from here on, replaced (never edited) when it changes: every report of this generator is handed this
        # very object, and the per-plan caches of the straggler ids / flag decoder are keyed on its identity
        if self.gather_on_rank0:
            pairs = dist_utils.all_gather_object((self.rank, self.node_name), self.group)
            self.rank_to_node = dict(pairs)
        else:
            self.rank_to_node = {self.rank: self.node_name}

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
        """Collective, once: every rank calls this at the top of its first exchanging report.  Builds the in-stream
        exchange routes -- ``ncclAllGather`` on a communicator of our own (rccl_direct) and, inside one node, direct
        xGMI peer stores into IPC-mapped windows (peer_exchange) -- and keeps the one ``peer_exchange.choose`` picks."""
        if self._direct_tried or self.world_size == 1 or not self._exchanged():
            return
        self._direct_tried = True
        be = _backend_mod.get_backend()
        from . import peer_exchange, rccl_direct

        mode = peer_exchange.exchange_mode()
        # c10d (the default) on ANY rank keeps every rank on torch.distributed (the decision has to be the same everywhere:
        # building a communicator is collective)
        if not dist_utils.is_all_true(mode != "c10d", self.group):
            self._direct = None
            self.exchange_info = {"route": "torch.distributed all-gather on the job's own process group (NVRX_EXCHANGE=c10d, the "
                                           "default; rccl | peer | auto select an in-stream route)", "mode": "c10d"}
            if self.rank == 0:
                _LOG.info("straggler report exchange route: %s", self.exchange_info["route"])
                if self.asynchronous:
                    # torch.distributed's collective is issued by the host, between the statistics and the score kernel: a
                    # report on this route is complete when generate_report returns, whatever `asynchronous` says
                    _LOG.warning("nvrx straggler: asynchronous=True has no effect on the default exchange route (c10d: the report's "
                                 "all-gather is a torch.distributed call on the job's process group, and generate_report() waits for "
                                 "the scores); NVRX_EXCHANGE=rccl | peer | auto selects an in-stream route whose reports are only enqueued")
            return
        maker = getattr(be, "create_direct_exchange", None)  # a test backend may bring its own in-call exchange
        if maker is not None:
            self._direct = maker(self.group)
            return
        index = getattr(be.device, "index", None)
        # the exchange kernel gives up a little BEFORE the host's wait for the completion word does: a peer that is
        # later than the report timeout then surfaces as the exchange's own error (NaN rows, error word, the next
        # report unaffected) instead of a timed-out wait that retires the workspace under a still-spinning kernel
        peer_wait_s = 0.9 * _backend_mod.report_timeout_s() or 1e9
        rccl = rccl_direct.create(self.group, index)
        peer = None
        if mode != "rccl" and (rccl is not None or mode == "peer"):
            # windows need a group that can reach every rank's GPU: built next to the RCCL route (a gloo group whose
            # ranks share one GPU qualifies too when asked for explicitly: that is how the tests run it)
            peer = peer_exchange.create(self.group, index, peer_wait_s)
        self._direct, self.exchange_info = (peer_exchange.choose(self.group, be, rccl, peer, peer_wait_s)
                                            if (rccl or peer) else (None, {}))
        if self._direct is not None:
            self.exchange_info["route"] = getattr(self._direct, "route", "ncclAllGather on the detector's stream")
            ranks = getattr(self._direct, "comm_ranks", None)
            if ranks is not None:
                self.exchange_info["rccl_comm_ranks"] = ranks()  # ncclCommCount of the communicator the reports use
        if self.rank == 0:
            _LOG.info("straggler report exchange route: %s %s", self.exchange_info.get("route", "torch.distributed"),
                      {k: v for k, v in self.exchange_info.items() if k != "route"})

    def _exchange(self, be, ws):
        """The report's one collective: this rank's rows -> the [R, L] table, on the backend's stream."""
        if self._direct is not None:
            return self._direct.exchange(ws, be)
        with be.stream_context():  # the collective must queue behind the statistics kernel
            return dist_utils.all_gather_rows(ws.send, ws.table, self.group)

    def enqueue_only(self) -> bool:
        """Whether a steady-state ring report of this generator is only ENQUEUED (``asynchronous=True`` and nothing on the way
        needs the host: a single process, no exchange, or an in-stream exchange route).  On torch.distributed's route the
        collective is issued by the host between the two kernels and the report waits, whatever ``asynchronous`` says."""
        if not self.asynchronous:
            return False
        return self.world_size == 1 or not self._exchanged() or self._direct is not None or not self._direct_tried

    def take_unreported_rows(self) -> list:
        """Ring rows that held samples at the last ``generate_report_from_rings`` but were in no report (an asynchronous
        report that met a new name runs on the old tables): their samples belong to the next window.  Empty otherwise."""
        rows, self._unreported_rows = self._unreported_rows, []
        return rows

    def close(self) -> None:
        """Release the direct-exchange communicator (collective-free; safe to call more than once)."""
        # The remembered (default process group, group, (world, rank)) holds the ProcessGroup OBJECT.  A generator that outlives
        # destroy_process_group() would keep that object -- and gloo's worker threads, which its destructor joins -- alive
        # until the interpreter finalises; a worker that then releases the last tensor of its last collective asks for the GIL
        # of a finalising interpreter, CPython ends the thread with pthread_exit inside a noexcept C++ frame and the process
        # dies with "terminate called without an active exception" (the once-in-forty child abort of the reference's own
        # test_sections.py on a loaded host: profiles/r06_reftest_soak.txt has the native stack).
        self._wr_cache[0] = None
        if self._inflight is not None:
            try:
                self._settle_inflight()
            except Exception:  # noqa: BLE001  (shutdown path: a report that never completed must not mask it)
                self._inflight = None
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
        if resync_first and exchanged and self.world_size > 1:
            # the planned path already ran this report's first exchange and saw an incomplete flag:
            # every rank is now heading for the name sync, so join it before exchanging rows again.
            # (Only where rows ARE exchanged: a generator that scores this rank alone -- individual scores, no gather -- has
            # nobody heading anywhere; its own new names get their ids in the loop below.  An asynchronous generator of that
            # kind used to call the collective here, alone, the report after one of ITS sections first appeared: the rank hung
            # in all_gather_object while its peers trained on -- tools/soak_mp.py, profiles/r06af_soak_mp.txt.)
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
            if world > 1:
                self._check_exchange()
            if ws.meta[0] == 1:
                return ws, mapper
            if names_ok and world > 1 and self._may_defer_sync and self.enqueue_only():
                # Asynchronous reports on an in-stream route, and the names without ids are ANOTHER rank's: that rank only
                # enqueued this report (on its old tables, the flag in its row) and joins the name sync at the START of its next
                # report, when it settles this one -- it is not waiting inside this report, so neither a sync nor a second
                # exchange may happen here.  This rank got here because it left its cached plan in the very same report (its
                # occupied rows changed): it does what the planned path would have done -- keep the report, sync first next
                # time.  (It used to sync alone and exchange again: from then on its exchanges were paired with its peers'
                # NEXT ones, the last one with nobody -- tools/soak_mp.py, profiles/r06af_soak_mp.txt.)
                # (Not in a generator's FIRST ring report: no rank has a plan then, whoever flags is inside this report too.)
                self._resync_pending = True
                return ws, mapper
            # some rank (maybe this one) has names without ids: cold path, then go again
            mapper.sync_names(kernel_names, section_names)

    # ---- report assembly --------------------------------------------------------------------------
    def _assemble(self, ws, mapper, local_section_names: Sequence[str], section_summaries, kernel_summaries,
                  t_start_ns: int, local_ranks: int = 1, stats: Optional[np.ndarray] = None):
        """Report of the general (name-syncing / dict-input) path.  ``section_summaries`` / ``kernel_summaries``
        are either the caller's dicts (handed on as they are, reporting.py:541-542) or name -> ring row
        tables to be resolved against ``stats``."""
        S = ws.S
        if self.gather_on_rank0:
            if self.rank != 0:
                return None
            lo, hi = 0, ws.R
            ranks = range(ws.R)
            names = [mapper.get_section_name(i) for i in range(S)]
            cols = {n: i for i, n in enumerate(names)}
        else:
            lo = self.rank * local_ranks if self._exchanged() else 0
            hi = lo + local_ranks
            ranks = range(self.rank * local_ranks, (self.rank + 1) * local_ranks)
            names = list(local_section_names)
            cols = {n: mapper.get_section_id(n) for n in names}
        view = _View()
        view.S, view.ranks, view.names, view.cols = S, ranks, names, cols
        view.has_rel, view.has_indiv = self.is_computing_rel_scores, self.is_computing_indiv_scores
        view.section_rows = section_summaries if stats is not None else None
        view.kernel_rows = kernel_summaries if stats is not None else None
        view.layout, view.thresholds = None, self.thresholds
        src = _ScoreSource(view)
        src.scores = ws.scores[lo:hi].copy()
        src.flags = ws.flags[lo:hi].copy()
        src.stats = stats
        report = Report._from_device(src, self._shared_rank_to_node(), (time.perf_counter_ns() - t_start_ns) * 1e-6,
                                     self.gather_on_rank0, self.rank)
        if stats is None:
            # the caller's own summaries travel with the report, untouched
            report.__dict__["local_section_summaries"] = section_summaries
            report.__dict__["local_kernel_summaries"] = kernel_summaries
        return report

    # ---- steady-state plan of the ring path ---------------------------------------------------------
    class _RingPlan:
        """Everything about a ring report that only changes when names, ids or topology change."""

        __slots__ = ("key", "ws", "mapper", "snames", "knames", "names", "cols", "ranks", "row_lo", "row_hi",
                     "rows_used", "stats_needed", "section_rows", "kernel_rows", "fused", "view")

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
        # the report's local summaries are those of this process' FIRST logical rank (rank 0 of a folded job): its rows are
        # the first rows_used statistics rows, whatever the number of logical ranks -- nothing else is forwarded or copied
        plan.stats_needed = rings.rows_used
        plan.section_rows, plan.kernel_rows = section_rows, kernel_rows
        plan.fused = hasattr(rings, "report_fused")  # the HIP engine; the CPU test backend takes the stepwise route
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
            g = mapper.kernel_name_to_id.get(name, -1) if not is_collective_kernel(name) else -1
            rings.configure(row, 1, g)
        for name, row in rings.section_row_names.items():
            g = mapper.section_name_to_id.get(name)
            rings.configure(row, 0, K + g if g is not None else -1)
        plan.ws.send_initialised = False
        plan.view = self._view_of(plan)
        self._ring_gid_state = None  # the general path must re-derive its own view if it runs next
        return plan

    def _settle_inflight(self) -> bool:
        """Wait for the asynchronous report still in flight (if any) before its workspace is touched again.
        Returns True when that report's exchange showed a rank with names that have no id yet: every rank sees the
        same table, so every rank learns it here, at the same point of the program, and goes through the name sync."""
        pend = self._inflight
        if pend is None:
            return False
        self._inflight = None
        names_word = pend.complete()  # a poll; the block is copied only if / when somebody reads or outlives it
        self._prev_async_settled = True  # observed, not assumed: the next report tells the library so (prev_settled)
        self._check_exchange()
        return names_word != 1

    def _check_exchange(self) -> None:
        check = getattr(self._direct, "check", None)  # the peer-window route reports peers that never arrived
        if check is not None:
            check()

    def _view_of(self, plan) -> _View:
        """The part of a report that every report of ``plan`` shares (built once per plan)."""
        ws = plan.ws
        v = _View()
        v.S, v.ranks, v.names, v.cols = ws.S, plan.ranks, plan.names, plan.cols
        v.has_rel, v.has_indiv = self.is_computing_rel_scores, self.is_computing_indiv_scores
        v.section_rows, v.kernel_rows = plan.section_rows, plan.kernel_rows
        v.layout = (ws._off_scores, ws._off_flags, ws._off_stats, ws.R, ws.W, plan.row_lo, plan.row_hi, plan.stats_needed)
        v.thresholds = self.thresholds
        return v

    def _report_from_plan(self, plan, rings, t0, order_after=None, names_ok: bool = True):
        """The steady-state report: ONE C call (statistics kernel, the collective, score kernel, wait), one host
        copy of the result block, mappings built on first read.  Asynchronous generators only enqueue."""
        be = _backend_mod.get_backend()
        ws = plan.ws
        multi = self.world_size > 1 and self._exchanged()
        fused = plan.fused and (not multi or self._direct is not None)
        if fused:
            # ONE C call: flush -> statistics kernel -> [ncclAllGather] -> score kernel -> completion word
            wait = not self.asynchronous
            seq = rings.report_fused(ws, plan.rows_used, plan.stats_needed, self.is_computing_indiv_scores,
                                     self.is_computing_rel_scores, self.thresholds, self._direct if multi else None,
                                     names_ok=names_ok, wait=wait, order_after=order_after if multi else None,
                                     resident=not (multi and getattr(self._direct, "shared_device", False)
                                                   and os.environ.get("NVRX_DEBUG_RESIDENT_SHARED_OK", "0") in ("", "0")),
                                     prev_settled=self._prev_async_settled)
            if not wait:
                self._prev_async_settled = False  # until somebody has seen THIS report complete
                pend = self._inflight = _PendingBlock(be, ws, seq)
                if self.gather_on_rank0 and self.rank != 0:
                    return None
                return Report._from_device(
                    _ScoreSource(plan.view, pend),
                    self._shared_rank_to_node(),
                    (time.perf_counter_ns() - t0) * 1e-6, self.gather_on_rank0, self.rank)
        elif multi:
            rings.report_local(ws, True, rows_active=plan.rows_used)
            table = self._exchange(be, ws)
            be.score(ws, table, self.is_computing_indiv_scores, self.is_computing_rel_scores, self.thresholds,
                     wait=True, stats_rows=plan.stats_needed)
        else:
            rings.report_local(ws, True, rows_active=plan.rows_used)
            be.score(ws, ws.send, self.is_computing_indiv_scores, self.is_computing_rel_scores, self.thresholds,
                     wait=True, stats_rows=plan.stats_needed)
        if fused:
            # a resident score kernel forwards the statistics rows AFTER the scores: whatever this call returns or
            # raises, the next user of the workspace must see them landed (meta[5]) before it enqueues anything that
            # writes them
            mark = getattr(ws, "mark_live", None)  # (the CPU checker backend of the tests has no deferred rows)
            if mark is not None:
                mark(ws.seq)
        if multi:
            self._check_exchange()  # a peer that never arrived: this report's scores are invalid, the next one is not
        if ws.meta[0] != 1:
            return False  # another rank met a new name: fall back to the general (name-syncing) path
        if self.gather_on_rank0 and self.rank != 0:
            return None
        if fused:
            # nothing is copied now: scores / flags / statistics leave the block when the report is first read, or when
            # the workspace is about to be reused and the report is still held
            pending = _LiveBlock(be, ws, ws.seq, plan.stats_needed)
            ws.attach(pending)  # the workspace collects it (if still held) before the block is reused by anybody
        else:
            pending = ws.host_block()
        return Report._from_device(
            _ScoreSource(plan.view, pending),
            self._shared_rank_to_node(),
            (time.perf_counter_ns() - t0) * 1e-6, self.gather_on_rank0, self.rank)

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
        if self._inflight is not None:
            self._settle_inflight()
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
                                   local_ranks: int = 1, order_after: Optional[int] = None):
        """Report straight from the device rings: statistics kernel -> exchange -> score kernel.

        ``section_rows`` / ``kernel_rows`` map the names that hold samples this window to their ring
        rows.  Replaces ``_get_section_summaries`` + ``_get_kernel_summaries`` + ``generate_report``
        of the reference (straggler.py:236-239) without materialising per-section Python objects.
        ``order_after``: raw ``hipStream_t`` of the caller's current stream; a multi-rank report is ordered after the
        work already enqueued there (the collectives of the training step) on the device, without a host wait.
        """
        t0 = time.perf_counter_ns()
        self.world_size, self.rank = dist_utils.world_and_rank(self.group, self._wr_cache)
        if not self._direct_tried:
            self._maybe_create_direct_exchange()
        # steady state: same name tables as last time -> run the cached plan
        key = (id(section_rows), len(section_rows), id(kernel_rows), len(kernel_rows), self.name_mapper.version,
               self._private_mapper.version, self.world_size, self.rank, rings.rows_used, local_ranks, id(rings))
        first_ring_report = self._ring_reports == 0
        self._ring_reports += 1
        resync_first, self._resync_pending = self._resync_pending, False  # (left by a report that saw an "ids missing" table)
        if resync_first:
            self._ring_plan = None  # every rank is heading for the name sync: no cached plan runs before it
        plan = self._ring_plan
        if self._inflight is not None and self._settle_inflight():
            # the previous (asynchronous) report's exchange carried an "ids missing" flag: every rank is here now
            self._ring_plan = plan = None
            resync_first = True
        if plan is not None and plan.key == key:
            try:
                out = self._report_from_plan(plan, rings, t0, order_after)
            except Exception:
                self._ring_plan = None  # e.g. a timed-out wait retired the plan's workspace: never run it again
                raise
            if out is not False:
                return out
            self._ring_plan = None
            resync_first = True  # some OTHER rank met a new name during this report's exchange
        elif (plan is not None and self.enqueue_only() and plan.fused and plan.key[6:8] == key[6:8] and plan.key[9:] == key[9:]
              and not plan.mapper.has_all_names(list(kernel_rows.keys()), list(section_rows.keys()))):
            # asynchronous + a name this rank has no id for: the other ranks will not wait inside this report, so the
            # name exchange cannot happen now.  Run the OLD plan (the new rows are not exchanged yet) with the
            # "ids missing" flag in this rank's row; every rank meets it when it settles this report and they all
            # sync names at the start of the next one.  The rows of the new names are in nobody's report this time: the caller
            # keeps their samples for the next window instead of dropping them with the rest (take_unreported_rows).
            known_k, known_s = plan.kernel_rows, plan.section_rows
            self._unreported_rows = ([r for n, r in kernel_rows.items() if n not in known_k and not is_collective_kernel(n)]
                                     + [r for n, r in section_rows.items() if n not in known_s])
            return self._report_from_plan(plan, rings, t0, order_after, names_ok=False)
        kernel_rows = {k: r for k, r in kernel_rows.items() if not is_collective_kernel(k)} if any(
            is_collective_kernel(k) for k in kernel_rows) else kernel_rows
        self._maybe_gather_rank_to_node()
        if local_ranks > 1 and not getattr(self, "_rank_to_node_folded", False):
            # folded runs: every logical rank inherits the node of the process that holds it
            per_proc = dict(self.rank_to_node)
            self.rank_to_node = {p * local_ranks + q: node for p, node in per_proc.items() for q in range(local_ranks)}
            self._rank_to_node_folded = True
        knames, snames = list(kernel_rows.keys()), list(section_rows.keys())
        rows_used = rings.rows_used
        total_rows = rings.local_ranks * rings.rows_per_rank
        stats_needed = rows_used  # (see _build_ring_plan)

        def fill_send(ws, mapper, names_ok):
            state = (id(mapper), mapper.version, rows_used, id(ws))
            if self._ring_gid_state != state:
                # ids changed (cold): re-point every ring row at its slot of the exchange row
                K = ws.K
                for name, row in rings.kernel_row_names.items():
                    g = mapper.kernel_name_to_id.get(name, -1) if not is_collective_kernel(name) else -1
                    rings.configure(row, 1, g)
                for name, row in rings.section_row_names.items():
                    g = mapper.section_name_to_id.get(name)
                    rings.configure(row, 0, K + g if g is not None else -1)
                ws.send_initialised = False
                self._ring_gid_state = state
            rings.report_local(ws, names_ok, rows_active=rows_used)

        self._may_defer_sync = not first_ring_report
        try:
            ws, mapper = self._score_round(knames, snames, fill_send, local_ranks=local_ranks, stats_rows=total_rows,
                                           stats_rows_used=stats_needed, resync_first=resync_first)
        finally:
            self._may_defer_sync = False
        report = self._assemble(ws, mapper, snames, dict(section_rows), dict(kernel_rows), t0, local_ranks=local_ranks,
                                stats=ws.stats[:stats_needed].copy())
        # names are settled now: the next report with the same tables takes the planned path
        key = (id(section_rows), len(section_rows), id(kernel_rows), len(kernel_rows), self.name_mapper.version,
               self._private_mapper.version, self.world_size, self.rank, rings.rows_used, local_ranks, id(rings))
        self._ring_plan = self._build_ring_plan(key, rings, section_rows, kernel_rows, local_ranks)
        return report
