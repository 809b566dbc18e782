"""Compute backend of the straggler path: the HIP engine, and nothing else.

``get_backend()`` returns the process-wide :class:`HipBackend`.  Constructing it requires the in-tree
``libnvrx_straggler_hip.so`` and a visible MI355X; otherwise it raises -- there is no CPU fallback in
the product.  ``set_backend()`` exists so the CPU-only unit tests can inject their own checker
backend (built on ``oracle/``, which lives outside this package) to exercise the host-side logic
(name mapping, exchange, report assembly, Detector plumbing) on a box without a GPU.

Data layout in HBM (all f32 unless noted; see include/nvrx_straggler.h):

* rings   ``[local_ranks * rows_per_rank][row_stride]``  one timing row per section / GPU-timed region
* stats   ``[rows][8]``        MIN MAX MED AVG STD NUM WEIGHT pad
* send    ``[local_ranks][L]`` this GPU's exchange rows, ``L = 2(K+S) + K + 1``
* table   ``[R][L]``           all ranks' rows after the all-gather
* scores  ``[R][2+2S]``, flags ``[R][2+2S]`` u8, meta ``[4]`` u32

``meta | scores | flags | stats`` live in ONE device allocation mirrored by ONE pinned host buffer, so
a report needs a single small D2H copy.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native

DEFAULT_THRESHOLDS = (0.75, 0.75, 0.75, 0.75)  # gpu_rel, section_rel, gpu_indiv, section_indiv



def report_timeout_s() -> float:
    """How long a report waits for its completion word.  The score kernel is queued behind the report's
    collective, so this bounds how late the slowest PEER may reach ``generate_report`` (first-report RCCL set-up,
    a checkpoint or evaluation on one rank, an actual straggler).  The reference simply blocks in its collective
    until the process group's timeout; the default here is c10d's 30 minutes.  ``NVRX_REPORT_TIMEOUT_S`` overrides
    it, ``0`` waits for ever."""
    t = _timeout_cache[0]
    if t is None:
        t = refresh_report_timeout()
    return t


_timeout_cache = [None]  # read once: a report (and the wait for the previous asynchronous one) does not look at the environment


def refresh_report_timeout() -> float:
    """Re-read ``NVRX_REPORT_TIMEOUT_S`` (a process that changes it after its first report calls this)."""
    try:
        t = float(os.environ.get("NVRX_REPORT_TIMEOUT_S", "1800"))
    except ValueError:
        t = 1800.0
    _timeout_cache[0] = t
    return t


_backend = None
_backend_lock = threading.Lock()

# torch.cuda.current_stream() builds a Stream object through several Python layers (~3 us); the raw
# handle is one C call.  Private API, so fall back to the public one if it ever goes away.
_raw_stream_fn = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_current_stream(device_index: int) -> int:
    if _raw_stream_fn is not None:
        return _raw_stream_fn(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def set_backend(backend) -> None:
    """Install a backend object (tests only; pass ``None`` to go back to the HIP engine)."""
    global _backend
    with _backend_lock:
        _backend = backend


_NO_DEVICE = ("nvrx_straggler: no HIP device is visible (torch.cuda.is_available() is False). "
              "The MI355X straggler-scoring path has no CPU fallback.")


def require_engine() -> None:
    """Fail NOW where the engine cannot run (library not built, no HIP device) -- without creating it: the engine binds to
    the device that is current when it is first used, which a script may select only after ``Detector.initialize``."""
    if _backend is not None:
        return
    _native.load()
    if not torch.cuda.is_available():
        raise RuntimeError(_NO_DEVICE)


def get_backend():
    """The active backend; creates the HIP engine on first use and fails loudly if it cannot."""
    global _backend
    if _backend is None:
        with _backend_lock:
            if _backend is None:
                _backend = HipBackend()
    return _backend


def _nobody():
    """A dead weak reference."""
    return None


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class ResultBlock:
    """One result block in PINNED, device-mapped host memory: ``meta | scores | flags | stats``.  The kernels store their
    results straight into it and the score kernel publishes a sequence word last (``meta[4]``; ``meta[5]`` for the
    statistics rows, which a resident score kernel forwards after the scores), so a report needs no D2H copy and no
    stream synchronisation.  A workspace owns TWO of them and alternates: the block of report n is not written again
    before report n+2, so a caller that still holds report n while it asks for n+1 (``report = generate_report()`` in
    a loop does exactly that) costs the next report nothing -- with one block the next report first had to wait for
    the previous one's statistics rows and copy 16 KB out of the way."""

    def __init__(self, ws: "Workspace", backend: "HipBackend"):
        R, W, stats_rows = ws.R, ws.W, ws.stats_rows
        self._off_meta = 0
        self._off_scores = self._off_meta + _align(_native.META_WORDS * 4)
        self._off_flags = self._off_scores + _align(R * W * 4)
        self._off_stats = self._off_flags + _align(R * W)
        self.nbytes = self._off_stats + _align(max(stats_rows, 1) * _native.STATS_STRIDE * 4)
        h_ptr, d_ptr = ctypes.c_void_p(), ctypes.c_void_p()
        _native.check(backend.lib.nvrx_host_alloc(ctypes.byref(h_ptr), ctypes.byref(d_ptr), self.nbytes))
        self._lib = backend.lib
        self._backend = backend
        self.h_ptr, self.d_ptr = h_ptr.value, d_ptr.value
        host = np.frombuffer((ctypes.c_uint8 * self.nbytes).from_address(self.h_ptr), dtype=np.uint8)
        self._host = host
        self.stats = host[self._off_stats : self._off_stats + stats_rows * 32].view(np.float32).reshape(stats_rows, _native.STATS_STRIDE)
        self.meta = host[self._off_meta : self._off_meta + _native.META_WORDS * 4].view(np.uint32)
        # the same words without numpy (a scalar read of the pinned block through numpy costs a report microseconds when cold)
        self.meta_words = (ctypes.c_uint32 * _native.META_WORDS).from_address(self.h_ptr + self._off_meta)
        self.scores = host[self._off_scores : self._off_scores + R * W * 4].view(np.float32).reshape(R, W)
        self.flags = host[self._off_flags : self._off_flags + R * W].reshape(R, W)
        self.h_stats_dst = self.d_ptr + self._off_stats
        self.d_meta = self.d_ptr + self._off_meta
        self.d_scores = self.d_ptr + self._off_scores
        self.d_flags = self.d_ptr + self._off_flags
        self.h_seq = self.h_ptr + self._off_meta + 16  # meta[4]: scores / flags / meta have landed
        self.h_seq2 = self.h_ptr + self._off_meta + 20  # meta[5]: the statistics rows have landed
        # descriptor of the one-call report (nvrx_report) into THIS block; the library advances desc.seq itself
        d = self.desc = _native.ReportDesc()
        d.R, d.K, d.S = ws.R, ws.K, ws.S
        d.names_ok, d.rows_active, d.do_indiv, d.do_rel, d.stats_rows = 1, 0, 1, 1, 0
        for i, t in enumerate(DEFAULT_THRESHOLDS):
            d.thresholds[i] = t
        d.d_stats, d.d_send, d.d_table = ws.d_stats, ws.send_ptr, ws.table_ptr
        d.d_scores, d.d_flags, d.d_meta, d.d_stats_dst = self.d_scores, self.d_flags, self.d_meta, self.h_stats_dst
        d.d_done_counter = ws.d_counter
        d.allgather_fn, d.comm, d.send_count = None, None, ws.local_ranks * ws.L
        d.seq = 0
        d.h_seq_word = self.h_seq
        d.timeout_s = report_timeout_s()
        self.desc_ref = ctypes.byref(d)
        self.desc_key = None
        # this block's last report may still be "live" (read lazily by a Report): see attach() / settle()
        self._live = None
        self._live_seq = 0

    def free(self) -> None:
        if self.h_ptr:
            self._lib.nvrx_host_free(self.h_ptr)
            self.h_ptr = None

    def attach(self, live) -> None:
        """``live`` (reporting._LiveBlock) reads this block lazily; ``settle()`` collects it before the block is reused."""
        self._live = weakref.ref(live)
        self._live_seq = live.seq

    def mark_live(self, seq: int) -> None:
        """A one-call report with sequence number ``seq`` ran on this block and nobody may be holding it (a rank that
        returns ``None``, a report abandoned for a name sync): ``settle()`` still has to see its statistics rows land."""
        self._live = _nobody
        self._live_seq = seq

    def settle(self) -> None:
        """Called before anything is enqueued that writes this block again: its last report's statistics rows must have
        landed (a resident score kernel forwards them after the scores), and if somebody still holds that report its data
        is copied out now."""
        ref = self._live
        if ref is not None:
            self._live = None
            live = ref()
            if live is not None:
                live.detach()
            elif self.meta_words[5] != self._live_seq:
                self._backend.wait_seq(None, self._live_seq, stats=True, block=self)

    def host_block(self) -> np.ndarray:
        """A private copy of the whole block (the block itself is overwritten two reports later)."""
        return self._host.copy()

    def host_head(self) -> np.ndarray:
        """A private copy of meta | scores | flags (valid once meta[4] shows the report's sequence number)."""
        return self._host[: self._off_stats].copy()

    def host_head_bytes(self) -> bytes:
        """The same as ``bytes``: one memcpy out of the pinned block, no numpy object involved."""
        return ctypes.string_at(self.h_ptr, self._off_stats)

    def host_stats(self, rows: int) -> np.ndarray:
        """A private copy of the first ``rows`` statistics rows (valid once meta[5] shows the sequence number)."""
        return self.stats[:rows].copy()


class Workspace:
    """Buffers of one report shape (R ranks, K kernel ids, S section ids): the exchange rows and the gathered table in
    device memory, and two result blocks (``ResultBlock``) that successive reports alternate between.  ``ws.meta /
    scores / flags / stats`` are those of the CURRENT block, i.e. of the report that ran last."""

    def __init__(self, backend: "HipBackend", R: int, K: int, S: int, local_ranks: int, stats_rows: int):
        self.R, self.K, self.S = R, K, S
        self.local_ranks = local_ranks
        self.stats_rows = stats_rows
        self.L = _native.table_len(K, S)
        self.W = _native.score_len(S)
        dev = backend.device
        self.send = torch.empty((local_ranks, self.L), dtype=torch.float32, device=dev)
        self.table = torch.empty((R, self.L), dtype=torch.float32, device=dev) if R != local_ranks else self.send
        self.send_initialised = False
        # statistics are produced in device memory and forwarded to the host block by the score kernel
        self.stats_dev = torch.zeros((max(stats_rows, 1), _native.STATS_STRIDE), dtype=torch.float32, device=dev)
        self.d_stats = self.stats_dev.data_ptr()
        self.done_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_counter = self.done_counter.data_ptr()
        self.seq = 0  # the one sequence every report on this workspace draws from, whichever block and route it takes
        self.send_ptr = self.send.data_ptr()
        self.table_ptr = self.table.data_ptr()
        self._backend = backend
        self.blocks = (ResultBlock(self, backend), ResultBlock(self, backend))
        self.block = self.blocks[0]
        self._cur = 0
        b = self.block
        self._off_scores, self._off_flags, self._off_stats, self.nbytes = b._off_scores, b._off_flags, b._off_stats, b.nbytes

    def __del__(self):  # pragma: no cover
        try:
            for b in self.blocks:
                b.free()
        except Exception:
            pass

    # ---- the current block's views / addresses ----------------------------------------------------------------
    meta = property(lambda self: self.block.meta)
    scores = property(lambda self: self.block.scores)
    flags = property(lambda self: self.block.flags)
    stats = property(lambda self: self.block.stats)
    desc = property(lambda self: self.block.desc)
    h_seq = property(lambda self: self.block.h_seq)
    h_seq2 = property(lambda self: self.block.h_seq2)
    d_meta = property(lambda self: self.block.d_meta)
    d_scores = property(lambda self: self.block.d_scores)
    d_flags = property(lambda self: self.block.d_flags)
    h_stats_dst = property(lambda self: self.block.h_stats_dst)

    def flip(self) -> ResultBlock:
        """Make the OTHER block the current one for the report about to be enqueued (settling whatever its previous
        report -- two reports ago -- left behind) and return it."""
        nxt = self.blocks[1 - self._cur]
        if nxt._live is not None:
            nxt.settle()
        self._cur = 1 - self._cur
        self.block = nxt
        return nxt

    def attach(self, live) -> None:
        self.block.attach(live)

    def mark_live(self, seq: int) -> None:
        self.block.mark_live(seq)

    def settle(self) -> None:
        """Both blocks: nothing of this workspace is in flight or lazily referenced afterwards."""
        for b in self.blocks:
            if b._live is not None:
                b.settle()

    def host_block(self) -> np.ndarray:
        return self.block.host_block()

    def host_head(self) -> np.ndarray:
        return self.block.host_head()

    def host_head_bytes(self) -> bytes:
        return self.block.host_head_bytes()

    def host_stats(self, rows: int) -> np.ndarray:
        return self.block.host_stats(rows)

    def set_send_row(self, lr: int, row: np.ndarray) -> None:
        """Host-packed exchange row (dict-input path)."""
        self.send[lr].copy_(torch.from_numpy(row), non_blocking=False)
        self.send_initialised = True


def _stream_priority() -> int:
    """Priority of the detector's own streams (``NVRX_STREAM_PRIORITY=high|normal``, default high): a report that runs BESIDE
    the training stream -- asynchronous reports, synchronous ones that cannot be re-homed -- competes with GEMMs for compute
    units; on a high-priority queue its few workgroups are dispatched ahead of the GEMM's next ones instead of behind them
    (nothing running is preempted).  Same-box A/B, three alternating rounds of config #4: a report every step costs the step
    1.15 / 1.15 / 0.80 % at normal priority, 0.96 / 0.83 / 0.77 % at high (asynchronous); 3.43 / 3.10 / 2.93 against
    3.01 / 3.00 / 2.91 % (synchronous); no effect on an idle GPU (profiles/r04p_stream_priority.txt).

    That is the default of a SINGLE-process job only.  In a multi-rank job the detector's stream also carries the report's
    RCCL all-gather, beside the job's own collectives on other streams and communicators; which of two communicators' kernels
    is dispatched first is better left to the order they were enqueued in until a run across real GPUs says otherwise, so the
    default there stays normal (``NVRX_STREAM_PRIORITY=high`` asks for it by name)."""
    want = os.environ.get("NVRX_STREAM_PRIORITY", "").strip().lower()
    if not want:
        from . import ktrace as _ktrace  # (the launcher's job size: WORLD_SIZE, else srun's / mpirun's / PMI's)

        want = "normal" if _ktrace._job_size()[0] > 1 else "high"  # (the library applies the same rule to the resident scorer's stream)
    return 0 if want in ("normal", "0", "off") else -1


class HipBackend:
    """MI355X engine: owns the side stream the report runs on and the per-shape workspaces."""

    name = "hip"

    def __init__(self, device: Optional[int] = None):
        self.lib = _native.load()
        if not torch.cuda.is_available():
            raise RuntimeError(_NO_DEVICE)
        index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", index)
        # the report pipeline runs on its own stream so it never serialises with the training stream
        self.stream = torch.cuda.Stream(device=self.device, priority=_stream_priority())
        self._workspaces = {}
        self._retired = []
        self._thr = (ctypes.c_double * 4)(*DEFAULT_THRESHOLDS)
        self._thr_src = DEFAULT_THRESHOLDS
        self._stream_handle = self.stream.cuda_stream

    @property
    def stream_handle(self) -> int:
        return self._stream_handle

    def stream_context(self):
        return torch.cuda.stream(self.stream)

    def current_stream_handle(self) -> int:
        """hipStream_t of the stream user code is currently launching on (for region timing)."""
        return _raw_current_stream(self.device.index)

    def workspace(self, R: int, K: int, S: int, local_ranks: int = 1, stats_rows: int = 0) -> Workspace:
        key = (R, K, S, local_ranks, stats_rows)
        ws = self._workspaces.get(key)
        if ws is None:
            if len(self._workspaces) > 8:  # shapes only change when new names appear
                self._workspaces.clear()
            with torch.cuda.device(self.device):
                ws = Workspace(self, R, K, S, local_ranks, stats_rows)
                # the buffers were zero-filled on torch's current stream; the report runs on ours
                torch.cuda.current_stream().synchronize()
            self._workspaces[key] = ws
        return ws

    def make_rings(self, local_ranks: int, rows_per_rank: int, ring_cap: int) -> "HipRings":
        return HipRings(self, local_ranks, rows_per_rank, ring_cap)

    def send_init(self, ws: Workspace) -> None:
        _native.check(self.lib.nvrx_send_init(ws.send.data_ptr(), ws.local_ranks, ws.K, ws.S, self.stream_handle))
        ws.send_initialised = True

    def score(self, ws: Workspace, table: torch.Tensor, do_indiv: bool, do_rel: bool,
              thresholds: Sequence[float] = DEFAULT_THRESHOLDS, wait: bool = True,
              stats_rows: Optional[int] = None) -> None:
        """Score kernel, then spin on the completion word it publishes into the pinned result block
        (two C calls, no torch dispatch, no D2H copy, no stream sync); on return
        ``ws.scores/flags/meta/stats`` hold this report's values."""
        if thresholds is not self._thr_src:
            for i in range(4):
                self._thr[i] = float(thresholds[i])
            self._thr_src = thresholds
        lib = self.lib
        blk = ws.flip()  # this report's result block; the previous report's block stays readable for one more report
        nrows = ws.stats_rows if stats_rows is None else min(stats_rows, ws.stats_rows)
        table_ptr = ws.table_ptr if table is ws.table else (ws.send_ptr if table is ws.send else table.data_ptr())
        ws.seq = (ws.seq % 0x7FFFFFFF) + 1  # one sequence for both report routes and both blocks
        rc = lib.nvrx_score(table_ptr, ws.R, ws.K, ws.S, int(do_indiv), int(do_rel), self._thr,
                            blk.d_scores, blk.d_flags, blk.d_meta, ws.d_counter, ws.seq,
                            ws.d_stats, blk.h_stats_dst, nrows, self._stream_handle)
        if rc < 0:
            _native.check(rc)
        if wait:
            timeout = report_timeout_s()
            rc = lib.nvrx_poll_u32(blk.h_seq, ws.seq, timeout if timeout > 0 else 1e30)
            if rc < 0:
                self.retire_workspace(ws)
                _native.check(rc)

    def wait_seq(self, ws: Optional[Workspace], seq: int, stats: bool = False, block: Optional[ResultBlock] = None) -> None:
        """Block until the report that was enqueued with sequence number ``seq`` has published its results into
        ``block`` (default: the workspace's current one); ``stats=True``: its statistics rows, which a resident score
        kernel forwards after the scores."""
        blk = block if block is not None else ws.block
        timeout = report_timeout_s()
        rc = self.lib.nvrx_poll_u32(blk.h_seq2 if stats else blk.h_seq, seq, timeout if timeout > 0 else 1e30)
        if rc < 0:
            if ws is None:
                ws = next((w for w in self._workspaces.values() if blk in w.blocks), None)
            if ws is not None:
                self.retire_workspace(ws)
            _native.check(rc)

    def retire_workspace(self, ws: Workspace) -> None:
        """After a timed-out wait the kernels of ``ws`` may still be queued behind a collective: the workspace
        must neither be reused (stale sequence / ticket state) nor freed under them, so it is parked for good."""
        for key, val in list(self._workspaces.items()):
            if val is ws:
                del self._workspaces[key]
        self._retired.append(ws)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def row_stats(self, samples: torch.Tensor, counts: torch.Tensor, kinds: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Stateless statistics operator on caller tensors ([rows, stride] f32, [rows] u32/i32)."""
        rows, stride = samples.shape
        # the inputs were produced on the caller's current stream; the output is allocated under ours
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            stats = torch.empty((rows, _native.STATS_STRIDE), dtype=torch.float32, device=samples.device)
            _native.check(
                self.lib.nvrx_row_stats(samples.data_ptr(), counts.data_ptr(), kinds.data_ptr() if kinds is not None else None,
                                        rows, stride, stats.data_ptr(), self.stream_handle)
            )
        self.stream.synchronize()
        return stats


class HipRings:
    """Device ring buffers + pinned staging + hipEvent timing (one ``nvrx_ctx``)."""

    def __init__(self, backend: HipBackend, local_ranks: int, rows_per_rank: int, ring_cap: int):
        self.backend = backend
        self.lib = backend.lib
        self.local_ranks = local_ranks
        self.rows_per_rank = rows_per_rank
        self.ring_cap = ring_cap
        ctx = ctypes.c_void_p()
        _native.check(self.lib.nvrx_ctx_create(backend.device.index, local_ranks, rows_per_rank, ring_cap, ctypes.byref(ctx)))
        self.ctx = ctx
        _native.check(self.lib.nvrx_ctx_set_stream(ctx, backend.stream_handle))
        self._counts_buf = np.zeros(64, dtype=np.int32)
        self._rows_used = 0
        #: every name that ever got a ring row (rows are never recycled; a reset only empties them)
        self.section_row_names = {}
        self.kernel_row_names = {}

    # ---- lifecycle -----------------------------------------------------------------------------
    def close(self) -> None:
        if self.ctx is not None:
            from . import ktrace as _ktrace

            _ktrace.detach_sink_of(self.ctx.value)  # (the kernel tracer's thread must not append to a destroyed context)
            self.lib.nvrx_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- rows ----------------------------------------------------------------------------------
    @property
    def rows_used(self) -> int:
        """Rows handed out so far, as far as the host has been TOLD: the count lives in the library (``nvrx_row_alloc``),
        because the per-kernel tracer's thread takes rows for new kernel keys on its own (``ktrace_sink``); the host's copy
        moves when it allocates a row itself and when the tracer's profiler learns new keys (``note_rows_used``, at
        report time) -- a report covers the rows whose names are known, and reads no C state for it."""
        return self._rows_used

    def note_rows_used(self) -> int:
        """Re-read the library's count (rows the tracer's thread has taken since)."""
        self._rows_used = max(self._rows_used, self.lib.nvrx_ctx_info(self.ctx, 8))
        return self._rows_used

    def alloc_row(self, kind: int = _native.KIND_SECTION) -> int:
        row = self.lib.nvrx_row_alloc(self.ctx, kind)
        if row >= self._rows_used:
            self._rows_used = row + 1
        if row < 0:
            if row == _native.ERR_RANGE:
                raise RuntimeError(
                    f"straggler rings are full: {self.rows_per_rank} timing rows per rank "
                    "(raise max_rows in Detector.initialize)"
                )
            _native.check(row)
        return row

    def row_for(self, kind: int, name: str) -> int:
        """Ring row of a section (kind 0) / GPU-timed region (kind 1), allocated on first use."""
        table = self.kernel_row_names if kind == _native.KIND_KERNEL else self.section_row_names
        row = table.get(name)
        if row is None:
            row = self.alloc_row(kind)
            table[name] = row
        return row

    def ktrace_sink(self):
        """``(ctx, push, row_alloc)`` as plain addresses: what ``nvrx_ktrace_set_sink`` needs to append kernel durations to
        these rings from the tracer's thread (include/nvrx_ktrace.h)."""
        cast = ctypes.cast
        return (self.ctx.value, cast(self.lib.nvrx_sink_push, ctypes.c_void_p).value,
                cast(self.lib.nvrx_sink_row_alloc, ctypes.c_void_p).value)

    def configure(self, row: int, kind: int, gid: int, lr: Optional[int] = None) -> None:
        lrs = range(self.local_ranks) if lr is None else (lr,)
        for q in lrs:
            _native.check(self.lib.nvrx_row_configure(self.ctx, q * self.rows_per_rank + row, kind, gid))

    def push(self, row: int, value: float, lr: int = 0) -> None:
        rc = self.lib.nvrx_ring_push(self.ctx, lr * self.rows_per_rank + row, value)
        if rc < 0:
            _native.check(rc)

    def push_many(self, row: int, values, lr: int = 0) -> None:
        a = np.ascontiguousarray(values, dtype=np.float32)
        _native.check(self.lib.nvrx_ring_push_many(self.ctx, lr * self.rows_per_rank + row, a.ctypes.data, a.size))

    def push_pairs(self, rows: np.ndarray, values: np.ndarray) -> None:
        """Append ``values[i]`` to ring row ``rows[i]`` (global row index; negative = skip) for all i, in order, with
        ONE scatter launch (``nvrx_ring_push_pairs``)."""
        r = np.ascontiguousarray(rows, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.float32)
        if r.size != v.size:
            raise ValueError("rows and values differ in length")
        _native.check(self.lib.nvrx_ring_push_pairs(self.ctx, r.ctypes.data, v.ctypes.data, r.size))

    def push_device(self, row: int, values: torch.Tensor, lr: int = 0) -> None:
        assert values.dtype == torch.float32 and values.is_contiguous() and values.is_cuda
        _native.check(self.lib.nvrx_ring_push_device(self.ctx, lr * self.rows_per_rank + row, values.data_ptr(),
                                                     values.numel(), self.backend.stream_handle))

    def push_device_rows(self, first_row: int, values: torch.Tensor, lr: int = 0) -> None:
        """``values[r]`` appended to row ``first_row + r`` for every r: one call, one strided copy when the rows stand at
        the same ring position."""
        assert values.dtype == torch.float32 and values.is_cuda and values.dim() == 2 and values.stride(1) == 1
        _native.check(self.lib.nvrx_ring_push_device_rows(self.ctx, lr * self.rows_per_rank + first_row, values.shape[0],
                                                          values.data_ptr(), values.shape[1], values.stride(0),
                                                          self.backend.stream_handle))

    def set_count(self, row: int, n: int, lr: int = 0) -> None:
        _native.check(self.lib.nvrx_ring_set_count(self.ctx, lr * self.rows_per_rank + row, n))

    def set_count_all(self, n: int) -> None:
        _native.check(self.lib.nvrx_ring_set_count_all(self.ctx, n))

    def count(self, row: int, lr: int = 0) -> int:
        return _native.check(self.lib.nvrx_ring_count(self.ctx, lr * self.rows_per_rank + row))

    def counts(self) -> np.ndarray:
        """Valid-sample counts of the used rows of logical rank 0 (one C call)."""
        n = self.rows_used
        if self._counts_buf.size < n:
            self._counts_buf = np.zeros(max(n, 64), dtype=np.int32)
        _native.check(self.lib.nvrx_ring_counts(self.ctx, self._counts_buf.ctypes.data, n))
        return self._counts_buf[:n]

    def occupancy_changed(self) -> bool:
        """Whether the SET of used rows that hold samples differs from what the previous call saw (one C call, nothing
        copied): the report's name tables are rebuilt only then."""
        rc = self.lib.nvrx_ring_occupancy_changed(self.ctx, self._rows_used)
        if rc < 0:
            _native.check(rc)
        return rc != 0

    def reset(self) -> None:
        _native.check(self.lib.nvrx_ring_reset(self.ctx))

    def reset_history(self) -> None:
        _native.check(self.lib.nvrx_history_reset(self.ctx, self.backend.stream_handle))

    def flush(self) -> None:
        _native.check(self.lib.nvrx_ring_flush(self.ctx, self.backend.stream_handle))

    def read_row(self, row: int, lr: int = 0) -> np.ndarray:
        stride = _native.check(self.lib.nvrx_ctx_info(self.ctx, 3))
        out = np.empty(stride, dtype=np.float32)
        _native.check(self.lib.nvrx_ring_read(self.ctx, lr * self.rows_per_rank + row, out.ctypes.data, stride,
                                              self.backend.stream_handle))
        return out

    # ---- GPU region timing -----------------------------------------------------------------------
    # (begin / end of a GPU-timed region return False when the region records nothing: its stream is being captured
    #  into a hipGraph -- include/nvrx_straggler.h, NVRX_REGION_SKIPPED)
    def event_begin(self, row: int, stream_handle: int, lr: int = 0) -> bool:
        return _native.check(self.lib.nvrx_event_begin(self.ctx, lr * self.rows_per_rank + row, stream_handle)) == 0

    def event_end(self, row: int, stream_handle: int, lr: int = 0) -> bool:
        return _native.check(self.lib.nvrx_event_end(self.ctx, lr * self.rows_per_rank + row, stream_handle)) == 0

    def harvest(self, wait: bool) -> int:
        return _native.check(self.lib.nvrx_event_harvest(self.ctx, int(wait)))

    # device-side timing: the elapsed time is written into the ring by a kernel on the user's stream
    def stamp_begin(self, row: int, stream_handle: int, lr: int = 0) -> bool:
        rc = self.lib.nvrx_stamp_begin(self.ctx, lr * self.rows_per_rank + row, stream_handle)
        if rc < 0:
            _native.check(rc)
        return rc == 0

    def stamp_end(self, row: int, stream_handle: int, cpu_row: int = -1, cpu_value: float = 0.0, lr: int = 0) -> bool:
        base = lr * self.rows_per_rank
        rc = self.lib.nvrx_stamp_end(self.ctx, base + row, base + cpu_row if cpu_row >= 0 else -1, cpu_value, stream_handle)
        if rc < 0:
            _native.check(rc)
        return rc == 0

    @property
    def regions_skipped(self) -> int:
        """GPU-timed regions that recorded nothing because their stream was being captured into a hipGraph."""
        return int(self.lib.nvrx_ctx_info(self.ctx, 7))

    # ---- report ----------------------------------------------------------------------------------
    def report_local(self, ws: Workspace, names_ok: bool, rows_active: int = 0) -> None:
        """flush -> statistics kernel -> exchange rows, all on the backend's stream (nothing here writes a result block)."""
        if not ws.send_initialised:
            self.backend.send_init(ws)
        rc = self.lib.nvrx_report_local(self.ctx, ws.d_stats, ws.send_ptr, ws.K, ws.S, int(names_ok),
                                        rows_active, self.backend.stream_handle)
        if rc < 0:
            _native.check(rc)

    @staticmethod
    def configure_desc(ws: Workspace, blk: ResultBlock, key: tuple) -> None:
        """Fill a result block's report descriptor for the switches in ``key`` -- ``(rows_active, stats_rows, do_indiv, do_rel,
        thresholds, direct, names_ok, wait, resident)`` -- and remember the key (cold: once per block and shape)."""
        rows_active, stats_rows, do_indiv, do_rel, thresholds, direct, names_ok, wait, resident = key
        d = blk.desc
        d.rows_active, d.stats_rows = rows_active, min(stats_rows, ws.stats_rows)
        d.do_indiv, d.do_rel = int(do_indiv), int(do_rel)
        d.names_ok = int(names_ok)
        for i in range(4):
            d.thresholds[i] = float(thresholds[i])
        if direct is not None:
            d.allgather_fn, d.comm = direct.fn_address, direct.comm_address
        else:
            d.allgather_fn, d.comm = None, None
        d.timeout_s = report_timeout_s()
        d.h_seq_word = blk.h_seq if wait else None
        d.guard_rings = 0 if wait else 1
        # resident score kernel on a stream of its own: the library decides among the eligible shapes (no exchange or
        # peer windows, table fits one workgroup); off when ranks share a device
        d.resident = 1 if (wait and resident) else 0
        blk.desc_key = key

    def report_fused(self, ws: Workspace, rows_active: int, stats_rows: int, do_indiv: bool, do_rel: bool,
                     thresholds: Sequence[float], direct=None, names_ok: bool = True, wait: bool = True,
                     order_after: Optional[int] = None, resident: bool = True, prev_settled: bool = False) -> int:
        """The whole report in one C call (``nvrx_report``): flush -> statistics kernel -> [``ncclAllGather`` of the
        exchange rows through ``direct``] -> score kernel -> wait for the completion word.  On return
        ``ws.scores / flags / meta / stats`` hold this report's values.  ``wait=False`` only enqueues (asynchronous
        report): the caller waits for the returned sequence number with ``backend.wait_seq`` later, and ring writers on
        other streams are ordered after the statistics kernel on the device."""
        blk = ws.flip()  # this report's result block; the previous report's block stays readable for one more report
        d = blk.desc
        key = (rows_active, stats_rows, do_indiv, do_rel, thresholds, direct, names_ok, wait, resident)
        if blk.desc_key != key:  # cold: the switches of this shape changed
            self.configure_desc(ws, blk, key)
        # prev_settled: the caller has SEEN this context's previous asynchronous report complete (ReportGenerator polls its
        # completion word in _settle_inflight and says so only when that poll returned): the library may then stop guarding
        # the rings against that report and, when reports are rare, enqueue this one on the stream it has to follow.  A wait
        # that raised, or a caller that never looked, leaves it 0 and the guards stay.
        d.prev_settled = 1 if (prev_settled and not wait) else 0
        if order_after is not None:  # the caller's current stream: the report follows what is enqueued there
            d.order_after_stream, d.order_after_enabled = order_after, 1
        elif d.order_after_enabled:
            d.order_after_enabled = 0
        if not ws.send_initialised:
            self.backend.send_init(ws)
            self.backend.synchronize()  # cold: a resident score kernel touches the exchange rows from its own stream
        d.seq = ws.seq
        rc = self.lib.nvrx_report(self.ctx, blk.desc_ref, self.backend._stream_handle)
        ws.seq = d.seq
        if rc < 0:
            if rc == _native.ERR_TIMEOUT:
                self.backend.retire_workspace(ws)
            _native.check(rc)
        return ws.seq

    def peek_stats(self) -> np.ndarray:
        """Statistics of every used row right now ([rows_used, 8] on the host); exchanges nothing and
        leaves the history minima alone."""
        total = self.local_ranks * self.rows_per_rank
        # allocated, written and read under the backend's stream: the caching allocator then never hands out a block
        # that kernels on the user's current stream may still be using
        with torch.cuda.stream(self.backend.stream):
            stats = torch.empty((total, _native.STATS_STRIDE), dtype=torch.float32, device=self.backend.device)
            _native.check(self.lib.nvrx_report_local(self.ctx, stats.data_ptr(), None, 0, 0, 1, self.rows_used,
                                                     self.backend.stream_handle))
            host = stats.cpu()
        self.backend.stream.synchronize()
        return host.numpy()

    def timing_enable(self, on: bool) -> None:
        _native.check(self.lib.nvrx_timing_enable(self.ctx, int(on)))

    def timing_read(self, reset: bool = True):
        total = ctypes.c_double()
        launches = ctypes.c_int()
        _native.check(self.lib.nvrx_timing_read(self.ctx, ctypes.byref(total), ctypes.byref(launches), int(reset)))
        return total.value, launches.value
