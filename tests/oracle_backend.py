"""CPU checker backend for the host-logic tests -- TEST INFRASTRUCTURE, lives outside the product.

Implements the small backend/rings interface of ``nvrx_straggler.backend`` on top of ``oracle/`` so
that the Python host side (Detector plumbing, name mapping, exchange protocol, report assembly, PTL
callback) can be exercised on a box without a GPU, including world_size-2 gloo runs.  It is injected
with ``nvrx_straggler.backend.set_backend(OracleBackend())``; the product never selects it by itself
and raises when the HIP engine is unavailable.
"""
import contextlib
import time

import numpy as np
import torch

from oracle import oracle

STATS_STRIDE = 8


def _table_len(K, S):
    return 2 * (K + S) + K + 1


class OracleWorkspace:
    def __init__(self, R, K, S, local_ranks, stats_rows):
        self.R, self.K, self.S = R, K, S
        self.local_ranks = local_ranks
        self.stats_rows = stats_rows
        self.L = _table_len(K, S)
        self.W = 2 + 2 * S
        self.send = torch.zeros((local_ranks, self.L), dtype=torch.float32)
        self.table = torch.zeros((R, self.L), dtype=torch.float32) if R != local_ranks else self.send
        self.send_initialised = False
        self.seq = 0
        # same single-block layout as the product workspace: meta | scores | flags | stats
        al = lambda n: (n + 63) // 64 * 64  # noqa: E731
        self._off_meta = 0
        self._off_scores = al(32)
        self._off_flags = self._off_scores + al(R * self.W * 4)
        self._off_stats = self._off_flags + al(R * self.W)
        self.nbytes = self._off_stats + al(max(stats_rows, 1) * STATS_STRIDE * 4)
        self._host = np.zeros(self.nbytes, dtype=np.uint8)
        h = self._host
        self.stats = h[self._off_stats : self._off_stats + stats_rows * 32].view(np.float32).reshape(stats_rows, STATS_STRIDE)
        self.meta = h[0:32].view(np.uint32)
        self.scores = h[self._off_scores : self._off_scores + R * self.W * 4].view(np.float32).reshape(R, self.W)
        self.flags = h[self._off_flags : self._off_flags + R * self.W].reshape(R, self.W)

    # same lazy-collection interface as the product workspace (backend.Workspace.attach / settle)
    _live = None

    def attach(self, live):
        import weakref

        self._live = weakref.ref(live)

    def settle(self):
        ref, self._live = self._live, None
        live = ref() if ref is not None else None
        if live is not None:
            live.detach()

    def host_block(self):
        return self._host.copy()

    def host_head(self):
        return self._host[: self._off_stats].copy()

    def host_stats(self, rows):
        return self.stats[:rows].copy()

    def set_send_row(self, lr, row):
        self.send[lr].copy_(torch.from_numpy(row))
        self.send_initialised = True


class OracleBackend:
    name = "oracle-test"
    device = torch.device("cpu")

    def __init__(self, emulate_fused=False):
        self._ws = {}
        self.score_calls = 0
        #: also offer the product's one-call report (``report_fused``, in-call exchange, deferred wait) so that the
        #: host logic of asynchronous reports can run on CPU / gloo ranks
        self.emulate_fused = emulate_fused

    def create_direct_exchange(self, group):
        import torch.distributed as dist

        if not self.emulate_fused or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        return _OracleDirect(group)

    def wait_seq(self, ws, seq, stats=False):
        assert ws.seq >= seq  # the emulation computes at enqueue time

    def stream_context(self):
        return contextlib.nullcontext()

    @staticmethod
    def current_stream_handle():
        return 0

    def synchronize(self):
        pass

    def workspace(self, R, K, S, local_ranks=1, stats_rows=0):
        key = (R, K, S, local_ranks, stats_rows)
        if key not in self._ws:
            self._ws[key] = OracleWorkspace(R, K, S, local_ranks, stats_rows)
        return self._ws[key]

    def make_rings(self, local_ranks, rows_per_rank, ring_cap):
        cls = OracleRingsFused if self.emulate_fused else OracleRings
        return cls(self, local_ranks, rows_per_rank, ring_cap)

    def send_init(self, ws):
        KS = ws.K + ws.S
        ws.send[:, :KS] = -1.0
        ws.send[:, KS : 2 * KS] = float("nan")
        ws.send[:, 2 * KS :] = 0.0
        ws.send_initialised = True

    def score(self, ws, table, do_indiv, do_rel, thresholds=(0.75,) * 4, wait=True, stats_rows=None):
        ws.settle()
        self.score_calls += 1
        T = table.numpy()
        ws.scores[:] = oracle.score_table(T, ws.K, ws.S, do_indiv, do_rel)
        thr = np.concatenate([[thresholds[2], thresholds[0]], np.full(ws.S, thresholds[3]), np.full(ws.S, thresholds[1])])
        with np.errstate(invalid="ignore"):
            ws.flags[:] = (ws.scores.astype(np.float64) < thr[None, :]).astype(np.uint8)
        ws.meta[:4] = [int((T[:, -1] > 0).all()), ws.R, ws.K, ws.S]


class _OracleDirect:
    """Stand-in for rccl_direct.DirectAllGather: the in-call exchange of ``report_fused`` on a CPU group."""

    def __init__(self, group):
        self.group = group

    def exchange(self, ws, backend):
        from nvrx_straggler import dist_utils

        return dist_utils.all_gather_rows(ws.send, ws.table, self.group)

    def close(self):
        pass


class OracleRings:
    """NumPy rings with the semantics of the device rings (overwrite-oldest, staged event timing)."""

    def __init__(self, backend, local_ranks, rows_per_rank, ring_cap):
        self.backend = backend
        self.local_ranks = local_ranks
        self.rows_per_rank = rows_per_rank
        self.ring_cap = ring_cap
        rows = local_ranks * rows_per_rank
        self.samples = np.zeros((rows, ring_cap), dtype=np.float32)
        self.total = np.zeros(rows, dtype=np.int64)
        self.kinds = np.zeros(rows, dtype=np.uint8)
        self.gid = np.full(rows, -1, dtype=np.int64)
        self.hist_min = np.full(rows, np.inf, dtype=np.float32)
        self.rows_used = 0
        self.section_row_names = {}
        self.kernel_row_names = {}
        self._open = []
        self._pending = []
        self.closed = False

    def close(self):
        self.closed = True

    def alloc_row(self, kind=0):
        if self.rows_used >= self.rows_per_rank:
            raise RuntimeError("straggler rings are full")
        self.rows_used += 1
        self.configure(self.rows_used - 1, kind, -1)
        return self.rows_used - 1

    def row_for(self, kind, name):
        table = self.kernel_row_names if kind == 1 else self.section_row_names
        if name not in table:
            table[name] = self.alloc_row(kind)
        return table[name]

    def ktrace_sink(self):
        """The checker's counterpart of HipRings.ktrace_sink: (ctx, push, row_alloc) as addresses -- two ctypes callbacks
        into these NumPy rings, so the per-kernel tracer's C data path (libnvrx_ktrace.so) can be driven on a box without
        a GPU.  push / row_alloc have the signatures of nvrx_ring_push_staged / nvrx_row_alloc."""
        import ctypes

        if getattr(self, "_sink_cbs", None) is None:
            def push(_ctx, rows, values, n):
                for i in range(n):
                    if rows[i] >= 0:
                        self.push(rows[i], float(values[i]))
                return 0

            def row_alloc(_ctx, kind):
                try:
                    return self.alloc_row(kind)
                except RuntimeError:
                    return -34

            PUSH = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float), ctypes.c_int)
            ALLOC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)
            self._sink_cbs = (PUSH(push), ALLOC(row_alloc))
        cast = ctypes.cast
        return (None, cast(self._sink_cbs[0], ctypes.c_void_p).value, cast(self._sink_cbs[1], ctypes.c_void_p).value)

    def configure(self, row, kind, gid, lr=None):
        for q in (range(self.local_ranks) if lr is None else (lr,)):
            self.kinds[q * self.rows_per_rank + row] = kind
            self.gid[q * self.rows_per_rank + row] = gid

    def push(self, row, value, lr=0):
        r = lr * self.rows_per_rank + row
        self.samples[r, self.total[r] % self.ring_cap] = value
        self.total[r] += 1

    def push_many(self, row, values, lr=0):
        for v in values:
            self.push(row, float(v), lr)

    def push_pairs(self, rows, values):
        for r, v in zip(np.asarray(rows).tolist(), np.asarray(values).tolist()):
            if r >= 0:
                self.push(r, float(v))

    def set_count(self, row, n, lr=0):
        self.total[lr * self.rows_per_rank + row] = n

    def set_count_all(self, n):
        self.total[:] = n

    def counts(self):
        return np.minimum(self.total[: self.rows_used], self.ring_cap).astype(np.int32)

    def count(self, row, lr=0):
        return int(min(self.total[lr * self.rows_per_rank + row], self.ring_cap))

    def occupancy_changed(self):
        now = (self.total[: self.rows_used] > 0).tobytes()
        changed = now != getattr(self, "_occupied_seen", None)
        self._occupied_seen = now
        return changed

    def reset(self):
        self.total[:] = 0

    def reset_history(self):
        self.hist_min[:] = np.inf

    def flush(self):
        pass

    def event_begin(self, row, stream_handle, lr=0):
        self._open.append((row, time.perf_counter_ns()))

    def event_end(self, row, stream_handle, lr=0):
        for i in range(len(self._open) - 1, -1, -1):
            if self._open[i][0] == row:
                _, t0 = self._open.pop(i)
                self._pending.append((row, (time.perf_counter_ns() - t0) * 1e-3))
                return
        raise RuntimeError("event_end without event_begin")

    def harvest(self, wait):
        for row, us in self._pending:
            self.push(row, us)
        self._pending.clear()
        return 0

    def _stats(self):
        counts = np.minimum(self.total, self.ring_cap).astype(np.uint32)
        st = oracle.rows_stats(self.samples, counts, self.kinds)
        out = np.zeros((self.samples.shape[0], STATS_STRIDE), dtype=np.float32)
        out[:, :6] = st
        out[:, 6] = np.where(counts > 0, out[:, 5] * out[:, 3], 0.0)
        return out, counts

    def peek_stats(self):
        return self._stats()[0]

    def report_local(self, ws, names_ok, rows_active=0):
        ws.settle()
        if not ws.send_initialised:
            self.backend.send_init(ws)
        st, counts = self._stats()
        ws.stats[: st.shape[0]] = st[: ws.stats.shape[0]]
        K, KS, L = ws.K, ws.K + ws.S, ws.L
        send = ws.send.numpy()
        active = rows_active or self.rows_per_rank
        for lr in range(self.local_ranks):
            for row in range(active):
                r = lr * self.rows_per_rank + row
                if counts[r] and st[r, 2] < self.hist_min[r]:
                    self.hist_min[r] = st[r, 2]
                g = int(self.gid[r])
                if 0 <= g < KS:
                    send[lr, g] = st[r, 2] if counts[r] else -1.0
                    send[lr, KS + g] = self.hist_min[r] if counts[r] else np.nan
                    if g < K:
                        send[lr, 2 * KS + g] = st[r, 6]
            send[lr, L - 1] = 1.0 if names_ok else 0.0

    def timing_enable(self, on):
        pass

    def timing_read(self, reset=True):
        return 0.0, 0


class OracleRingsFused(OracleRings):
    """Adds the product's one-call report; results are computed at enqueue time, ``wait=False`` just skips nothing."""

    def report_fused(self, ws, rows_active, stats_rows, do_indiv, do_rel, thresholds, direct=None, names_ok=True, wait=True,
                     order_after=None, resident=True, prev_settled=False):
        from nvrx_straggler import dist_utils

        ws.settle()
        self.report_local(ws, names_ok, rows_active=rows_active)
        table = ws.send
        if direct is not None:
            table = dist_utils.all_gather_rows(ws.send, ws.table, direct.group)
        self.backend.score(ws, table, do_indiv, do_rel, thresholds)
        ws.seq = getattr(ws, "seq", 0) + 1
        return ws.seq
