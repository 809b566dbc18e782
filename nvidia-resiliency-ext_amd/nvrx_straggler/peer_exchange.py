"""The report's ONE collective as direct xGMI peer stores (``nvrx_peer_*`` in include/nvrx_straggler.h).

A report exchanges ~0.5 KB per rank: the wire time over xGMI is far below a microsecond, so what an RCCL all-gather
costs here is its own machinery (kernel launch, protocol hand-shakes between eight ranks, proxy progress).  On one
node every GPU can store straight into its peers' memory, so the exchange becomes: one single-workgroup kernel per
rank that writes this rank's row into a window of every peer (HIP IPC mapping, 8-byte ``{epoch, value}`` granules: the
data is its own flag) and polls its own window until all rows of this epoch are there.

``create()`` is the cold, COLLECTIVE set-up (handles travel through ``torch.distributed``); every step is agreed on
by all ranks so that either every rank gets a :class:`PeerAllGather` or none does.  It presents the interface of
``rccl_direct.DirectAllGather`` (``fn_address`` / ``comm_address`` for ``nvrx_report``, ``exchange``, ``close``).
Only ranks of ONE node can share windows; multi-node groups stay on RCCL.  ``choose()`` picks between the two routes:
``NVRX_EXCHANGE=rccl|peer|auto``.  The default is neither (``c10d``: the job's own process group, see ``exchange_mode``):
both in-stream routes are OPT-IN until they have a committed run across real GPUs behind them (so far the windows have only
run between processes sharing one device, and ``ncclAllGather`` on our communicator with one rank).  ``rccl`` = the second
communicator if its checked trial passes on every rank; ``peer`` = the windows if their
checked trial passes on every rank, else RCCL; ``auto`` = both are built, each is timed and checked on a dummy row, the
faster one that delivered the right table on every rank wins (a calm-state timing, so not reproducible run to run).
The chosen route is logged once on rank 0.
"""
from __future__ import annotations

import ctypes
import os
import socket
import time
from typing import Optional

import logging

import numpy as np

import torch
import torch.distributed as dist

from . import _native

_NCCL_FLOAT32 = 7
_LOG = logging.getLogger(__name__)
MAX_FLOATS_PER_RANK = 65536  # floats per rank a window slot holds: local_ranks * (2(K+S)+K+1); 8 MB of windows at 8 ranks


def _all_ok(ok: bool, group) -> bool:
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == dist.Backend.NCCL else torch.device("cpu")
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() > 0)


class PeerAllGather:
    """Window exchange of f32 rows on a caller-supplied ``hipStream_t``."""

    route = "xGMI peer stores (IPC windows)"

    def __init__(self, lib, peer: ctypes.c_void_p, world: int, rank: int, shared_device: bool = False):
        #: some ranks of the group sit on ONE device (tests, folded runs): kernels that poll for each other are then
        #: multiplexed by the hardware scheduler, so the reports keep to one stream per process (no resident scorer)
        self.shared_device = shared_device
        self._lib = lib
        self._peer = peer
        self.world = world
        self.rank = rank
        self.fn_address = lib.nvrx_peer_allgather_address()
        self.comm_address = peer.value
        self.max_count = MAX_FLOATS_PER_RANK
        self._raised_epoch = 0

    def all_gather(self, send_ptr: int, recv_ptr: int, count: int, stream_handle: int) -> None:
        rc = self._lib.nvrx_peer_allgather(send_ptr, recv_ptr, count, _NCCL_FLOAT32, self._peer, stream_handle)
        if rc != 0:
            msg = self._lib.nvrx_last_error()
            raise RuntimeError(f"peer exchange failed: {msg.decode() if msg else rc}")

    def exchange(self, ws, backend):
        self.all_gather(ws.send_ptr, ws.table_ptr, ws.local_ranks * ws.L, backend.stream_handle)
        return ws.table

    def set_timeout(self, seconds: float) -> None:
        """How long the exchange kernel polls for a late peer before it gives up."""
        _native.check(self._lib.nvrx_peer_ready(self._peer, float(seconds)))

    def check(self) -> None:
        """Raise if an exchange kernel gave up waiting for a peer (its table rows are NaN)."""
        epoch = self.timed_out_epoch()
        if epoch and epoch != self._raised_epoch:
            # raised once per timed-out exchange: the error word keeps the epoch of the LAST failure, later reports that
            # complete must not inherit it
            self._raised_epoch = epoch
            raise _native.NativeError(
                f"straggler report exchange: a peer did not publish its row within the timeout (exchange #{epoch}); "
                "the report's scores are invalid")

    def timed_out_epoch(self) -> int:
        """Epoch of the last exchange whose kernel gave up waiting for a peer (0 = never)."""
        e = ctypes.c_uint32(0)
        _native.check(self._lib.nvrx_peer_error(self._peer, ctypes.byref(e)))
        return e.value

    def close(self) -> None:
        if self._peer is not None:
            try:
                self._lib.nvrx_peer_destroy(self._peer)
            finally:
                self._peer = None


def create(group=None, device_index: Optional[int] = None, timeout_s: float = 1800.0) -> Optional[PeerAllGather]:
    """Collective over ``group``: a :class:`PeerAllGather`, or ``None`` on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or not torch.cuda.is_available():
        return None
    world = dist.get_world_size(group)
    if world == 1:
        return None
    rank = dist.get_rank(group)
    lib = _native.load()
    dev = torch.cuda.current_device() if device_index is None else int(device_index)
    peer = ctypes.c_void_p()
    handle = None
    pci = b""
    ok = True
    why = ""
    try:
        _native.check(lib.nvrx_peer_create(dev, world, rank, MAX_FLOATS_PER_RANK, ctypes.byref(peer)))
        buf = ctypes.create_string_buffer(64)
        _native.check(lib.nvrx_peer_ipc_handle(peer, buf))
        handle = buf.raw
        idbuf = ctypes.create_string_buffer(64)
        _native.check(lib.nvrx_peer_device_id(peer, idbuf, 64))
        pci = idbuf.value
    except Exception as e:  # noqa: BLE001  (the decision below must stay collective)
        ok, why = False, f"window set-up failed: {e}"
    # windows can only be shared inside one node
    infos = [None] * world
    try:
        device_id = str(torch.cuda.get_device_properties(dev).uuid)
    except Exception:  # noqa: BLE001
        device_id = f"index{dev}"
    dist.all_gather_object(infos, (socket.gethostname(), handle if ok else None, device_id, pci), group=group)
    same_node = len({i[0] for i in infos}) == 1
    shared_device = len({i[2] for i in infos}) < world
    if ok and not same_node:
        why = "the ranks of the group are on more than one node (windows are shared through HIP IPC)"
    ok = ok and same_node and all(i[1] is not None for i in infos)
    if ok:
        try:
            # preconditions first, for every peer, before anything is mapped: a pair of GPUs without a peer-to-peer path
            # is reported as what it is instead of as a failing (or, worse, hanging) exchange later
            for r, info in enumerate(infos):
                if r != rank and info[3]:
                    _native.check(lib.nvrx_peer_check_access(peer, r, info[3]))
            for r, info in enumerate(infos):
                if r != rank:
                    _native.check(lib.nvrx_peer_connect(peer, r, info[1]))
            _native.check(lib.nvrx_peer_ready(peer, float(timeout_s)))
        except Exception as e:  # noqa: BLE001
            ok, why = False, str(e)
    if why:
        _LOG.warning("straggler report exchange: peer windows not available on rank %d: %s", rank, why)
    if not _all_ok(ok, group):  # one failure sends every rank back to the other route
        if peer.value:
            lib.nvrx_peer_destroy(peer)
        return None
    return PeerAllGather(lib, peer, world, rank, shared_device)


def exchange_mode() -> str:
    """``NVRX_EXCHANGE``: ``c10d`` (default) | ``rccl`` | ``peer`` | ``auto``.

    ``c10d`` is the conservative choice and the default: no communicator of our own, the report's all-gather is a plain
    ``torch.distributed.all_gather_into_tensor`` on the JOB's process group, i.e. on the communicator and in the launch order
    of the job's own collectives (c10d serialises a group's collectives on its one RCCL stream).  It costs a c10d dispatch
    and two event hops per report; what it buys is that nothing of ours can be launched out of order against the job's
    collectives ("two communicators").  The in-stream routes -- ``rccl`` (``ncclAllGather`` on a second
    communicator, on the detector's stream, inside the report's one C call), ``peer`` (xGMI peer stores into IPC windows) and
    ``auto`` (times and checks both, keeps the faster) -- are faster and OPT-IN: none of them has a committed run across
    more than one real GPU yet (this pool's boxes have one), and a default must not be the route nobody has seen work."""
    mode = os.environ.get("NVRX_EXCHANGE", "") or "c10d"
    return mode if mode in ("rccl", "peer", "auto", "c10d") else "c10d"


def trial_timeout_s() -> float:
    """``NVRX_DEBUG_TRIAL_TIMEOUT_S`` (default 5): the longest ONE exchange of a route's trial may take before the route is
    given up as not working on this machine."""
    try:
        v = float(os.environ.get("NVRX_DEBUG_TRIAL_TIMEOUT_S", "5"))
    except ValueError:
        v = 5.0
    return v if v > 0 else 5.0


def _trial(route, group, backend, reps: int = 30):
    """Time ``reps`` exchanges of a recognisable dummy row on ``route`` and check what arrived.  Collective.
    Returns (median microseconds, table correct) for THIS rank.  Never raises and never waits past the trial timeout
    for one exchange: a rank whose exchange fails, delivers a wrong table or does not complete in time reports "not
    correct", which the callers turn into the same decision on every rank (its peers' exchanges run into the
    route's own bounded wait, or into this one)."""
    try:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n = 129
        send = torch.full((n,), float(rank + 1), dtype=torch.float32, device=backend.device)
        recv = torch.zeros((world, n), dtype=torch.float32, device=backend.device)
        torch.cuda.current_stream().synchronize()
        st = backend.stream_handle
        times = []
        check = getattr(route, "timed_out_epoch", None)
        limit = trial_timeout_s()
        done = torch.cuda.Event()
        for i in range(reps + 5):
            t0 = time.perf_counter()
            route.all_gather(send.data_ptr(), recv.data_ptr(), n, st)
            done.record(backend.stream)
            while not done.query():  # (a spin: the exchange is a few microseconds when it works)
                if time.perf_counter() - t0 > limit:
                    # the exchange is stuck (a peer that never joined, a transport that does not move data): the work stays
                    # queued on the detector's stream, the caller drops -- aborts -- the route
                    return float("inf"), False
            if i >= 5:
                times.append(time.perf_counter() - t0)
            if check is not None and check():
                # an unreachable window: every further exchange would sit out the kernel's bounded wait again (the
                # peers run into theirs once and leave the same way), so the trial ends here with "not correct"
                return float("inf"), False
        exp = torch.arange(1, world + 1, dtype=torch.float32).view(world, 1).expand(world, n)
        good = bool(torch.equal(recv.cpu(), exp))
        return float(np.median(times)) * 1e6, good
    except Exception:  # noqa: BLE001
        return float("inf"), False


def _drop(route) -> None:
    """Give a route up after a failed trial: ``abort`` where the route has one (an RCCL communicator whose all-gather may
    still be waiting for a peer on the GPU is torn down with ncclCommAbort, not waited for), else ``close``."""
    try:
        getattr(route, "abort", route.close)()
    except Exception:  # noqa: BLE001
        pass


def choose(group, backend, rccl, peer, timeout_s: float = 1800.0):
    """Pick the exchange route for ``group`` (collective; same answer on every rank).  Returns (route, info)."""
    mode = exchange_mode()
    info = {"mode": mode}
    if peer is not None:
        # the trial must not be able to park a kernel on the GPU for long if a window is unreachable
        peer.set_timeout(float(os.environ.get("NVRX_DEBUG_PEER_TRIAL_TIMEOUT_S", "5")))
    try:
        return _choose(mode, info, group, backend, rccl, peer)
    finally:
        if peer is not None and peer._peer is not None:
            peer.set_timeout(timeout_s)


def _choose(mode, info, group, backend, rccl, peer):
    if mode == "rccl" or peer is None:
        if peer is not None:
            peer.close()
        if rccl is not None:
            # the same checked trial the other modes run: a communicator that does not deliver the right table on every
            # rank is dropped (the reports then stay on torch.distributed), and the figure says what the route costs here
            us, good = _trial(rccl, group, backend)
            info["rccl_us"] = round(us, 2) if np.isfinite(us) else None
            if not _all_ok(good, group):
                _drop(rccl)
                info["rccl_rejected"] = True
                return None, info
        return rccl, info
    if mode == "peer" or rccl is None:
        us, good = _trial(peer, group, backend)
        info["peer_us"] = us
        if _all_ok(good, group):
            if rccl is not None:
                rccl.close()
            return peer, info
        _drop(peer)
        info["peer_rejected"] = True
        if rccl is not None:
            # the fallback gets the same checked trial before any report depends on it
            us, good = _trial(rccl, group, backend)
            info["rccl_us"] = round(us, 2) if np.isfinite(us) else None
            if not _all_ok(good, group):
                _drop(rccl)
                info["rccl_rejected"] = True
                return None, info
        return rccl, info
    # auto: measure both, keep the faster one that was right everywhere
    r_us, r_good = _trial(rccl, group, backend)
    p_us, p_good = _trial(peer, group, backend)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([r_us, p_us, 0.0 if r_good else 1.0, 0.0 if p_good else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    r_us, p_us, r_bad, p_bad = t.tolist()
    info.update({"rccl_us": round(r_us, 2), "peer_us": round(p_us, 2), "rccl_ok": r_bad == 0.0, "peer_ok": p_bad == 0.0})
    use_peer = p_bad == 0.0 and (r_bad != 0.0 or p_us < r_us)
    if use_peer:
        (rccl.close if r_bad == 0.0 else lambda: _drop(rccl))()
        return peer, info
    (peer.close if p_bad == 0.0 else lambda: _drop(peer))()
    if r_bad != 0.0:  # neither route delivered the right table everywhere: the reports stay on torch.distributed
        _drop(rccl)
        info["rccl_rejected"] = info["peer_rejected"] = True
        return None, info
    return rccl, info
