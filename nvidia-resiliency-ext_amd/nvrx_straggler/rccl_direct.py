"""The report's ONE collective, issued straight into RCCL on the detector's own HIP stream.

``torch.distributed.all_gather_into_tensor`` costs a c10d dispatch plus two cross-stream event hops
(current stream -> c10d's internal RCCL stream -> current stream) around a ~0.5 KB all-gather whose
wire time over xGMI is well under a microsecond.  For the steady-state report that overhead is most
of the latency, so the exchange rows are gathered with ``ncclAllGather`` on a communicator of our
own, enqueued on the same stream as the statistics and score kernels: no events, no stream switch,
the three enqueues are back to back.

Only the per-report all-gather goes this way.  Creating the communicator is collective and happens
once, on the first multi-rank report of a process group whose backend is NCCL (= RCCL on ROCm); the
unique id travels through ``torch.distributed`` (cold path), and every step is agreed on by all ranks
(MIN all-reduce of an "ok" flag) so that either every rank uses the direct path or none does -- in
which case the caller stays on ``dist_utils.all_gather_rows``.  Opt-in: ``NVRX_EXCHANGE=rccl`` (``peer_exchange.exchange_mode``).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

_NCCL_UNIQUE_ID_BYTES = 128  # rccl.h:40
_NCCL_FLOAT32 = 7            # rccl.h:466


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * _NCCL_UNIQUE_ID_BYTES)]


def _load_rccl() -> Optional[ctypes.CDLL]:
    """The RCCL that PyTorch-ROCm itself uses (one RCCL per process), else the ROCm one."""
    candidates = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"]
    for path in candidates:
        try:
            lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
        try:
            lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
            lib.ncclGetUniqueId.restype = ctypes.c_int
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
            lib.ncclCommInitRank.restype = ctypes.c_int
            lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
            lib.ncclAllGather.restype = ctypes.c_int
            lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            lib.ncclCommDestroy.restype = ctypes.c_int
            lib.ncclGetErrorString.argtypes = [ctypes.c_int]
            lib.ncclGetErrorString.restype = ctypes.c_char_p
            lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
            lib.ncclCommCount.restype = ctypes.c_int
            lib.ncclCommAbort.argtypes = [ctypes.c_void_p]
            lib.ncclCommAbort.restype = ctypes.c_int
        except AttributeError:
            continue
        return lib
    return None


def _all_ok(ok: bool, group) -> bool:
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() > 0)


class DirectAllGather:
    """``ncclAllGather`` of f32 rows on a caller-supplied ``hipStream_t``."""

    route = "ncclAllGather on the detector's stream"

    def __init__(self, lib: ctypes.CDLL, comm: ctypes.c_void_p, world: int, rank: int):
        self._lib = lib
        self._comm = comm
        self.world = world
        self.rank = rank
        # raw addresses for nvrx_report, which enqueues the all-gather itself between its two kernels
        self.fn_address = ctypes.cast(lib.ncclAllGather, ctypes.c_void_p).value
        self.comm_address = comm.value

    def all_gather(self, send_ptr: int, recv_ptr: int, count: int, stream_handle: int) -> None:
        """Enqueue: every rank's ``count`` floats at ``send_ptr`` -> ``recv_ptr`` ([world, count])."""
        rc = self._lib.ncclAllGather(send_ptr, recv_ptr, count, _NCCL_FLOAT32, self._comm, stream_handle)
        if rc != 0:
            raise RuntimeError(f"ncclAllGather failed: {self._lib.ncclGetErrorString(rc).decode()}")

    def comm_ranks(self) -> int:
        """``ncclCommCount`` of the communicator: how many ranks RCCL itself says take part in the reports' all-gather."""
        n = ctypes.c_int(0)
        rc = self._lib.ncclCommCount(self._comm, ctypes.byref(n))
        return n.value if rc == 0 else -1

    def exchange(self, ws, backend):
        """The report's exchange for workspace ``ws`` on the backend's stream; returns the gathered table."""
        self.all_gather(ws.send_ptr, ws.table_ptr, ws.local_ranks * ws.L, backend.stream_handle)
        return ws.table

    def close(self) -> None:
        if self._comm is not None:
            try:
                self._lib.ncclCommDestroy(self._comm)
            finally:
                self._comm = None

    def abort(self) -> None:
        """Tear the communicator down WITHOUT waiting for what is enqueued on it (``ncclCommAbort``): the way out when a
        trial exchange did not complete -- ``ncclCommDestroy`` would wait for the all-gather that is waiting for a peer."""
        if self._comm is not None:
            try:
                self._lib.ncclCommAbort(self._comm)
            finally:
                self._comm = None


def create(group=None, device_index: Optional[int] = None) -> Optional[DirectAllGather]:
    """Collective over ``group``: a :class:`DirectAllGather`, or ``None`` on every rank.

    Ordering against the job's own collectives: the all-gather runs on a second communicator on the detector's
    stream.  Every rank calls ``generate_report`` at the same point of its program, so the report's all-gather is
    enqueued after the same set of c10d collectives on every rank (no cross-communicator launch-order inversion), and
    ``nvrx_report`` orders the detector's stream after the caller's current stream before it enqueues anything."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(group)
    if world == 1 or dist.get_backend(group) != dist.Backend.NCCL or not torch.cuda.is_available():
        return None
    rank = dist.get_rank(group)
    lib = _load_rccl()
    if not _all_ok(lib is not None, group):  # phase 1: nobody enters the init unless everybody can
        return None
    uid = _UniqueId()
    ok = True
    if rank == 0:
        try:
            ok = lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
        except Exception:  # pragma: no cover - defensive: the decision below must stay collective
            ok = False
    box = [ctypes.string_at(ctypes.byref(uid), _NCCL_UNIQUE_ID_BYTES) if ok else None]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    if box[0] is None:
        return None  # same decision on every rank
    ctypes.memmove(ctypes.byref(uid), box[0], _NCCL_UNIQUE_ID_BYTES)
    comm = ctypes.c_void_p()
    try:
        # the communicator binds to the device that is current during init: make it the backend's for the duration of
        # the call only (the caller's current device is restored)
        with torch.cuda.device(device_index if device_index is not None else torch.cuda.current_device()):
            rc = lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
        ok = rc == 0 and bool(comm.value)
    except Exception:  # pragma: no cover
        ok = False
    if not _all_ok(ok, group):  # phase 2: one failure sends every rank back to the c10d path
        if ok:
            lib.ncclCommDestroy(comm)
        return None
    return DirectAllGather(lib, comm, world, rank)
