"""Thin wrappers over ``torch.distributed`` used by the straggler path.

Same function names and behaviour as the reference's ``dist_utils`` (dist_utils.py:19-115): every
helper degrades to a no-op at world size 1 / without an initialised process group, and tensors travel
on ``cuda`` when the group's backend is NCCL (= RCCL on ROCm) and on ``cpu`` otherwise (gloo).

New here: ``all_gather_rows`` -- the ONE collective of a steady-state report.  It replaces the
reference's per-report all_reduce(MIN) flag (C1), all_reduce(MIN) of medians (C4) and gather of
scores (C5) with a single fixed-length all-gather of each rank's exchange row (RCCL over xGMI inside
a node; the payload is ~0.5 KB per rank, so the collective is latency-bound and one is the minimum).
"""
from __future__ import annotations

from typing import Any, List, Optional

import torch
import torch.distributed as dist


_DIST_AVAILABLE = dist.is_available()


def _dist_ready() -> bool:
    return _DIST_AVAILABLE and dist.is_initialized()


def get_world_size(group=None) -> int:
    return dist.get_world_size(group) if _dist_ready() else 1


def get_rank(group=None) -> int:
    return dist.get_rank(group) if _dist_ready() else 0


_C10D_WORLD = getattr(getattr(dist, "distributed_c10d", None), "_world", None) if _DIST_AVAILABLE else None
_WR_CACHE = [None]  # fallback cache of callers that bring none: ONE immutable (default group, group, (world, rank)) tuple


def world_and_rank(group=None, cache=None):
    """(world size, rank) of ``group``: the per-report lookup.  World size and rank of a group never change while
    the default process group lives, so the answer is remembered per (default group object, group): a report pays one
    attribute read instead of three c10d calls (~1 us).  ``cache`` is a one-element list owned by the caller (every
    ``ReportGenerator`` has its own, so two generators on different groups or threads never see each other's entry);
    the entry is ONE immutable tuple, read once and compared by identity, so a concurrent writer can only make a
    reader miss, never hand it another group's answer."""
    if not _DIST_AVAILABLE:
        return 1, 0
    world = _C10D_WORLD
    if world is not None:
        try:
            pg = world.default_pg
        except Exception:  # noqa: BLE001  (a torch without this private handle: the plain calls below)
            pg = False
        if pg is None:
            if cache is not None:
                cache[0] = None  # nothing of a destroyed group is kept alive
            return 1, 0
        if pg is not False:
            if cache is None:
                cache = _WR_CACHE
            entry = cache[0]
            if entry is not None and entry[0] is pg and entry[1] is group:
                return entry[2]
            res = (dist.get_world_size(group), dist.get_rank(group))
            cache[0] = (pg, group, res)
            return res
    if dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def get_device_for_backend(group=None) -> torch.device:
    """``cuda`` for an NCCL/RCCL group, ``cpu`` otherwise (dist_utils.py:68-76)."""
    if _dist_ready() and dist.get_backend(group) == dist.Backend.NCCL:
        return torch.device("cuda")
    return torch.device("cpu")


def all_gather_object(obj: Any, group=None) -> List[Any]:
    """Pickle-based gather of small Python objects; cold path only (names, node names)."""
    world = get_world_size(group)
    if world == 1:
        return [obj]
    out: List[Any] = [None] * world
    dist.all_gather_object(out, obj, group)
    return out


def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
    if get_world_size(group) > 1:
        dist.all_reduce(tensor=tensor, op=op, group=group, async_op=async_op)


def gather_on_rank0(tensor, group=None) -> Optional[List[torch.Tensor]]:
    """Gather equal-shaped tensors on rank 0 (None elsewhere); results return to the input's device."""
    world = get_world_size(group)
    if world == 1:
        return [tensor]
    rank = get_rank(group)
    home = tensor.device
    wire = tensor.to(get_device_for_backend(group))
    bucket = [torch.empty_like(wire) for _ in range(world)] if rank == 0 else None
    dist.gather(tensor=wire, gather_list=bucket, dst=0, group=group)
    if rank != 0:
        return None
    return [t.to(home) for t in bucket]


def is_all_true(flag: bool, group=None) -> bool:
    """True iff ``flag`` is true on every rank (MIN all-reduce of a 0/1 float, dist_utils.py:107-115)."""
    if get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=get_device_for_backend(group))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() > 0)


def all_gather_rows(send: torch.Tensor, table: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather ``send`` ([local_ranks, L] f32) from every rank into ``table`` ([world*local_ranks, L]).

    * world size 1: no collective, ``send`` is returned as the table.
    * NCCL/RCCL group: ``all_gather_into_tensor`` on the current stream, device to device.
    * other backends (gloo): the row makes a host round trip (D2H, gloo all_gather, H2D); the scoring
      still runs on the device.
    """
    world = get_world_size(group)
    if world == 1:
        return send
    wire = get_device_for_backend(group)
    if wire.type == send.device.type:
        if send.device.type == "cuda":
            dist.all_gather_into_tensor(table, send.contiguous(), group=group)
        else:
            dist.all_gather(list(table.view(world, *send.shape).unbind(0)), send.contiguous(), group=group)
        return table
    host_send = send.to(wire)
    parts = [torch.empty_like(host_send) for _ in range(world)]
    dist.all_gather(parts, host_send, group=group)
    table.copy_(torch.cat(parts, dim=0).view_as(table), non_blocking=False)
    return table
