"""The direct RCCL exchange (rccl_direct.py) on one MI355X: a 1-rank communicator exercises the ctypes ABI
(128-byte unique id by value, comm handle, ncclAllGather on the detector's own stream).  Multi-rank RCCL
needs one GPU per rank and is covered by the driver's multi-GPU bench; the multi-rank host logic runs on
gloo in test_host_logic.py through the c10d path the direct one falls back to."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_all_gather_on_our_stream():
    from nvrx_straggler import rccl_direct
    from nvrx_straggler.backend import get_backend

    be = get_backend()
    lib = rccl_direct._load_rccl()
    assert lib is not None, "PyTorch-ROCm ships librccl.so"
    uid = rccl_direct._UniqueId()
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    # the id must survive the bytes round trip create() sends through torch.distributed
    blob = ctypes.string_at(ctypes.byref(uid), 128)
    uid2 = rccl_direct._UniqueId()
    ctypes.memmove(ctypes.byref(uid2), blob, 128)
    comm = ctypes.c_void_p()
    torch.cuda.set_device(be.device)
    assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid2, 0) == 0 and comm.value
    ex = rccl_direct.DirectAllGather(lib, comm, 1, 0)
    try:
        L = 129
        rows = np.arange(3 * L, dtype=np.float32).reshape(3, L)
        send = torch.from_numpy(rows).to(be.device)
        table = torch.full((3, L), -7.0, dtype=torch.float32, device=be.device)
        torch.cuda.synchronize()
        ex.all_gather(send.data_ptr(), table.data_ptr(), 3 * L, be.stream_handle)
        be.synchronize()
        assert np.array_equal(table.cpu().numpy(), rows)
        assert ex.comm_ranks() == 1   # ncclCommCount: what bench.py reports as exchange.selection.rccl_comm_ranks
    finally:
        ex.close()


def test_create_is_none_without_a_multi_rank_nccl_group():
    from nvrx_straggler import rccl_direct

    assert rccl_direct.create() is None
