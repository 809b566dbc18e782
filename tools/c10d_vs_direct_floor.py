"""One MI355X, a 1-rank RCCL process group: what ONE all-gather of the report's exchange row (129 f32) costs the host and the stream
(a) through torch.distributed on the job's process group -- the default route (NVRX_EXCHANGE=c10d) -- and (b) as a bare
ncclAllGather on a communicator of our own from C (the `rccl` route, rccl_direct.py), both on the detector's stream, enqueue and
enqueue + stream wait.  World size 1 moves no data between GPUs: this is the software floor of each route (launch path, c10d's
work objects and stream bookkeeping), the part that does not depend on xGMI.   python tools/c10d_vs_direct_floor.py"""
import ctypes
import json
import os
import socket
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
from nvrx_straggler import rccl_direct
from nvrx_straggler.backend import get_backend

be = get_backend()
L = 129
send = torch.arange(L, dtype=torch.float32, device=be.device).view(1, L)
table = torch.zeros((1, L), dtype=torch.float32, device=be.device)
torch.cuda.synchronize()


def timed(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    be.synchronize()
    enq, tot = [], []
    for _ in range(n):
        t0 = time.perf_counter_ns()
        fn()
        t1 = time.perf_counter_ns()
        be.synchronize()
        t2 = time.perf_counter_ns()
        enq.append((t1 - t0) / 1e3)
        tot.append((t2 - t0) / 1e3)
    return {"enqueue_us_median": round(float(np.median(enq)), 2), "enqueue_plus_wait_us_median": round(float(np.median(tot)), 2),
            "enqueue_us_p95": round(float(np.percentile(enq, 95)), 2)}


def c10d():
    with be.stream_context():
        dist.all_gather_into_tensor(table, send)


pg = dist.distributed_c10d._get_default_group()


def c10d_raw():
    # the ProcessGroup binding itself, without torch.distributed's Python wrapper (argument checks, logging decorator, group lookup)
    with be.stream_context():
        pg._allgather_base(table, send).wait()


lib = rccl_direct._load_rccl()
uid = rccl_direct._UniqueId()
assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
comm = ctypes.c_void_p()
assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
ex = rccl_direct.DirectAllGather(lib, comm, 1, 0)


def direct():
    ex.all_gather(send.data_ptr(), table.data_ptr(), L, be.stream_handle)


def empty_launch():
    with be.stream_context():
        table.add_(0.0)


out = {"what": "software floor of one 516-byte all-gather at world size 1 (no xGMI traffic), MI355X, detector's stream",
       "c10d_torch_distributed": timed(c10d), "c10d_process_group_binding": timed(c10d_raw), "direct_ncclAllGather_from_python_ctypes": timed(direct),
       "one_torch_elementwise_launch_for_scale": timed(empty_launch)}
ex.close()
dist.destroy_process_group()
print(json.dumps(out))
