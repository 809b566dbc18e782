"""The DEFAULT exchange route of a real multi-GPU job is ``torch.distributed`` on the job's own NCCL (= RCCL) process group
(``NVRX_EXCHANGE=c10d``).  RCCL refuses two ranks on one device, so on a 1-GPU box that backend can only be brought up with ONE
rank -- where the package short-cuts every collective (world size 1).  This file therefore runs, in a fresh interpreter with
``init_process_group("nccl", world_size=1)``:

* the route's collective primitives exactly as ``dist_utils.all_gather_rows`` / ``ReportIntervalTracker`` /
  ``Detector._agree_timing_mode`` / ``dist_utils`` issue them on an RCCL group -- device tensors of a real workspace, inside the
  backend's ``stream_context()`` (a stream that is not torch's current one), with the statistics queued in front -- and
* a whole ``Detector`` flow on that group (device selection "cuda for NCCL", object collectives of the name exchange, the
  interval tracker's all-reduce on the device).

What it cannot show is two ranks; that is ``bench.py --gpus N`` on a multi-GPU node (S/dist_utils.py:70-76,
tests/straggler/func/ddp_test.py:175-180)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, json
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ['NVRX_TEST_PORT']}", world_size=1, rank=0)
from nvrx_straggler import Detector, Statistic, dist_utils
from nvrx_straggler.backend import get_backend
from nvrx_straggler.interval_tracker import ReportIntervalTracker

out = {"backend": str(dist.get_backend()), "wire": dist_utils.get_device_for_backend(None).type}
be = get_backend()

# (1) the report's collective as all_gather_rows issues it on an RCCL group: a workspace's own send row -> a table, on the
#     backend's stream, behind work queued on that stream (the fill stands for the statistics kernel)
ws = be.workspace(1, 0, 64, 1, 0)
L = ws.send.shape[1]
table = torch.full((1, L), -7.0, dtype=torch.float32, device=be.device)
big = torch.zeros(1 << 24, dtype=torch.float32, device=be.device)
torch.cuda.synchronize()
with be.stream_context():
    big.add_(1.0)                                       # work in front, on the detector's stream
    ws.send.copy_(big[:L].view(1, L) * 0.5 + 2.0)       # = 2.5 everywhere, only if ordered behind the add
    assert torch.cuda.current_stream().cuda_stream == be.stream_handle
    dist.all_gather_into_tensor(table, ws.send.contiguous(), group=None)
be.synchronize()
out["table_ok"] = bool(np.array_equal(table.cpu().numpy(), np.full((1, L), 2.5, np.float32)))

# (2) the tracker's one-off all-reduce on the group's device and the timing-mode agreement (int32 MIN)
tr = ReportIntervalTracker(time_interval=0.5)
out["agreed_interval"] = tr._agreed_interval([0.01] * 16)
t = torch.tensor([3, -3], dtype=torch.int32, device=dist_utils.get_device_for_backend(None))
dist.all_reduce(t, op=dist.ReduceOp.MIN)
out["mode_pair"] = t.tolist()
out["all_true"] = bool(dist_utils.is_all_true(True, None))
out["objects"] = dist_utils.all_gather_object({"names": ["a", "b"]}, None)

# (3) a whole Detector flow on the RCCL group
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0", report_time_interval=0.2)
try:
    x = torch.randn(1024, 1024, device="cuda")
    reports = 0
    rep = None
    for i in range(40):
        with Detector.detection_section("step", profile_cuda=True):
            y = x @ x
        with Detector.detection_section("host", profile_cuda=False):
            pass
        r = Detector.generate_report_if_interval_elapsed()
        if r is not None:
            rep, reports = r, reports + 1
    torch.cuda.synchronize()
    final = Detector.generate_report()
    out["interval_reports"] = reports
    out["iter_interval"] = Detector.report_interval_tracker.iter_interval
    out["sections"] = sorted(final.local_section_summaries)
    out["num_step"] = final.local_section_summaries["step"][Statistic.NUM] if "step" in final.local_section_summaries else None
    out["rel_step"] = final.section_relative_perf_scores["step"][0]
    out["gpu_rel"] = final.gpu_relative_perf_scores[0]
    out["kernel_keys"] = len(final.local_kernel_summaries)
    out["rank_to_node"] = final.rank_to_node
    out["stragglers"] = {k: len(v) for k, v in final.identify_stragglers().items()}
finally:
    Detector.shutdown()
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
"""


def test_rccl_process_group_of_one_rank_carries_the_default_routes_collectives_and_a_detector_flow():
    import json

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NVRX_TEST_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert p.returncode == 0 and lines, (p.stdout[-2000:], p.stderr[-3000:])
    d = json.loads(lines[0][7:])
    assert d["backend"] == "nccl" and d["wire"] == "cuda"
    assert d["table_ok"], "all_gather_into_tensor on the detector's stream did not see the work queued in front of it"
    assert d["agreed_interval"] == 50 and d["mode_pair"] == [3, -3] and d["all_true"] and d["objects"] == [{"names": ["a", "b"]}]
    assert d["sections"] == ["host", "step"] and d["rel_step"] == pytest.approx(1.0, abs=1e-6) and d["gpu_rel"] == pytest.approx(1.0, abs=1e-6)
    assert d["kernel_keys"] >= 1 and d["rank_to_node"] == {"0": "n0"}
    assert d["iter_interval"] is not None and d["iter_interval"] >= 1
    assert all(v == 0 for v in d["stragglers"].values())
