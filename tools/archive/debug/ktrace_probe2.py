import os, sys, time, faulthandler
faulthandler.enable()
faulthandler.dump_traceback_later(80, exit=True)
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
def log(*a):
    print("[probe %.3f]" % time.time(), *a, file=sys.stderr, flush=True)
import numpy as np
import nvrx_straggler
from nvrx_straggler import Detector, Statistic, ktrace
import torch
torch.cuda.set_device(0)
x = torch.randn(1024, 1024, device="cuda")
y = torch.randn(1 << 20, device="cuda")
(x @ x).sum().item()
log("initialize")
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n0")
log("initialized; sections")
for i in range(6):
    with Detector.detection_section("step", profile_cuda=True):
        z = x @ x
        w = torch.relu(y) + 1.0
    torch.sigmoid(y)
log("sections done; sync")
torch.cuda.synchronize()
log("generate_report")
rep = Detector.generate_report()
log("report: kernels", list(rep.local_kernel_summaries.keys())[:3], rep.gpu_relative_perf_scores)
for i in range(3):
    with Detector.detection_section("step", profile_cuda=True):
        z = x @ x
rep = Detector.generate_report()
log("report2:", {k[:30]: int(v[Statistic.NUM]) for k, v in rep.local_kernel_summaries.items()})
Detector.shutdown()
log("exiting")
