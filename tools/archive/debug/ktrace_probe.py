import os, sys, time, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
os.environ["NVRX_GPU_TIMING"] = "kernels"
def log(*a):
    print("[probe %.3f]" % time.time(), *a, file=sys.stderr, flush=True)
log("import nvrx")
import nvrx_straggler
from nvrx_straggler import ktrace
lib = ktrace.load()
log("setup err:", ktrace._setup_error, "ready", lib.nvrx_ktrace_ready())
import torch
log("torch imported; cuda init")
torch.cuda.init()
log("ready after init:", lib.nvrx_ktrace_ready())
x = torch.randn(1024, 1024, device="cuda")
(x @ x).sum().item()
log("warm-up done; pending", lib.nvrx_ktrace_pending())
log("start ->", lib.nvrx_ktrace_start())
for i in range(5):
    z = x @ x
    w = torch.relu(x) + 1.0
torch.cuda.synchronize()
log("stop ->", lib.nvrx_ktrace_stop())
log("flush ->", lib.nvrx_ktrace_flush(), "pending", lib.nvrx_ktrace_pending(), "keys", lib.nvrx_ktrace_num_keys())
recs = ktrace.drain_all()
log("drained", recs.size)
for k in sorted(set(recs["key"].tolist())):
    v = recs["us"][recs["key"] == k]
    log(" key", k, ktrace.key_name(k)[:120], "n", v.size, "med us", float(sorted(v)[len(v)//2]))
log("second start/stop cycle")
lib.nvrx_ktrace_start(); (x @ x); torch.cuda.synchronize(); lib.nvrx_ktrace_stop(); lib.nvrx_ktrace_flush()
log("pending", lib.nvrx_ktrace_pending())
log("exiting")
