"""How many bytes does the first HIP call read with and without the kernel-trace tool registered?  (rchar / read_bytes of
this process around torch.cuda.init(): rocprofiler-sdk start-up is slow on boxes with cold storage because every GPU
code object of every library PyTorch links is loaded eagerly when a tool that asked for code-object callbacks is attached.)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO]
mode = sys.argv[1]
if mode != "plain":
    os.environ["NVRX_GPU_TIMING"] = "kernels"
import nvrx_straggler  # noqa: F401  (registers the tool when NVRX_GPU_TIMING=kernels)
import torch
def io():
    d = dict(l.split(": ") for l in open("/proc/self/io").read().strip().splitlines())
    return int(d["rchar"]), int(d["read_bytes"])
a = io(); t0 = time.time()
torch.cuda.init(); x = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
b = io()
print(f"{mode:10s} HIP_ENABLE_DEFERRED_LOADING={os.environ.get('HIP_ENABLE_DEFERRED_LOADING')}: first HIP call {time.time()-t0:6.1f} s, read() bytes {(b[0]-a[0])/1e9:6.2f} GB, from storage {(b[1]-a[1])/1e9:6.2f} GB", flush=True)
