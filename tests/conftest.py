"""pytest configuration: marker registration and import paths.

* ``gpu`` marks tests that need a real MI355X (run with ``-m gpu`` through gpurun / the driver).
* everything else must pass on a CPU-only box (``-m "not gpu"``).
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(REPO, "nvidia-resiliency-ext_amd")
for p in (REPO, PKG_ROOT, os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def probe_devices_first_on_a_gpu_less_host():
    """CPU boxes only (no /dev/kfd).  Several CPU tests load ``libnvrx_ktrace.so`` -- and with it librocprofiler-sdk -- to feed
    the tracer's data path by hand.  The FIRST device probe of a process that has the SDK loaded (``torch.cuda.is_available()``,
    or the ``torch._C._get_accelerator()`` inside ``dist.barrier()``) then runs the SDK's tool search over every loaded library
    before it finds that there is no device: 25-100 s of system time per process on the build container, which is what made
    the two-rank CPU tests slow and their sleep-timed assertions fragile.  Probed BEFORE the SDK is loaded the answer costs
    nothing and torch remembers it.  Never done where a device exists: there the probe would start HIP ahead of the tracer's
    registration."""
    if os.path.exists("/dev/kfd"):
        return
    import torch

    torch.cuda.is_available()
    torch._C._get_accelerator()


probe_devices_first_on_a_gpu_less_host()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run via gpurun / driver round-end)")
