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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run via gpurun / driver round-end)")
