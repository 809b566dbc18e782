"""A short run of tools/soak.py: context create / destroy cycles, matrix loads, synchronous and asynchronous Detector
reports with GPU-timed sections on two streams, reports read late or never, PyTorch allocations in between -- every report
checked, and a GPU memory fault anywhere aborts the process (the long form, 2 x 50 s = 5 100 contexts / 36 000 reports, is in
docs/MEASUREMENTS.md)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "kernels"])
def test_soak_of_the_product_flows_for_a_few_seconds(mode):
    """``kernels``: the same flows with GPU time measured per kernel by the tracer (what a multi-rank job runs) -- the long form of
    this mode found the deadlock of ``nvrx_window_report`` (profiles/r06ae_window_miss_deadlock.txt)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "NVRX_GPU_TIMING")}
    if mode != "default":
        env["NVRX_GPU_TIMING"] = mode
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "soak.py"), "8"], capture_output=True, text=True, timeout=240, env=env)
    assert p.returncode == 0, p.stdout[-1500:] + "\n" + p.stderr[-3000:]
    assert "soak ok" in p.stdout


@pytest.mark.gpu
def test_multi_process_soak_of_detector_cycles_on_both_routes_and_timing_modes():
    """tools/soak_mp.py for a few seconds per combination: two ranks (gloo group, sharing the GPU) run randomised Detector cycles
    together -- asynchronous or not, gathered or not, sections that come and go per rank, names only one rank has -- on the default
    ``c10d`` route and on the peer windows, on region stamps and per kernel.  The long form (2-4 ranks, 20-25 s each) found a
    lone collective and an exchange pairing bug of asynchronous generators (profiles/r06af_soak_mp.txt)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "NVRX_GPU_TIMING", "NVRX_EXCHANGE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "soak_mp.py"), "5", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    assert p.stdout.count(": ok {") == 4, p.stdout[-2000:]
