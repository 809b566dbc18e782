#!/usr/bin/env python3
"""Multi-process soak: N ranks (gloo group, sharing the one GPU of the box) run randomised Detector cycles together --
tests/workers.py::detector_soak_ranks -- per exchange route and GPU-timing mode.
usage (GPU box): python tools/soak_mp.py [seconds per combination] [world]"""
import faulthandler
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
import mp_util  # noqa: E402
import workers  # noqa: E402

if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    faulthandler.enable()
    rc = 0
    routes = os.environ.get("SOAK_MP_ROUTES", "c10d,peer").split(",")
    modes = os.environ.get("SOAK_MP_MODES", "stamp,kernels").split(",")
    for route in routes:
        for mode in modes:
            env = {"NVRX_EXCHANGE": route, "NVRX_GPU_TIMING": mode, "NVRX_REPORT_TIMEOUT_S": "60", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                   "NVRX_SOAK_TRACE_DIR": os.environ.get("NVRX_SOAK_TRACE_DIR", "")}
            try:
                out = mp_util.run_ranks(workers.detector_soak_ranks, world, timeout=seconds + 150, use_oracle_backend=False, device=None, env=env,
                                        seconds=seconds, seed=int.from_bytes(os.urandom(2), "little"))
                print(f"soak_mp route={route} mode={mode} world={world}: ok {out[0]}", flush=True)
            except BaseException as e:  # noqa: BLE001
                rc = 1
                print(f"soak_mp route={route} mode={mode} world={world}: FAILED {type(e).__name__}: {str(e)[-1500:]}", flush=True)
    sys.exit(rc)
