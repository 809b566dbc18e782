"""The example of examples/straggler_example.py (the reference's examples/straggler/example.py:60-119 on ROCm) run as a
user would run it, and a GPU that REALLY gets slower -- by contention for its compute units, and, where the driver
allows it, by a shader clock held low through ROCm SMI (the counterpart of the reference's `nvidia-smi -lgc 800`,
example.py:20) -- must show in its individual GPU score and be flagged."""
import json
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NVRX_GPU_TIMING")}
    env.update(extra)
    return env


def test_example_two_ranks_report_scores_and_flag_the_slow_rank():
    """Two ranks sharing this box's GPU (gloo), DDP, the forward pass in a GPU-timed section, a report every 60 steps;
    from step 60 on rank 1's stand-in kernel takes 1.5x longer: rank 0 prints relative scores ~1.0 / ~0.67 and flags rank 1."""
    p = subprocess.run([sys.executable, os.path.join(REPO, "examples", "straggler_example.py"), "--num-processes", "2", "--share-gpu",
                        "--steps", "181", "--report-interval", "60", "--batch-size", "512", "--width", "512", "--slow-rank", "1",
                        "--slow-from", "60", "--slow-by", "simulated", "--threshold", "0.8"],
                       capture_output=True, text=True, timeout=240, env=_clean_env(), cwd=REPO)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    out = p.stdout
    print(out[-1500:])
    rel = re.findall(r"step (\d+): GPUs relative perf: (\{.*\})", out)
    assert [int(s) for s, _ in rel] == [60, 120, 180], out[-1500:]
    first, last = eval(rel[0][1]), eval(rel[-1][1])           # {rank: score} literals printed by the example
    assert abs(first[0] - 1.0) < 0.1 and abs(first[1] - 1.0) < 0.1, first      # nobody is slow in the first window
    assert abs(last[0] - 1.0) < 0.05 and 0.55 < last[1] < 0.78, last            # 1 / 1.5, diluted a little by the model's own kernels
    assert re.search(r"step 180: straggler_gpus_relative: \[\(1, ", out), out[-1500:]
    assert "step 60: straggler_gpus_relative" not in out
    assert "gpu telemetry: sclk_mhz=" in out
    assert "time per step [ms]" in out


SLOW_SCRIPT = r'''
import json, os, subprocess, sys, time
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO]
MODE = os.environ["NVRX_TEST_SLOW_MODE"]
import nvrx_straggler
from nvrx_straggler import Detector, gpu_telemetry
import torch

torch.cuda.set_device(0)
Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="node0")
x = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)

def window(steps=50):
    for _ in range(steps):
        with Detector.detection_section("fwd", profile_cuda=True):
            y = x
            for _ in range(4):
                y = x @ y
    torch.cuda.synchronize()
    rep = Detector.generate_report()
    found = rep.identify_stragglers()
    return {"indiv": float(rep.gpu_individual_perf_scores[0]), "flagged": sorted(s.rank for s in found["straggler_gpus_individual"]),
            "telemetry": gpu_telemetry.sample(0), "line": Detector.gpu_telemetry_line()}

out = {"mode": MODE}
window(20)                                   # warm-up window: libraries, the first report's cold steps
out["before"] = window()
if MODE == "contention":
    hog = "import torch,time,sys\nx=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)\nt=time.time()\n" \
          "for i in range(20): y=x@x\ntorch.cuda.synchronize(); print('READY',flush=True)\n" \
          "while time.time()-t<60:\n  for i in range(50): y=x@x\n  torch.cuda.synchronize()\n"
    hogs = [subprocess.Popen([sys.executable, "-c", hog], stdout=subprocess.PIPE, text=True) for _ in range(2)]
    try:
        for h in hogs:
            assert h.stdout.readline().strip() == "READY"
        out["during"] = window()
    finally:
        for h in hogs:
            h.kill()
            h.wait()
else:
    try:
        with gpu_telemetry.slowed_down(0):
            time.sleep(0.5)
            out["during"] = window()
    except gpu_telemetry.SmiRefused as e:
        out["refused"] = str(e)
time.sleep(0.5)
out["after"] = window()
Detector.shutdown()
print("RESULT " + json.dumps(out))
'''


def _slow_run(mode):
    p = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + SLOW_SCRIPT], capture_output=True, text=True, timeout=240,
                       env=_clean_env(NVRX_TEST_SLOW_MODE=mode), cwd=REPO)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])


def test_a_gpu_slowed_by_contention_drops_its_individual_score_and_is_flagged():
    """Two other processes saturate the GPU's compute units for one window: the same four GEMMs per step take well over
    1 / 0.75 times as long, the individual GPU score (this window's GPU time against this GPU's own best, reporting.py:219-253
    with the history of :298-314) falls below the default threshold and the rank is flagged; the window after they are
    gone scores near 1 again."""
    out = _slow_run("contention")
    print({k: (v["indiv"], v["flagged"]) for k, v in out.items() if isinstance(v, dict)}, out["during"]["line"])
    # (the shader clock of an idle box ramps while the first windows run: a window within 20 % of the best one is "as fast")
    assert out["before"]["indiv"] > 0.8 and out["before"]["flagged"] == []
    assert out["during"]["indiv"] < 0.75 and out["during"]["flagged"] == [0], out["during"]
    assert out["after"]["indiv"] > 0.8 and out["after"]["flagged"] == []


def test_a_gpu_with_its_clock_held_low_drops_its_individual_score_and_telemetry_shows_the_clock():
    """``gpu_telemetry.slowed_down`` (rsmi_dev_perf_level_set_v1) for one window; skipped where the driver refuses."""
    out = _slow_run("clock")
    if "refused" in out:
        pytest.skip(f"ROCm SMI would not change the performance level on this box: {out['refused']}")
    print({k: (v["indiv"], v["flagged"], v["telemetry"].get("sclk_mhz")) for k, v in out.items() if isinstance(v, dict)})
    assert out["before"]["indiv"] > 0.8
    assert out["during"]["indiv"] < 0.75 and out["during"]["flagged"] == [0], out["during"]
    assert out["during"]["telemetry"]["sclk_mhz"] < 0.75 * out["during"]["telemetry"]["sclk_peak_mhz"], out["during"]["telemetry"]
    assert out["after"]["indiv"] > 0.8
