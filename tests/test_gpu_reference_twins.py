"""Own-code twins of the reference's GPU-only unit modules, on the HIP engine, once per GPU-timing mode.

The reference's ``tests/straggler/unit/test_det_section_api.py``, ``test_reporting.py`` and ``test_reporting_elapsed.py`` create CUDA
tensors and therefore cannot run in the build container; a Python reference must not travel to the GPU box, so they cannot run
there either (rounds 4-5 staged them).  Each scenario below restates one of those tests -- same workload
shape (10 x Linear(128, 128) on a batch of 16, ``wrap_callables`` on ``forward`` or a ``detection_section``), same
assertions, our own code -- and cites the lines it mirrors.  One fresh interpreter per mode: ``kernels`` has to register its
tracer before the HIP runtime starts.  (``test_cupti_ext.py`` / ``test_cupti_manager.py``: tests/test_gpu_00_ktrace.py.)
"""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import faulthandler, inspect, json, os, sys
faulthandler.enable()
faulthandler.dump_traceback_later(280, exit=True)
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO, os.path.join(REPO, "tests")]
from nvidia_resiliency_ext.attribution import straggler          # the reference's import path; registers the tracer in kernels mode
from unittest.mock import patch
import pytest
import torch
import torch.nn as nn

torch.cuda.set_device(0)
device = torch.device("cuda")
Detector, Statistic = straggler.Detector, straggler.Statistic
ran = []


class Layer(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.layer = nn.Linear(n, n, bias=False)

    def forward(self, x):
        return self.layer(x)


def model10():
    return nn.Sequential(*[Layer(128) for _ in range(10)]).to(device)


# ---- test_det_section_api.py -----------------------------------------------------------------------------------------------
with pytest.raises(RuntimeError):                                   # :41-44 test_fail_if_not_initialized
    with Detector.detection_section("section00"):
        pass
ran.append("fail_if_not_initialized")

Detector.initialize()
try:
    # :48-54 test_unique_name_is_enforced is SKIPPED in the reference and the check it tests is commented out in the shipped code
    # (S/straggler.py:312-316, "TODO: uncomment after Cython issue is resolved"): a name used again from another line is
    # accepted and lands in the same section.  The twin pins that shipped behaviour; the check itself exists here as there
    # (Detector._ensure_section_name_is_valid) and raises when called.
    with Detector.detection_section("section00"):
        pass
    with Detector.detection_section("section00"):
        pass
    assert list(Detector.custom_sections) == ["section00"] and Detector.custom_sections["section00"].total_entry_cnt == 2
    with pytest.raises(ValueError):
        Detector._ensure_section_name_is_valid("section00", "elsewhere.py:1")
finally:
    Detector.shutdown()
ran.append("unique_name_is_enforced")

Detector.initialize()
try:
    with Detector.detection_section():                              # :57-64 test_default_names_are_unique
        pass
    with Detector.detection_section():
        pass
    assert len(Detector.custom_sections) == 2
    a, b = list(Detector.custom_sections.values())
    assert a.name != b.name
finally:
    Detector.shutdown()
ran.append("default_names_are_unique")

Detector.initialize()
try:
    section = Detector.detection_section()                          # :68-79 test_with_block_location_is_used
    _ = 1
    _ = 2
    with section:
        pass
    frameinfo = inspect.getframeinfo(inspect.currentframe())
    assert list(Detector.custom_sections.values())[0].location.endswith(f"{frameinfo.filename}:{frameinfo.lineno - 2}")
finally:
    Detector.shutdown()
ran.append("with_block_location_is_used")

try:                                                                # :82-100 test_periodic_capture
    Detector.initialize(profiling_interval=2)
    a = torch.randn(1000, 1000, device="cuda")
    b = torch.randn(1000, 1000, device="cuda")
    torch.cuda.synchronize()
    for _ in range(4):
        with Detector.detection_section(name="one"):
            _ = torch.matmul(a, b)
    report = Detector.generate_report()
    assert len(report.local_kernel_summaries) == 1, list(report.local_kernel_summaries)     # one kernel (one region row on stamps)
    assert list(report.local_kernel_summaries.values())[0][Statistic.NUM] == 2              # 2 of 4 matmuls were profiled
    assert len(report.local_section_summaries) == 1
    assert report.local_section_summaries["one"][Statistic.NUM] == 2
finally:
    Detector.shutdown()
ran.append("periodic_capture")

try:                                                                # :103-119 test_cuda_profiling_disabled
    Detector.initialize(profiling_interval=1)
    for _ in range(4):
        with Detector.detection_section(name="one", profile_cuda=False):
            _ = torch.matmul(a, b)
    report = Detector.generate_report()
    assert len(report.local_kernel_summaries) == 0
    assert len(report.local_section_summaries) == 1 and report.local_section_summaries["one"][Statistic.NUM] == 4
finally:
    Detector.shutdown()
ran.append("cuda_profiling_disabled")

Detector.initialize()
try:
    with Detector.detection_section(name="one", profile_cuda=True):  # :122-133 test_can_handle_empty_elapseds
        pass
    _ = Detector.generate_report()
    _ = Detector.generate_report()
finally:
    Detector.shutdown()
ran.append("can_handle_empty_elapseds")


# ---- test_reporting.py:87-185 test_reporting_options (one process: world_size 1, rank 0, report_len 1) --------------------
def reporting_options(scores_to_compute, gather_on_rank0, iters=300, report_interval=100):
    model = model10()
    Detector.initialize(scores_to_compute=scores_to_compute, gather_on_rank0=gather_on_rank0)
    try:
        Detector.wrap_callables(callable_ids=[straggler.CallableId(model, "forward")])
        for i in range(iters):
            model(torch.rand(16, 128, device=device))
            if i % report_interval:
                continue
            report = Detector.generate_report()
            stragglers = report.identify_stragglers()
            wanted = ["relative_perf_scores", "individual_perf_scores"] if scores_to_compute == "all" else scores_to_compute
            if "relative_perf_scores" in wanted:
                assert len(report.gpu_relative_perf_scores) == 1
                assert all(len(v) == 1 for v in report.section_relative_perf_scores.values())
            if "individual_perf_scores" in wanted:
                assert len(report.gpu_individual_perf_scores) == 1
                assert all(len(v) == 1 for v in report.section_individual_perf_scores.values())
            for kind in ("straggler_gpus_relative", "straggler_gpus_individual", "straggler_sections_relative", "straggler_sections_individual"):
                assert kind in stragglers
            assert bool(report.gpu_relative_perf_scores) == ("relative_perf_scores" in wanted)
            assert bool(report.section_relative_perf_scores) == ("relative_perf_scores" in wanted)
            assert bool(report.gpu_individual_perf_scores) == ("individual_perf_scores" in wanted)
            assert bool(report.section_individual_perf_scores) == ("individual_perf_scores" in wanted)
    finally:
        Detector.shutdown()


for scores in ("all", ["relative_perf_scores"], ["individual_perf_scores"]):
    for gather in (True, False):        # (the reference skips gather_on_rank0=True on one GPU; with one rank it is rank 0's report)
        reporting_options(scores, gather)
ran.append("reporting_options x6")

with patch("torch.distributed.all_gather_object") as mock_all_gather:   # :188-200 test_no_gather_called
    mock_all_gather.side_effect = RuntimeError("Distributed communication should not be used")
    reporting_options(["individual_perf_scores"], False)
    mock_all_gather.assert_not_called()
ran.append("no_gather_called")


# ---- test_reporting_elapsed.py ---------------------------------------------------------------------------------------------
def report_elapsed(report_time_interval, gather_on_rank0, with_section, iters=400):
    """:71-137 (wrap_callables) and :149-209 (detection_section): the tracker counts every call, has an estimate after its 16
    timed iterations, and a report comes exactly when the interval has elapsed."""
    model = model10()
    Detector.initialize(report_time_interval=report_time_interval, gather_on_rank0=gather_on_rank0)
    reports = 0
    try:
        if not with_section:
            Detector.wrap_callables(callable_ids=[straggler.CallableId(model, "forward")])
        tracker = Detector.report_interval_tracker
        for i in range(iters):
            data = torch.rand(16, 128, device=device)
            if with_section:
                with Detector.detection_section("fwd", profile_cuda=True):
                    model(data)
            else:
                model(data)
            assert i == tracker.current_iter
            if i > tracker.INTERVAL_ESTIMATION_ITERS:
                assert tracker.iter_interval is not None
            report = Detector.generate_report_if_interval_elapsed()
            assert tracker.is_interval_elapsed() == bool(report is not None)
            assert tracker.is_interval_elapsed() == Detector.is_interval_elapsed()
            reports += report is not None
    finally:
        Detector.shutdown()
    return reports


counts = {}
for interval in (5, 0):
    for gather in (True, False):
        counts[f"wrap/{interval}/{gather}"] = report_elapsed(interval, gather, with_section=False)
        counts[f"section/{interval}/{gather}"] = report_elapsed(interval, gather, with_section=True)
assert all(n > 300 for k, n in counts.items() if "/0/" in k), counts      # report_time_interval 0: a report every iteration once estimated
ran.append("report_elapsed x8")

model = model10()                                                   # :212-245 test_report_min_interval_is_profiling_interval
Detector.initialize(profiling_interval=1000, report_time_interval=0.01, gather_on_rank0=True)
try:
    for _ in range(200):
        with Detector.detection_section("fwd", profile_cuda=True):
            model(torch.rand(16, 128, device=device))
        Detector.generate_report_if_interval_elapsed()
    assert Detector.report_interval_tracker.iter_interval >= 1000
finally:
    Detector.shutdown()
ran.append("report_min_interval_is_profiling_interval")

from nvrx_straggler import ktrace
print("RESULT " + json.dumps({"mode": ktrace.timing_mode(), "ran": ran, "reports": counts}))
'''


def _run(mode):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["NVRX_GPU_TIMING"] = mode
    p = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["kernels", "stamp"])
def test_twins_of_the_reference_section_api_reporting_and_elapsed_tests(mode):
    out = _run(mode)
    print(f"[reference twins, {mode}]", out["ran"], out["reports"])
    assert out["mode"] == mode
    assert out["ran"] == ["fail_if_not_initialized", "unique_name_is_enforced", "default_names_are_unique", "with_block_location_is_used",
                          "periodic_capture", "cuda_profiling_disabled", "can_handle_empty_elapseds", "reporting_options x6",
                          "no_gather_called", "report_elapsed x8", "report_min_interval_is_profiling_interval"]
