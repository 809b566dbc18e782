"""Acceptance gate: the REFERENCE's own unit tests (read in place from /root/reference, unmodified) run against this
package -- tools/run_reference_tests.sh.  Skipped where the reference tree is absent (the GPU box)."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/tests/straggler/unit"


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference tree not present")
def test_reference_unit_tests_pass_against_this_package():
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    # test_sections / test_wrap_callables time `time.sleep` sections of N(10 ms, 3 ms) against N(15 ms, 3 ms) ones on four
    # gloo processes and threshold the outcome: on a loaded host a sleep overshoot can flip one (seen once in ~20 runs of
    # this suite while other work shared the box, with any implementation behind the API), so one failed attempt is
    # repeated; both tails are shown if it fails twice.
    tails = []
    for _ in range(2):
        r = subprocess.run(["bash", os.path.join(REPO, "tools", "run_reference_tests.sh")], env=env, capture_output=True,
                           text=True, timeout=900)
        tails.append((r.stdout + r.stderr)[-3000:])
        if r.returncode == 0 and "24 passed" in r.stdout:
            return  # relative / individual scores, name mapper, data shared, sections x8, wrap_callables; interval tracker
    raise AssertionError("\n======== second attempt ========\n".join(tails))
