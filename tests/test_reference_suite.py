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
    r = subprocess.run(["bash", os.path.join(REPO, "tools", "run_reference_tests.sh")], env=env, capture_output=True,
                       text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "24 passed" in r.stdout, tail  # relative / individual scores, name mapper, data shared, sections x8, wrap_callables
