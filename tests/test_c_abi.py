"""The C ABI from a host that is neither Python nor PyTorch: ``tests/c_abi/abi_host.c`` -- plain C99, gcc, only
``include/nvrx_straggler.h`` -- drives rings -> row statistics -> exchange rows -> scores for a small three-rank job and
checks every number against the C oracle and the reference's formulas itself (the boundary of section 2 of the brief:
"plain pointers and sizes, no torch types in the signatures").  ``build()`` compiles it (tests/c_abi/Makefile)."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(REPO, "tests", "c_abi", "_build", "abi_host")


def _need_binary():
    """``build()`` makes it; a tree that was not built here tries once (gcc + the built libraries are all it takes)."""
    if not os.path.exists(HOST):
        p = subprocess.run(["make", "-C", os.path.join(REPO, "tests", "c_abi"), "all"], capture_output=True, text=True)
        if p.returncode != 0 or not os.path.exists(HOST):
            pytest.skip("tests/c_abi/_build/abi_host is not built and could not be built here: " + (p.stderr or p.stdout)[-300:])


def test_the_plain_c_host_links_against_the_library_and_agrees_on_the_descriptor_layout():
    """No device needed: the binary resolves every ABI symbol it uses at load time, ``nvrx_abi_version()`` is the header's
    ``NVRX_ABI_VERSION`` and ``nvrx_report_desc_size()`` is ``sizeof(nvrx_report_desc)`` as a C compiler lays it out."""
    _need_binary()
    p = subprocess.run([HOST, "--link-only"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "abi version 2" in p.stdout


@pytest.mark.gpu
def test_a_plain_c_host_runs_one_report_through_the_abi_and_matches_the_oracle():
    _need_binary()
    p = subprocess.run([HOST], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    assert "ABI HOST OK" in p.stdout


KHOST = os.path.join(REPO, "tests", "c_abi", "_build", "ktrace_host")


def _need_ktrace_host(target="all", path=KHOST):
    """``make`` every time (a no-op when up to date): a sanitizer build left over from an older source would check nothing."""
    p = subprocess.run(["make", "-C", os.path.join(REPO, "tests", "c_abi"), target], capture_output=True, text=True)
    if not os.path.exists(path):
        pytest.skip(f"{path} is not built and could not be built here: " + (p.stderr or p.stdout)[-300:])
    assert p.returncode == 0, (p.stderr or p.stdout)[-1500:]


def test_a_plain_c_host_with_threads_drives_the_kernel_tracer_abi():
    """``tests/c_abi/ktrace_host.c`` (C11 + pthreads, only ``include/nvrx_ktrace.h``): two feeder threads play the
    rocprofiler-sdk callback thread, a sink of two C callbacks plays the engine's rings, the main thread plays the training
    thread (looks, holds, waits).  60 000 records: every key's ring ends with exactly the newest 16 durations in order, the
    counters add up, nothing is lost.  No GPU, no Python in the loop."""
    _need_ktrace_host()
    p = subprocess.run([KHOST], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "KTRACE HOST OK" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


def test_the_kernel_tracer_is_clean_under_thread_sanitizer():
    """The same host and the same library source built with ``-fsanitize=thread`` (``make -C tests/c_abi tsan``): the mutex
    discipline of ``nvrx_ktrace.cpp`` -- feeders, sink, holds, counters, key tables read from another thread -- checked by
    ThreadSanitizer instead of by reading."""
    host = os.path.join(REPO, "tests", "c_abi", "_build", "tsan", "ktrace_host")
    _need_ktrace_host("tsan", host)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    for _ in range(3):
        p = subprocess.run([host], capture_output=True, text=True, timeout=300, env=env)
        if "unexpected memory mapping" in p.stderr:  # (the TSan runtime cannot lay out its shadow under this kernel's ASLR settings)
            pytest.skip("ThreadSanitizer cannot start on this host: " + p.stderr.strip()[-200:])
        assert p.returncode == 0 and "KTRACE HOST OK" in p.stdout and "ThreadSanitizer" not in p.stderr, p.stdout[-800:] + p.stderr[-3000:]
