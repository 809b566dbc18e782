"""bench.py's sidecar (N > 1): a process of its own that holds the finished headline line while the untested exchange routes are
timed, prints it if rank 0's process goes away without saying DONE, and stays silent otherwise.  CPU-only: the sidecar neither
needs a GPU nor torch."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, signal
sys.path.insert(0, REPO)
import bench
out = {"metric": "generate_report_latency_us", "value": 33.0, "us_per_report_median": 30.0, "n_gpus": 2}
p = bench._headline_sidecar(out, "c10d")
assert p is not None
how = sys.argv[1]
if how == "done":
    p.stdin.write(b"DONE\n"); p.stdin.close(); p.wait(timeout=20)
    print("rank0 printed its own line", flush=True)
elif how == "segv":
    os.kill(os.getpid(), signal.SIGSEGV)
elif how == "kill":
    os.kill(os.getpid(), signal.SIGKILL)
else:
    os._exit(3)
"""


def _run(how):
    r = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\n" + SCRIPT, how], capture_output=True, text=True, timeout=120)
    return r, [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_sidecar_is_silent_when_rank_0_says_done():
    r, lines = _run("done")
    assert r.returncode == 0 and not lines and "rank0 printed its own line" in r.stdout, (r.stdout, r.stderr[-800:])


def test_sidecar_prints_the_headline_line_when_rank_0_dies_by_a_signal_or_leaves_without_a_word():
    for how in ("segv", "kill", "exit"):
        r, lines = _run(how)
        assert r.returncode != 0 and len(lines) == 1, (how, r.stdout, r.stderr[-800:])
        d = json.loads(lines[0])
        assert d["value"] == 33.0 and d["routes"]["c10d"]["status"].startswith("ok") and d["routes"]["c10d"]["us_median"] == 30.0
        assert "died" in d["routes"]["rccl"]["status"] and "died" in d["routes"]["peer"]["status"]
