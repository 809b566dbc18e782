"""The driver's contract with bench.py, on the GPU box: `python bench.py --gpus N ...` WITHOUT a launcher must start its own
ranks and print ONE JSON line from rank 0 (the old bench exited with "launch with torch.distributed.run").
Ranks share the one GPU here over gloo; on a multi-GPU node the same path runs one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--no-cpu-baseline", "--no-overhead", "--no-host-inputs", "--no-extra-legs", "--no-cadence"]


def _run(args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=e,
                       cwd=REPO)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    return r, lines


@pytest.mark.gpu
def test_single_gpu_line_has_the_contract_fields_and_the_round3_legs():
    r, lines = _run(["--gpus", "1", "--steps", "6", "--warmup", "2", "--dump-steps"] + [f for f in FAST if f != "--no-extra-legs"])
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    # a side leg that fails reports its exception as its value instead of taking the line down: none may have
    broken = {k: v["error"] for k, v in d.items() if isinstance(v, dict) and "error" in v}
    assert not broken, broken
    assert d["roofline_n8_shape"]["report_us"] > 0 and d["detector_report"]["us_median"] > 0 and d["section_entry_us"]["profile_cuda_true_us"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is False and d["unit"] == "us"
    assert abs(d["value"] - d["ms_per_step"] * 1e3) < 0.02 and len(d["per_step_us"]) == 6
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]) and d["roofline"]["bound"] == "hbm"
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    # the timed steps are held and read: the read shows up as a leg of its own, and value covers the whole bracketed region
    assert d["report_read"]["identify_stragglers_us"] > 0 and d["us_per_report_fully_read"] > d["value"] and 0 < d["us_per_call_median"] < d["us_per_report_median"]
    tr = d["timed_region"]
    assert tr["region_us"] >= tr["sum_of_steps_us"] and abs(tr["region_us"] / 6 - d["value"]) < 0.5


@pytest.mark.gpu
def test_gpus_2_without_a_launcher_spawns_its_own_ranks():
    r, lines = _run(["--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--cpu-reps", "6", "--route-timeout", "60"]
                    + [f for f in FAST if f != "--no-cpu-baseline"], timeout=500)
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    import torch

    # two ranks, and n_gpus says how many physical devices they really ran on (both on the one GPU of a 1-GPU box)
    devices = min(2, torch.cuda.device_count())
    assert d["ranks"] == 2 and d["n_gpus"] == devices and d["config"]["ranks_share_devices"] == (devices < 2)
    assert d["config"]["logical_ranks_per_gpu"] == 4 and d["config"]["rows_per_gpu"] == 256
    ex = d["exchange"]
    assert ex["bytes_per_rank"] == 4 * 129 * 4 and ex["us_median"] > 0 and ex["ranks"] == 2 and ex["distinct_devices"] == devices
    # which route the reports took and what the checked trial said: a gloo group has no RCCL communicator, so the rows
    # travel through torch.distributed and the selection says so
    assert "route" in ex and isinstance(ex["selection"], dict)
    assert ex["route"].startswith("torch.distributed") or "ncclCommCount" in str(ex["selection"]) or "rccl_comm_ranks" in ex["selection"]
    assert ex["selection"].get("mode") == "c10d"      # the default route: the job's own process group, no second communicator
    assert d["gpu_timing_mode"] in ("stamp", "kernels")
    # one run yields the whole route table: the headline's route plus the same timed loop on each other route, each with its
    # floor and what became of it.  On this box the ranks are gloo processes sharing one GPU: there is no RCCL communicator
    # to build (dropped, the reports ran on torch.distributed) and the peer windows work between processes on one device
    routes = d["routes"]
    assert set(routes) == {"c10d", "rccl", "peer"}, routes
    assert routes["c10d"]["status"].startswith("ok") and routes["c10d"]["us_median"] > 0
    assert routes["rccl"]["status"].startswith("dropped") and routes["rccl"]["flagged_set_right"] and routes["rccl"]["floor_us"] > 0
    assert routes["peer"]["status"] == "ok" and routes["peer"]["flagged_set_right"], routes["peer"]
    assert routes["peer"]["us_median"] > 0 and routes["peer"]["floor_us"] > 0
    # rule (d): roofline and cpu_baseline next to the value at every N
    assert d["roofline"]["kernel"] == "k_row_stats" and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_one_rank_through_the_launcher_path_prints_the_same_line_shape_as_the_plain_run():
    """`--gpus 1` is the single-process path whatever the backend flag says: same keys, same workload, no exchange leg."""
    a, la = _run(["--gpus", "1", "--steps", "4", "--warmup", "1"] + FAST)
    b, lb = _run(["--gpus", "1", "--backend", "gloo", "--steps", "4", "--warmup", "1"] + FAST)
    assert a.returncode == 0 and b.returncode == 0 and len(la) == 1 and len(lb) == 1
    da, db = json.loads(la[0]), json.loads(lb[0])
    assert set(da) == set(db) and da["config"] == db["config"] and "exchange" not in da
    assert da["n_gpus"] == db["n_gpus"] == 1 and da["ranks"] == 1 and da["gpu_timing_mode"] == "stamp"


@pytest.mark.gpu
def test_rccl_needs_one_gpu_per_rank_and_says_so():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node: the RCCL path itself would run")
    r, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"] + FAST, timeout=120)
    assert r.returncode != 0 and not lines
    assert "RCCL" in (r.stderr + r.stdout) and "--backend gloo" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_a_route_leg_that_kills_rank_0_still_leaves_the_headline_line():
    """None of the in-stream routes has run across two real devices: a fault inside one is a signal, not an exception.  The
    finished headline line is held by a sidecar process and printed when rank 0's pipe closes without DONE."""
    r, lines = _run(["--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--route-timeout", "60"] + FAST,
                    env={"NVRX_BENCH_TEST_DIE_IN_ROUTE": "peer"}, timeout=400)
    assert len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode != 0        # (the run did fail: a rank died)
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["ranks"] == 2 and d["roofline"]["kernel"] == "k_row_stats"
    assert d["routes"]["c10d"]["status"].startswith("ok") and "died" in d["routes"]["peer"]["status"] and "died" in d["routes"]["rccl"]["status"]


@pytest.mark.gpu
def test_a_route_leg_that_never_comes_back_costs_its_own_entry_only():
    """The legs run on the main thread under a watchdog: when one does not come back inside --route-timeout, rank 0 prints the
    line it has (the routes measured so far, the stuck one marked) and every rank leaves."""
    r, lines = _run(["--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2", "--route-timeout", "25"] + FAST,
                    env={"NVRX_BENCH_TEST_HANG_IN_ROUTE": "peer"}, timeout=400)
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    routes = d["routes"]
    assert routes["c10d"]["status"].startswith("ok") and routes["peer"]["status"].startswith("timed out")
    assert routes["rccl"]["status"].startswith("dropped") and routes["rccl"]["us_median"] > 0     # (ran before the stuck one)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2])
def test_the_drivers_own_launch_line_through_torch_distributed_run(n):
    """What the driver types for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`), here with N = 1 and with two gloo ranks sharing the GPU: rank 0 prints ONE line, the
    agent exits 0."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "6", "--warmup", "2",
           "--route-timeout", "60"] + FAST + (["--backend", "gloo"] if n > 1 else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=e, cwd=REPO)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-2500:])
    d = json.loads(lines[0])
    assert d["ranks"] == n and d["steps"] == 6 and d["value"] > 0 and d["roofline"]["kernel"] == "k_row_stats"
    if n > 1:
        assert set(d["routes"]) == {"c10d", "rccl", "peer"} and d["exchange"]["selection"].get("mode") == "c10d"


@pytest.mark.gpu
def test_under_the_drivers_launcher_a_rank_0_that_dies_in_a_route_leg_still_leaves_the_headline_line():
    """The same accident as above, under `torch.distributed.run`: the agent tears the other rank down and exits non-zero; the
    sidecar (its own session) prints the finished headline line onto the agent's stdout."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e["NVRX_BENCH_TEST_DIE_IN_ROUTE"] = "rccl"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6", "--warmup", "2",
           "--route-timeout", "60"] + FAST
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=e, cwd=REPO)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode != 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-2500:])
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["ranks"] == 2 and d["routes"]["c10d"]["status"].startswith("ok") and "died" in d["routes"]["rccl"]["status"]
