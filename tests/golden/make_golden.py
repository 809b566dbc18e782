#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's own Python package from /root/reference/src (with a stub
``nvrx_cupti_module`` because libcupti does not exist in a ROCm image, and ``torch.cuda.synchronize``
no-op'd because there is no GPU here) and runs

* ``Detector._get_section_summaries``            -> section_stats.json (+ section_rows.npz inputs)
* ``ReportGenerator.generate_report`` on N gloo ranks -> scoring.json
* the full ``Detector.generate_report`` on 8 gloo ranks x 64 sections x 10 000 samples (BASELINE
  configs #3/#5) -> stress.json   (inputs are re-derived from the recipe in ``synth.py``)
* ten consecutive ``Detector.generate_report`` calls on 8 gloo ranks x 4 sections x 100 samples per report
  (BASELINE config #2: history minima across reports, rank 3 slow from report 5 on) -> loop.json
* the reference's ``StragglerDetectionCallback`` (ptl_resiliency/straggler_det_callback.py) under a scripted trainer and a
  scripted sequence of reports (``callback_script.py``) -> callback.json
* the reference's native ``computeStats``/``CircularBuffer``/``CuptiProfiler`` through
  ``oracle/_ref/libnvrx_ref.so`` -> native.json

Nothing here is read at test time except the emitted files; the GPU box has no /root/reference.
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
REF_SRC = "/root/reference/src"

import synth  # noqa: E402  (tests/golden/synth.py: the shared synthetic-input recipes)


def _install_reference():
    """Make the reference importable: stub native module + no-op cuda sync (no GPU here)."""
    sys.path.insert(0, REF_SRC)
    stub = types.ModuleType("nvrx_cupti_module")

    class CuptiProfiler:  # the reference's pybind surface, cupti_module_py.cpp:44-55
        def __init__(self, bufferSize=0, numBuffers=0, statsMaxLenPerKernel=0):
            pass

        def initialize(self):
            pass

        def shutdown(self):
            pass

        def start(self):
            pass

        def stop(self):
            pass

        def reset(self):
            pass

        def get_stats(self):
            return {}

    stub.CuptiProfiler = CuptiProfiler
    sys.modules["nvrx_cupti_module"] = stub
    import torch

    torch.cuda.synchronize = lambda *a, **k: None
    from nvidia_resiliency_ext.attribution import straggler

    return straggler


def _jsonable(x):
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        if np.isnan(x):
            return "nan"
        if np.isinf(x):
            return "inf" if x > 0 else "-inf"
        return x
    if isinstance(x, (np.integer, int)):
        return int(x)
    return x


def _summ_to_json(summaries):
    return {n: {str(k): _jsonable(v) for k, v in s.items()} for n, s in summaries.items()}


def _report_to_json(report):
    if report is None:
        return None
    return {
        "gpu_relative_perf_scores": _jsonable(dict(report.gpu_relative_perf_scores)),
        "section_relative_perf_scores": _jsonable({k: dict(v) for k, v in report.section_relative_perf_scores.items()}),
        "gpu_individual_perf_scores": _jsonable(dict(report.gpu_individual_perf_scores)),
        "section_individual_perf_scores": _jsonable(
            {k: dict(v) for k, v in report.section_individual_perf_scores.items()}
        ),
        "rank_to_node": _jsonable(dict(report.rank_to_node)),
        "gather_on_rank0": report.gather_on_rank0,
        "rank": report.rank,
    }


def _stragglers_to_json(report, thresholds):
    out = {}
    for thr in thresholds:
        s = report.identify_stragglers(
            gpu_rel_threshold=thr, section_rel_threshold=thr, gpu_indiv_threshold=thr, section_indiv_threshold=thr
        )
        out[str(thr)] = {
            "straggler_gpus_relative": sorted(x.rank for x in s["straggler_gpus_relative"]),
            "straggler_gpus_individual": sorted(x.rank for x in s["straggler_gpus_individual"]),
            "straggler_sections_relative": {
                k: sorted(x.rank for x in v) for k, v in s["straggler_sections_relative"].items()
            },
            "straggler_sections_individual": {
                k: sorted(x.rank for x in v) for k, v in s["straggler_sections_individual"].items()
            },
        }
    return out


# ------------------------------------------------------------------------------------------------
# 1. section statistics through the real Detector._get_section_summaries
# ------------------------------------------------------------------------------------------------
def make_section_stats(straggler):
    from nvidia_resiliency_ext.attribution.straggler.straggler import CustomSection, Detector

    cases = synth.section_stat_cases()
    Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=False)
    out = []
    big = {}
    try:
        for case in cases:
            name, pushed = case["name"], case["values"]
            sec = CustomSection(name=name, location="golden")
            sec.cpu_elapsed_times.extend(float(v) for v in pushed)  # deque(maxlen=8192): newest 8192 kept
            Detector.custom_sections = {name: sec}
            summ = Detector._get_section_summaries()
            rec = {
                "name": name,
                "n_pushed": len(pushed),
                "sha256": hashlib.sha256(np.asarray(pushed, dtype=np.float32).tobytes()).hexdigest(),
                "ring_capacity": CustomSection.max_elapseds_len,
                "expected": _summ_to_json(summ).get(name),
            }
            if len(pushed) <= 16:
                rec["values"] = [float(v) for v in pushed]
            out.append(rec)
    finally:
        Detector.shutdown()
    with open(os.path.join(HERE, "section_stats.json"), "w") as f:
        json.dump({"generator": "reference Detector._get_section_summaries (straggler.py:172-197)", "cases": out}, f, indent=1)
    print("section_stats.json:", len(out), "cases")


# ------------------------------------------------------------------------------------------------
# 2. native module (computeStats / CircularBuffer / CuptiProfiler) through oracle/_ref
# ------------------------------------------------------------------------------------------------
def make_native():
    from oracle import oracle

    R = oracle.ref_lib()
    assert R is not None, "oracle/_ref/libnvrx_ref.so missing: run `make -C oracle ref`"
    out = {"generator": "reference cupti_src/*.cpp via oracle/_ref (fake CUPTI feed)", "compute_stats": [], "ring": [], "profiler": []}
    for case in synth.section_stat_cases():
        vals = np.asarray(case["values"], dtype=np.float32)
        st = oracle.ref_kernel_stats(vals)
        out["compute_stats"].append({"name": case["name"], "n": int(vals.size), "expected": _jsonable(list(st))})
    for n, cap in [(0, 4), (3, 4), (4, 4), (5, 4), (21, 7), (100, 32), (10000, 8192)]:
        vals = np.arange(n, dtype=np.float32) * 0.5 + 1.0
        lin = oracle.ref_ring_run(vals, cap)
        out["ring"].append({"n": n, "capacity": cap, "size": int(lin.size), "first": _jsonable(float(lin[0]) if lin.size else "nan"),
                            "last": _jsonable(float(lin[-1]) if lin.size else "nan"),
                            "sha256": hashlib.sha256(lin.tobytes()).hexdigest()})
    # profiler lifecycle with fake activity records: key format, us conversion, zero-timestamp skip,
    # start/stop gating, ring cap (test_cupti_ext.py:98-116), reset, singleton
    assert R.ref_profiler_create(1 << 20, 8, 7) == 0
    singleton_rc = R.ref_profiler_create(1 << 20, 8, 7)  # second instance must fail
    R.ref_profiler_initialize()
    R.ref_profiler_launch(b"ignored_before_start", 1, 1, 1, 1, 1, 1, 1000, 2000)
    R.ref_profiler_start()
    for i in range(21):
        R.ref_profiler_launch(b"gemm", 256, 1, 1, 1024, 2, 1, 1000, 1000 + 1500 * (i + 1))
    R.ref_profiler_launch(b"gemm", 128, 1, 1, 1024, 2, 1, 5000, 9000)  # different block dims -> new key
    R.ref_profiler_launch(b"zero_ts", 1, 1, 1, 1, 1, 1, 0, 9000)  # skipped (CuptiProfiler.cpp:182-184)
    R.ref_profiler_stop()
    R.ref_profiler_launch(b"ignored_after_stop", 1, 1, 1, 1, 1, 1, 1000, 2000)
    n = R.ref_profiler_get_stats()
    stats = {}
    for i in range(n):
        buf = np.empty(5, dtype=np.float32)
        num = R.ref_profiler_stats(i, buf.ctypes.data)
        stats[R.ref_profiler_key(i).decode()] = _jsonable(list(buf.astype(np.float64)) + [num])
    R.ref_profiler_reset()
    n_after_reset = R.ref_profiler_get_stats()
    R.ref_profiler_shutdown()
    R.ref_profiler_destroy()
    out["profiler"] = {"singleton_second_create_rc": singleton_rc, "stats": stats, "n_after_reset": n_after_reset,
                       "stats_max_len": 7}
    with open(os.path.join(HERE, "native.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("native.json:", len(out["compute_stats"]), "stat cases,", len(stats), "profiler keys")


# ------------------------------------------------------------------------------------------------
# 3. scoring through the real ReportGenerator on gloo ranks
# ------------------------------------------------------------------------------------------------
def _scoring_worker(rank, world_size, store_file, scenario, ret_queue):
    import torch

    _install_reference()
    from nvidia_resiliency_ext.attribution import straggler

    torch.set_num_threads(1)
    if world_size > 1:
        torch.distributed.init_process_group("gloo", init_method=f"file://{store_file}", world_size=world_size, rank=rank)
    S = straggler.Statistic
    key = {"MIN": S.MIN, "MAX": S.MAX, "MED": S.MED, "AVG": S.AVG, "STD": S.STD, "NUM": S.NUM}

    def conv(summ):
        return {n: {key[k]: (float(v) if k != "NUM" else int(v)) for k, v in s.items()} for n, s in summ.items()}

    gen = straggler.reporting.ReportGenerator(
        scenario["scores_to_compute"], gather_on_rank0=scenario["gather_on_rank0"], node_name=f"node{rank}"
    )
    reports = []
    for step in scenario["steps"]:
        sec, ker = step[rank]
        rep = gen.generate_report(conv(sec), conv(ker))
        d = _report_to_json(rep)
        if rep is not None:
            d["stragglers"] = _stragglers_to_json(rep, scenario.get("thresholds", [0.75]))
        reports.append(d)
    ids = {"sections": dict(gen.name_mapper.section_name_to_id), "kernels": dict(gen.name_mapper.kernel_name_to_id)}
    ret_queue.put((rank, reports, ids))
    if world_size > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _run_scenario(scenario):
    import torch.multiprocessing as mp

    W = scenario["world_size"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.NamedTemporaryFile(delete=True) as tmpf:
        store = tmpf.name
    procs = [ctx.Process(target=_scoring_worker, args=(r, W, store, scenario, q)) for r in range(W)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(W):
        r, reps, ids = q.get(timeout=300)
        got[r] = {"reports": reps, "ids": ids}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0, p.exitcode
    return [got[r] for r in range(W)]


def make_scoring():
    scenarios = synth.scoring_scenarios()
    out = []
    for sc in scenarios:
        res = _run_scenario(sc)
        out.append({"scenario": sc, "per_rank": res})
        print("  scenario", sc["name"], "ok")
    with open(os.path.join(HERE, "scoring.json"), "w") as f:
        json.dump({"generator": "reference ReportGenerator.generate_report on gloo ranks (reporting.py:421-554)",
                   "scenarios": _jsonable(out)}, f)
    print("scoring.json:", len(out), "scenarios")


def _fuzz_worker(rank, world_size, store_file, scenarios, ret_queue):
    """All scenarios of one world size in ONE set of processes (a fresh ReportGenerator each)."""
    import torch

    _install_reference()
    from nvidia_resiliency_ext.attribution import straggler

    torch.set_num_threads(1)
    if world_size > 1:
        torch.distributed.init_process_group("gloo", init_method=f"file://{store_file}", world_size=world_size, rank=rank)
    S = straggler.Statistic
    key = {"MIN": S.MIN, "MAX": S.MAX, "MED": S.MED, "AVG": S.AVG, "STD": S.STD, "NUM": S.NUM}

    def conv(summ):
        return {n: {key[k]: (float(v) if k != "NUM" else int(v)) for k, v in s.items()} for n, s in summ.items()}

    results = []
    for scenario in scenarios:
        gen = straggler.reporting.ReportGenerator(
            scenario["scores_to_compute"], gather_on_rank0=scenario["gather_on_rank0"], node_name=f"node{rank}"
        )
        reports = []
        for step in scenario["steps"]:
            sec, ker = step[rank]
            rep = gen.generate_report(conv(sec), conv(ker))
            d = _report_to_json(rep)
            if rep is not None:
                d["stragglers"] = _stragglers_to_json(rep, scenario.get("thresholds", [0.75]))
            reports.append(d)
        ids = {"sections": dict(gen.name_mapper.section_name_to_id), "kernels": dict(gen.name_mapper.kernel_name_to_id)}
        results.append({"reports": reports, "ids": ids})
    ret_queue.put((rank, results))
    if world_size > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def make_fuzz():
    """Random scenarios (synth.fuzz_scenarios) through the real ReportGenerator -> scoring_fuzz.json."""
    import torch.multiprocessing as mp

    scenarios = synth.fuzz_scenarios()
    per_rank = {}
    for W in sorted({sc["world_size"] for sc in scenarios}):
        batch = [sc for sc in scenarios if sc["world_size"] == W]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        with tempfile.NamedTemporaryFile(delete=True) as tmpf:
            store = tmpf.name
        procs = [ctx.Process(target=_fuzz_worker, args=(r, W, store, batch, q)) for r in range(W)]
        for p in procs:
            p.start()
        got = {}
        for _ in range(W):
            r, res = q.get(timeout=600)
            got[r] = res
        for p in procs:
            p.join(60)
            assert p.exitcode == 0, p.exitcode
        for i, sc in enumerate(batch):
            per_rank[sc["name"]] = [got[r][i] for r in range(W)]
        print("  world", W, ":", len(batch), "scenarios ok")
    out = [{"scenario": sc, "per_rank": per_rank[sc["name"]]} for sc in scenarios]
    with open(os.path.join(HERE, "scoring_fuzz.json"), "w") as f:
        json.dump({"generator": "reference ReportGenerator.generate_report on gloo ranks (reporting.py:421-554), random "
                                "scenarios of synth.fuzz_scenarios()", "scenarios": _jsonable(out)}, f)
    print("scoring_fuzz.json:", len(out), "scenarios")


# ------------------------------------------------------------------------------------------------
# 4. full Detector.generate_report on 8 gloo ranks x 64 sections x 10k samples (configs #3 / #5)
# ------------------------------------------------------------------------------------------------
def _stress_worker(rank, world_size, store_file, variants, ret_queue):
    import torch

    _install_reference()
    from nvidia_resiliency_ext.attribution import straggler
    from nvidia_resiliency_ext.attribution.straggler.straggler import CustomSection, Detector

    torch.set_num_threads(1)
    torch.distributed.init_process_group("gloo", init_method=f"file://{store_file}", world_size=world_size, rank=rank)
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name=f"node{rank}")
    results = {}
    for var in variants:
        x = synth.stress_samples(rank, var["S"], var["n"], slow_rank=var["slow_rank"], slow_factor=var["slow_factor"])
        for s in range(var["S"]):
            name = synth.section_name(s)
            if name not in Detector.custom_sections:
                Detector.custom_sections[name] = CustomSection(name=name, location="golden")
            Detector.custom_sections[name].cpu_elapsed_times.extend(x[s].astype(np.float64).tolist())
        rep = Detector.generate_report()
        d = _report_to_json(rep)
        if rep is not None:
            d["stragglers"] = _stragglers_to_json(rep, [0.75, 0.9])
            d["local_section_summaries"] = _summ_to_json(rep.local_section_summaries)
        results[var["name"]] = d
    ret_queue.put((rank, results))
    torch.distributed.barrier()
    Detector.shutdown()
    torch.distributed.destroy_process_group()


def make_stress():
    import torch.multiprocessing as mp

    variants = synth.stress_variants()
    W = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.NamedTemporaryFile(delete=True) as tmpf:
        store = tmpf.name
    procs = [ctx.Process(target=_stress_worker, args=(r, W, store, variants, q)) for r in range(W)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(W):
        r, res = q.get(timeout=900)
        got[r] = res
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # only rank 0 holds reports (gather_on_rank0); keep per-variant input fingerprints too
    fp = {}
    for var in variants:
        h = hashlib.sha256()
        for r in range(W):
            h.update(synth.stress_samples(r, var["S"], var["n"], var["slow_rank"], var["slow_factor"]).tobytes())
        fp[var["name"]] = h.hexdigest()
    with open(os.path.join(HERE, "stress.json"), "w") as f:
        json.dump({"generator": "reference Detector.generate_report, 8 gloo ranks (straggler.py:228-244)",
                   "note": "history (individual scores) carries across variants in list order, as in the reference",
                   "variants": variants, "input_sha256": fp, "rank0": _jsonable(got[0])}, f)
    print("stress.json:", [v["name"] for v in variants])


# ------------------------------------------------------------------------------------------------
# 5. BASELINE config #2: 8 ranks, 4 sections, 1000-step loop, a report every 100 steps (10 consecutive
#    reports through ONE Detector: history minima persist, rings are cleared by every report;
#    reference behaviour pinned by tests/straggler/unit/test_individual_gpu_scores.py:46-121 and
#    reporting.py:298-314).  Rank 3 runs 1.2x slower from report 5 on.
# ------------------------------------------------------------------------------------------------
LOOP = {"world": 8, "S": 4, "n": 100, "reports": 10, "slow_rank": 3, "slow_factor": 1.2, "slow_from": 5}


def _loop_worker(rank, world_size, store_file, ret_queue):
    import torch

    _install_reference()
    from nvidia_resiliency_ext.attribution.straggler.straggler import CustomSection, Detector

    torch.set_num_threads(1)
    torch.distributed.init_process_group("gloo", init_method=f"file://{store_file}", world_size=world_size, rank=rank)
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name=f"node{rank}")
    reports = []
    for t in range(LOOP["reports"]):
        slow = LOOP["slow_rank"] if t >= LOOP["slow_from"] else -1
        x = synth.loop_samples(rank, t, LOOP["S"], LOOP["n"], slow_rank=slow, slow_factor=LOOP["slow_factor"])
        for s in range(LOOP["S"]):
            name = synth.section_name(s)
            if name not in Detector.custom_sections:
                Detector.custom_sections[name] = CustomSection(name=name, location="golden")
            # one append per training step, as detection_section does (straggler.py:343)
            for v in x[s].astype(np.float64).tolist():
                Detector.custom_sections[name].cpu_elapsed_times.append(v)
        rep = Detector.generate_report()
        d = _report_to_json(rep)
        if rep is not None:
            d["stragglers"] = _stragglers_to_json(rep, [0.75, 0.9])
            d["local_section_summaries"] = _summ_to_json(rep.local_section_summaries)
        reports.append(d)
    ret_queue.put((rank, reports))
    torch.distributed.barrier()
    Detector.shutdown()
    torch.distributed.destroy_process_group()


def make_loop():
    import torch.multiprocessing as mp

    W = LOOP["world"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.NamedTemporaryFile(delete=True) as tmpf:
        store = tmpf.name
    procs = [ctx.Process(target=_loop_worker, args=(r, W, store, q)) for r in range(W)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(W):
        r, res = q.get(timeout=900)
        got[r] = res
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(rep is None for r in range(1, W) for rep in got[r])  # gather_on_rank0: only rank 0 reports
    h = hashlib.sha256()
    for t in range(LOOP["reports"]):
        for r in range(W):
            slow = LOOP["slow_rank"] if t >= LOOP["slow_from"] else -1
            h.update(synth.loop_samples(r, t, LOOP["S"], LOOP["n"], slow_rank=slow, slow_factor=LOOP["slow_factor"]).tobytes())
    with open(os.path.join(HERE, "loop.json"), "w") as f:
        json.dump({"generator": "reference Detector.generate_report x10 on 8 gloo ranks (straggler.py:228-244), "
                                "inputs synth.loop_samples (BASELINE config #2)",
                   "config": LOOP, "input_sha256": h.hexdigest(), "rank0_reports": _jsonable(got[0])}, f)
    print("loop.json:", len(got[0]), "reports")


def make_callback():
    """The reference's ``StragglerDetectionCallback`` (ptl_resiliency/straggler_det_callback.py:37-265) driven by the
    scripted trainer of ``callback_script.py``.  Lightning is not in the image: ``lightning.pytorch.callbacks.Callback``
    is a two-line stub (the callback only inherits from it), and the module is loaded from its file so that the package's
    ``__init__`` (which imports the fault-tolerance callbacks and their dependencies) is not executed."""
    import importlib.machinery
    import importlib.util

    straggler = _install_reference()
    for name in ("lightning", "lightning.pytorch", "lightning.pytorch.callbacks"):
        mod = types.ModuleType(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        mod.__path__ = []
        sys.modules[name] = mod
    sys.modules["lightning.pytorch.callbacks"].Callback = type("Callback", (), {})
    pkg_dir = os.path.join(REF_SRC, "nvidia_resiliency_ext", "ptl_resiliency")
    pkg = types.ModuleType("nvidia_resiliency_ext.ptl_resiliency")
    pkg.__path__ = [pkg_dir]
    sys.modules["nvidia_resiliency_ext.ptl_resiliency"] = pkg
    import warnings

    def load(sub):
        full = f"nvidia_resiliency_ext.ptl_resiliency.{sub}"
        spec = importlib.util.spec_from_file_location(full, os.path.join(pkg_dir, sub + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        return mod

    load("_utils")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        ref = load("straggler_det_callback")
    import callback_script

    out = {"generator": "reference StragglerDetectionCallback (ptl_resiliency/straggler_det_callback.py) driven by "
                        "tests/golden/callback_script.py; _gather_flag_from_rank0 replaced by the identity (it needs a CUDA "
                        "device, :231-238); 'processing time' figures and the print order of sets normalised",
           "constructor_error": callback_script.constructor_error(ref.StragglerDetectionCallback),
           "scenarios": [callback_script.drive(ref.StragglerDetectionCallback, ref.straggler, straggler.Report, sc, patch_gather=True)
                         for sc in callback_script.SCENARIOS]}
    with open(os.path.join(HERE, "callback.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("callback.json:", len(out["scenarios"]), "scenarios,",
          sum(len(it["records"]) for sc in out["scenarios"] for it in sc["iterations"]), "log records")


if __name__ == "__main__":
    assert os.path.isdir(REF_SRC), "reference tree not found; golden vectors can only be regenerated in the build container"
    which = sys.argv[1:] or ["section", "native", "scoring", "stress", "loop", "callback", "fuzz"]
    if "native" in which:
        make_native()
    if "section" in which:
        make_section_stats(_install_reference())
    if "scoring" in which:
        make_scoring()
    if "stress" in which:
        make_stress()
    if "loop" in which:
        make_loop()
    if "callback" in which:
        make_callback()
    if "fuzz" in which:
        make_fuzz()
