"""Worker functions for the multi-process CPU tests (must be importable by spawned processes)."""
import os
import time
from unittest import mock

import numpy as np


def _summ(d):
    from nvrx_straggler import Statistic as S

    key = {"MIN": S.MIN, "MAX": S.MAX, "MED": S.MED, "AVG": S.AVG, "STD": S.STD, "NUM": S.NUM}
    return {n: {key[k]: v for k, v in s.items()} for n, s in d.items()}


def report_to_plain(rep, thresholds=(0.75,)):
    if rep is None:
        return None
    out = {
        "gpu_relative_perf_scores": dict(rep.gpu_relative_perf_scores),
        "section_relative_perf_scores": {k: dict(v) for k, v in rep.section_relative_perf_scores.items()},
        "gpu_individual_perf_scores": dict(rep.gpu_individual_perf_scores),
        "section_individual_perf_scores": {k: dict(v) for k, v in rep.section_individual_perf_scores.items()},
        "rank_to_node": dict(rep.rank_to_node),
        "gather_on_rank0": rep.gather_on_rank0,
        "rank": rep.rank,
        "stragglers": {},
    }
    for thr in thresholds:
        s = rep.identify_stragglers(thr, thr, thr, thr)
        out["stragglers"][str(thr)] = {
            "straggler_gpus_relative": sorted(x.rank for x in s["straggler_gpus_relative"]),
            "straggler_gpus_individual": sorted(x.rank for x in s["straggler_gpus_individual"]),
            "straggler_sections_relative": {k: sorted(x.rank for x in v) for k, v in s["straggler_sections_relative"].items()},
            "straggler_sections_individual": {k: sorted(x.rank for x in v) for k, v in s["straggler_sections_individual"].items()},
        }
    return out


def scoring_scenario(rank, world, scenario):
    """Replay one golden scenario through OUR ReportGenerator (dict-input path)."""
    from nvrx_straggler.reporting import ReportGenerator

    gen = ReportGenerator(scenario["scores_to_compute"], gather_on_rank0=scenario["gather_on_rank0"], node_name=f"node{rank}")
    reports = []
    for step in scenario["steps"]:
        sec, ker = step[rank]
        rep = gen.generate_report(_summ(sec), _summ(ker))
        reports.append(report_to_plain(rep, scenario.get("thresholds", [0.75])))
    ids = {"sections": dict(gen.name_mapper.section_name_to_id), "kernels": dict(gen.name_mapper.kernel_name_to_id)}
    return {"reports": reports, "ids": ids}


def scoring_scenarios_batch(rank, world, scenarios):
    """Several scenarios of one world size through OUR ReportGenerator in one set of processes (a fresh generator per
    scenario; its exchange route is closed before the next one builds its own)."""
    out = []
    for scenario in scenarios:
        from nvrx_straggler.reporting import ReportGenerator

        gen = ReportGenerator(scenario["scores_to_compute"], gather_on_rank0=scenario["gather_on_rank0"], node_name=f"node{rank}")
        try:
            reports = []
            for step in scenario["steps"]:
                sec, ker = step[rank]
                reports.append(report_to_plain(gen.generate_report(_summ(sec), _summ(ker)), scenario.get("thresholds", [0.75])))
            ids = {"sections": dict(gen.name_mapper.section_name_to_id), "kernels": dict(gen.name_mapper.kernel_name_to_id)}
            out.append({"reports": reports, "ids": ids})
        finally:
            gen.close()
    return out


def gather_object_call_counts(rank, world, n_kernels):
    """all_gather_object is used only when some rank meets a new name
    (reference: tests/straggler/unit/test_data_shared.py:69-100 -> 2, 0, 1, 0, 1)."""
    import torch

    from nvrx_straggler import Statistic as S
    from nvrx_straggler.reporting import ReportGenerator

    def summ(v):
        return {S.MIN: v, S.MAX: v, S.MED: v, S.AVG: v, S.STD: 0.0, S.NUM: 3}

    gen = ReportGenerator(["relative_perf_scores", "individual_perf_scores"], gather_on_rank0=True, node_name=f"n{rank}")
    counts = []
    kernels = {f"kernel_{i}_" + "x" * 64: summ(1.0 + i + rank) for i in range(n_kernels)}
    for round_ in range(5):
        if round_ == 2:
            kernels.update({f"late_{i}": summ(2.0 + rank) for i in range(n_kernels)})
        if round_ == 4 and rank == world - 1:
            kernels["only_last_rank_sees_this"] = summ(3.0)
        with mock.patch("torch.distributed.all_gather_object", wraps=torch.distributed.all_gather_object) as m:
            gen.generate_report({}, kernels)
            counts.append(m.call_count)
    return counts


def name_exchange(rank, world, mode):
    """Cold-path name exchange (name_mapper.sync_names): ids, resolved names and all_gather_object call counts
    for `mode` in {"strings", "auto", "digests"} over four syncs that cover SPMD bulk sets, a rank-private bulk
    set, a mixed small/bulk sync and a single late name."""
    import os

    import torch

    os.environ["NVRX_DEBUG_NAME_EXCHANGE"] = mode
    from nvrx_straggler.name_mapper import NameMapper

    long = "Cijk_Ailk_Bljk_" + "X" * 500
    m = NameMapper()
    calls = []

    def sync(kernels, sections):
        with mock.patch("torch.distributed.all_gather_object", wraps=torch.distributed.all_gather_object) as g:
            m.sync_names(kernels, sections)
            calls.append(g.call_count)

    # 1: SPMD -- every rank holds the same 300 kernel keys (in a rank-dependent order) + rank-specific sections
    common = [f"{long}_{i}_blk_256_1_1_grid_{i}_1_1" for i in range(300)]
    mine = common[rank:] + common[:rank]
    sync(mine, ["fwd", f"only_rank{rank}"])
    # 2: every rank brings 40 private keys on top (nobody else can resolve their digests)
    private = [f"private_r{rank}_{i}" for i in range(40)]
    sync(mine + private, ["fwd", "bwd"])
    # 3: mixed: rank 0 sends 3 strings, the others 50 digests of keys that overlap between the others only
    if rank == 0:
        sync(["late_a", "late_b", "shared_0"], ["fwd"])
    else:
        sync([f"shared_{i}" for i in range(50)], ["fwd"])
    # 4: one late name on the last rank only (reference contract: one call)
    sync(["only_last"] if rank == world - 1 else [], [])
    return {"kernel_ids": dict(m.kernel_name_to_id), "section_ids": dict(m.section_name_to_id),
            "id_to_kernel": dict(m.id_to_kernel_name), "calls": calls, "counter": m.kernel_counter}


def detector_sleep_sections(rank, world, slow_rank, iters):
    """BASELINE config #1: Detector wrapping time.sleep sections on gloo ranks (plumbing, no GPU)."""
    from nvrx_straggler import Detector

    Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=True, node_name=f"host{rank}")
    try:
        for _ in range(iters):
            with Detector.detection_section("section_a", profile_cuda=False):
                time.sleep(0.004)
            with Detector.detection_section("section_b", profile_cuda=False):
                time.sleep(0.008 if rank == slow_rank else 0.004)
        rep = Detector.generate_report()
        out = report_to_plain(rep)
        n_after = len(Detector.custom_sections["section_a"].cpu_elapsed_times)
        names = dict(Detector.reporter.name_mapper.id_to_section_name)
        return {"report": out, "n_after": n_after, "names": names}
    finally:
        Detector.shutdown()


def detector_wrap_callables(rank, world):
    from nvrx_straggler import CallableId, Detector

    class Trainer:
        def training_step(self, x):
            time.sleep(0.002 * (rank + 1))
            return x + 1

    t = Trainer()
    Detector.initialize(scores_to_compute="all", gather_on_rank0=False, profiling_interval=2)
    try:
        Detector.wrap_callables([CallableId(t, "training_step")])
        for i in range(6):
            assert t.training_step(i) == i + 1
        rep = Detector.generate_report()
        Detector.restore_original_callables()
        t.training_step(0)
        rep2 = Detector.generate_report()
        from nvrx_straggler import Statistic

        return {
            "names": list(rep.local_section_summaries.keys()),
            "num": rep.local_section_summaries["Trainer.training_step"][Statistic.NUM],
            "rel": dict(rep.section_relative_perf_scores["Trainer.training_step"]),
            "after_restore": len(rep2.local_section_summaries),
        }
    finally:
        Detector.shutdown()


def folded_job_gloo(rank, world, total_ranks, sections, n):
    """FoldedJob across gloo ranks: the all-gather carries local_ranks rows per process."""
    import synth
    from nvrx_straggler.folded import FoldedJob
    from oracle_backend import OracleBackend  # noqa: F401

    names = [synth.section_name(s) for s in range(sections)]
    job = FoldedJob(total_ranks=total_ranks, section_names=names, ring_cap=n, node_name=f"node{rank}")
    # CPU backend: feed through the host path
    for lr, r in enumerate(job.logical_ranks()):
        x = synth.stress_samples(r, sections, n, slow_rank=3, slow_factor=1.5)
        for s, name in enumerate(names):
            job.rings.push_many(job.rows[name], x[s], lr=lr)
    rep = job.report()
    return report_to_plain(rep, (0.75, 0.9))


def interval_tracker_agreement(rank, world):
    from nvrx_straggler.interval_tracker import ReportIntervalTracker

    tr = ReportIntervalTracker(time_interval=0.2, profiling_interval=1)
    for _ in range(20):
        tr.iter_increase()
        time.sleep(0.002 * (1 + rank))  # ranks run at different speeds; the MAX must win everywhere
    return tr.iter_interval


def detector_name_change_midway(rank, world):
    """Several reports through the cached steady-state plan; then one rank alone meets a new section,
    which must push EVERY rank through the name-syncing path (the flag rides in the exchange row)."""
    import numpy as np

    from nvrx_straggler import Detector

    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name=f"h{rank}")
    out = []
    try:
        def feed(name, value, n=5):
            with Detector.detection_section(name, profile_cuda=False):
                pass
            sec = Detector.custom_sections[name]
            sec.cpu_elapsed_times.clear()
            sec.cpu_elapsed_times.extend(np.full(n, value, dtype=np.float32))

        for t in range(6):
            feed("a", 2.0 * (rank + 1))
            feed("b", 4.0 if t < 4 else 8.0 * (rank + 1))
            if t >= 3 and rank == 1:
                feed("late_rank1_only", 1.0)
            if t == 5:
                feed("late_everywhere", 3.0)
            rep = Detector.generate_report()
            out.append(report_to_plain(rep))
        planned = Detector.reporter._ring_plan is not None
        return {"reports": out, "planned": planned,
                "ids": dict(Detector.reporter.name_mapper.section_name_to_id)}
    finally:
        Detector.shutdown()


def detector_reports_pickle(rank, world):
    """First, second (cached plan) and third report of a Detector run all pickle / deep-copy / json-dump."""
    import copy
    import json
    import pickle as pk

    from nvrx_straggler import Detector

    Detector.initialize(node_name="nodeX")
    out = []
    try:
        for rep_no in range(3):
            for i in range(4):
                for name in ("a", "b"):
                    with Detector.detection_section(name, profile_cuda=False):
                        pass
            report = Detector.generate_report()
            if rank != 0:
                assert report is None
                continue
            back = pk.loads(pk.dumps(report))
            deep = copy.deepcopy(report)
            for r in (report, back, deep):
                assert type(r.section_relative_perf_scores) is dict and set(r.section_relative_perf_scores) == {"a", "b"}
                assert set(r.section_relative_perf_scores["a"]) == set(range(world))
                assert type(r.local_section_summaries) is dict and set(r.local_section_summaries) == {"a", "b"}
                json.dumps(r.gpu_relative_perf_scores), json.dumps(r.section_individual_perf_scores)
                r.identify_stragglers()
            assert back.section_relative_perf_scores == report.section_relative_perf_scores
            assert {str(k): v for k, v in back.local_section_summaries["a"].items()} == \
                   {str(k): v for k, v in report.local_section_summaries["a"].items()}
            out.append(rep_no)
    finally:
        Detector.shutdown()
    return out


def detector_loop_config2(rank, world, device=None):
    """BASELINE config #2 through the real ``Detector``: 4 sections, one sample appended per training step (the way
    ``detection_section`` does), a collective ``generate_report()`` every 100 steps, ten reports; rank 3 runs 1.2x
    slower from report 5 on.  History minima live across reports, every report empties the rings."""
    import synth
    from nvrx_straggler import Detector, Statistic

    cfg = {"S": 4, "n": 100, "reports": 10, "slow_rank": 3, "slow_factor": 1.2, "slow_from": 5}
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name=f"node{rank}")
    out = []
    try:
        names = [synth.section_name(s) for s in range(cfg["S"])]
        for name in names:  # create the sections (one untimed entry each), then start from empty rings
            with Detector.detection_section(name, profile_cuda=False):
                pass
        Detector._reset_sections_elapseds()
        for t in range(cfg["reports"]):
            slow = cfg["slow_rank"] if t >= cfg["slow_from"] else -1
            x = synth.loop_samples(rank, t, cfg["S"], cfg["n"], slow_rank=slow, slow_factor=cfg["slow_factor"])
            for i in range(cfg["n"]):
                for s, name in enumerate(names):
                    Detector.custom_sections[name].cpu_elapsed_times.append(float(x[s, i]))
            rep = Detector.generate_report()
            assert all(len(Detector.custom_sections[n].cpu_elapsed_times) == 0 for n in names)
            d = report_to_plain(rep, (0.75, 0.9))
            if d is not None:
                d["local_section_summaries"] = {n: {str(k): v for k, v in rep.local_section_summaries[n].items()} for n in names}
            out.append(d)
        return out
    finally:
        Detector.shutdown()


def folded_job_device(rank, world, total_ranks, variant):
    """FoldedJob over gloo ranks that SHARE one GPU, real HIP backend: device rings, statistics kernel, host
    round trip of the exchange rows (gloo), score kernel."""
    import synth
    from nvrx_straggler.folded import FoldedJob

    names = [synth.section_name(s) for s in range(variant["S"])]
    job = FoldedJob(total_ranks=total_ranks, section_names=names, ring_cap=8192, node_name=f"node{rank}")
    try:
        out = []
        for rep_no in range(2):  # the second report runs the cached steady-state plan
            for lr, r in enumerate(job.logical_ranks()):
                job.load(lr, synth.stress_samples(r, variant["S"], variant["n"], variant["slow_rank"], variant["slow_factor"]))
            out.append(report_to_plain(job.report(), (0.75, 0.9)))
        return out
    finally:
        job.close()


def detector_async_sequence(rank, world, asynchronous):
    """Six reports through the Detector; at report 3 rank 1 ALONE meets a new section.  ``asynchronous=True``: reports
    come back unread-able until waited for, the new name enters one report later (name sync at the start of report 4,
    where every rank is).  Returns rank 0's reports (plain) and which of them took the cached plan."""
    import numpy as np

    from nvrx_straggler import Detector

    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name=f"h{rank}", asynchronous=asynchronous)
    out, planned = [], []
    try:
        def feed(name, value, n=5):
            with Detector.detection_section(name, profile_cuda=False):
                pass
            sec = Detector.custom_sections[name]
            sec.cpu_elapsed_times.clear()
            sec.cpu_elapsed_times.extend(np.full(n, value, dtype=np.float32))

        held = None
        for t in range(6):
            feed("a", 2.0 * (rank + 1) * (1 + 0.1 * t))
            feed("b", 4.0 + t + rank)
            if t >= 3 and rank == 1:
                feed("late_rank1_only", 1.0 + t)
            plan_before = Detector.reporter._ring_plan
            rep = Detector.generate_report()
            planned.append(plan_before is not None and Detector.reporter._ring_plan is plan_before)
            if held is not None:  # consume report t-1 one step late, the way an asynchronous user would
                out.append(report_to_plain(held))
            held = rep
        out.append(report_to_plain(held) if held is not None else None)
        if rank != 0:
            out = [None] * 6
        return {"reports": out, "planned": planned, "ids": dict(Detector.reporter.name_mapper.section_name_to_id)}
    finally:
        Detector.shutdown()


def detector_async_rows_change_beside_a_new_name(rank, world, asynchronous):
    """Six reports; at report 3 rank 1 ALONE meets a new section while rank 0's set of occupied rows changes in the SAME report
    (its section "b" stays empty): rank 0 leaves its cached plan for the general path exactly when rank 1's row says "ids
    missing".  Returns every rank's reports (plain) -- the exchanges of the ranks must stay paired."""
    import numpy as np

    from nvrx_straggler import Detector

    Detector.initialize(scores_to_compute="all", gather_on_rank0=False, node_name=f"h{rank}", asynchronous=asynchronous)
    try:
        def feed(name, value, n=5):
            with Detector.detection_section(name, profile_cuda=False):
                pass
            sec = Detector.custom_sections[name]
            sec.cpu_elapsed_times.clear()
            sec.cpu_elapsed_times.extend(np.full(n, value, dtype=np.float32))

        out = []
        for t in range(7):
            feed("a", 2.0 * (rank + 1) * (1 + 0.1 * t))
            if not (rank == 0 and t == 3):
                feed("b", 4.0 + t + rank)
            if t >= 3 and rank == 1:
                feed("late_rank1_only", 1.0 + t)
            out.append(report_to_plain(Detector.generate_report()))
        return out
    finally:
        Detector.shutdown()


def detector_async_first_window_empty_on_one_rank(rank, world, asynchronous):
    """Rank 1 has recorded nothing when the job's FIRST report comes (its sections start one report later): every other rank's
    names are new, nobody has a plan -- the name sync of that report happens inside it, on every rank."""
    import numpy as np

    from nvrx_straggler import Detector

    Detector.initialize(scores_to_compute="all", gather_on_rank0=False, node_name=f"h{rank}", asynchronous=asynchronous)
    try:
        def feed(name, value, n=5):
            with Detector.detection_section(name, profile_cuda=False):
                pass
            sec = Detector.custom_sections[name]
            sec.cpu_elapsed_times.clear()
            sec.cpu_elapsed_times.extend(np.full(n, value, dtype=np.float32))

        out = []
        for t in range(5):
            if not (rank == 1 and t == 0):
                feed("a", 2.0 * (rank + 1))
                feed("b", 4.0 + rank)
            out.append(report_to_plain(Detector.generate_report()))
        return out
    finally:
        Detector.shutdown()


def detector_async_individual_only(rank, world):
    """An asynchronous generator that scores this rank ALONE (individual scores, nothing gathered: no collective in any
    report); at report 3 rank 1 meets a new section.  Returns, per report, the sections that have an individual score."""
    import numpy as np

    from nvrx_straggler import Detector

    Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=False, node_name=f"h{rank}", asynchronous=True)
    try:
        def feed(name, value, n=5):
            with Detector.detection_section(name, profile_cuda=False):
                pass
            sec = Detector.custom_sections[name]
            sec.cpu_elapsed_times.clear()
            sec.cpu_elapsed_times.extend(np.full(n, value, dtype=np.float32))

        seen = []
        for t in range(6):
            feed("a", 2.0 * (rank + 1))
            feed("b", 4.0 + rank)
            if t >= 3 and rank == 1:
                feed("late_rank1_only", 1.0 + t)
            rep = Detector.generate_report()
            seen.append(sorted(rep.section_individual_perf_scores))
        return seen
    finally:
        Detector.shutdown()


def peer_exchange_stress(rank, world, iters, count):
    """The peer-window exchange on its own: `world` processes (sharing one GPU in the tests), `iters` exchanges of a
    `count`-float row whose content changes every time; every rank checks every gathered table."""
    import torch
    import torch.distributed as dist

    from nvrx_straggler import peer_exchange
    from nvrx_straggler.backend import get_backend

    be = get_backend()
    pg = peer_exchange.create(None, be.device.index, timeout_s=8.0)
    assert pg is not None
    try:
        send = torch.zeros(count, dtype=torch.float32, device=be.device)
        recv = torch.zeros((world, count), dtype=torch.float32, device=be.device)
        base = torch.arange(count, dtype=torch.float32)
        torch.cuda.synchronize()
        bad = 0
        with torch.cuda.stream(be.stream):
            for it in range(iters):
                send.copy_((base + 1000.0 * rank + it).to(be.device), non_blocking=False)
                pg.all_gather(send.data_ptr(), recv.data_ptr(), count, be.stream_handle)
                got = recv.cpu()
                exp = torch.stack([base + 1000.0 * r + it for r in range(world)])
                bad += int(not torch.equal(got, exp))
                if it % 7 == rank % 7:
                    time.sleep(0.003)  # uneven arrival: the other ranks' kernels poll while this one is late
        assert pg.timed_out_epoch() == 0
        return bad
    finally:
        dist.barrier()
        pg.close()


def folded_loop_config2(rank, world, asynchronous):
    """Config #2 through FoldedJob on `world` processes (8 / world logical ranks each): ten reports, history minima
    across reports.  Returns the plain reports plus which exchange route / report path served them."""
    import synth
    from nvrx_straggler.folded import FoldedJob

    cfg = {"S": 4, "n": 100, "reports": 10, "slow_rank": 3, "slow_factor": 1.2, "slow_from": 5}
    names = [synth.section_name(s) for s in range(cfg["S"])]
    job = FoldedJob(total_ranks=8, section_names=names, ring_cap=8192, node_name=f"node{rank}")
    job.reporter.asynchronous = asynchronous
    try:
        held, out = [], []
        for t in range(cfg["reports"]):
            slow = cfg["slow_rank"] if t >= cfg["slow_from"] else -1
            for lr, r in enumerate(job.logical_ranks()):
                job.load(lr, synth.loop_samples(r, t, cfg["S"], cfg["n"], slow_rank=slow, slow_factor=cfg["slow_factor"]))
            held.append(job.report())
            if len(held) > 1:  # read one report late, as an asynchronous consumer would
                out.append(report_to_plain(held[-2], (0.75, 0.9)))
        out.append(report_to_plain(held[-1], (0.75, 0.9)))
        plan = job.reporter._ring_plan
        return {"reports": out, "route": getattr(job.reporter._direct, "route", "none"),
                "fused": bool(plan is not None and plan.fused)}
    finally:
        job.close()


def folded_loop_late_process(rank, world, late_report, late_by_s, peer_wait_s):
    """Config #2 as in folded_loop_config2, but the LAST process reaches report `late_report` `late_by_s` seconds after
    the others, whose exchange kernels give up on it after `peer_wait_s`: their report raises (once), the late process'
    own report completes (its peers' rows were published long ago), and every later report is the golden one again."""
    import torch.distributed as dist

    import synth
    from nvrx_straggler import _native
    from nvrx_straggler.folded import FoldedJob

    cfg = {"S": 4, "n": 100, "reports": 10, "slow_rank": 3, "slow_factor": 1.2, "slow_from": 5}
    names = [synth.section_name(s) for s in range(cfg["S"])]
    job = FoldedJob(total_ranks=8, section_names=names, ring_cap=8192, node_name=f"node{rank}")
    try:
        out, raised = [], []
        for t in range(cfg["reports"]):
            slow = cfg["slow_rank"] if t >= cfg["slow_from"] else -1
            for lr, r in enumerate(job.logical_ranks()):
                job.load(lr, synth.loop_samples(r, t, cfg["S"], cfg["n"], slow_rank=slow, slow_factor=cfg["slow_factor"]))
            dist.barrier()
            if t == late_report:
                job.reporter._direct.set_timeout(peer_wait_s)
                if rank == world - 1:
                    time.sleep(late_by_s)
            try:
                out.append(report_to_plain(job.report(), (0.75, 0.9)))
            except _native.NativeError as e:
                raised.append((t, str(e)))
                out.append("raised")
                job.rings.reset()  # what job.report() would have done after the report
            if t == late_report:
                dist.barrier()  # the late process has finished the report the others gave up on
                job.reporter._direct.set_timeout(20.0)
        route = job.reporter._direct
        return {"reports": out, "raised": raised, "route": getattr(route, "route", "none"),
                "timed_out_epoch": route.timed_out_epoch()}
    finally:
        job.close()


def ptl_callback_run(rank, world, slow_rank):
    """StragglerDetectionCallback driven by a duck-typed trainer (Lightning is not in the image): training_step does
    real GPU work when a GPU backend is active, the slow rank does 8x of it; returns what the callback logged/decided."""
    import logging

    import torch

    from nvidia_resiliency_ext.ptl_resiliency import StragglerDetectionCallback
    from nvrx_straggler import Detector
    from nvrx_straggler.backend import get_backend

    on_gpu = get_backend().name == "hip"
    x = torch.randn(1024, 1024, device="cuda") if on_gpu else None

    class Strategy:
        def training_step(self, batch):
            reps = 8 if rank == slow_rank else 1   # far from the 0.7 threshold even when the ranks share one GPU
            if on_gpu:
                y = x
                for _ in range(4 * reps):
                    y = y @ x
                torch.cuda.current_stream().synchronize()
            else:
                time.sleep(0.002 * reps)
            return batch

    class Trainer:
        def __init__(self):
            self.strategy = Strategy()
            self.global_rank = rank
            self.should_stop = False
            self.checkpoint_callback = None

    class Module:
        def __init__(self):
            self.logged = []

        def log_dict(self, d, **kw):
            self.logged.append(dict(d))

    records = []

    class Grab(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())

    log = logging.getLogger("test.straggler.ptl")
    log.setLevel(logging.INFO)
    log.addHandler(Grab())
    cb = StragglerDetectionCallback(report_time_interval=0.05, calc_relative_gpu_perf=True, calc_individual_gpu_perf=True,
                                    num_gpu_perf_scores_to_print=2, gpu_relative_perf_threshold=0.7,
                                    gpu_individual_perf_threshold=0.7, stop_if_detected=True, enable_ptl_logging=True,
                                    logger_name="test.straggler.ptl")
    trainer, module = Trainer(), Module()
    cb.setup(trainer, module, "fit")
    try:
        # the report interval comes from the measured step time (0.05 s / median step): run until two reports have
        # been produced on rank 0's clock; the iteration count is agreed through the report collectives themselves
        n_iters = 80
        tr = Detector.report_interval_tracker
        i = 0
        while i < n_iters:
            trainer.strategy.training_step(i)
            cb.on_train_batch_end(trainer, module, None, None, i)
            i += 1
            if i == 20 and tr.iter_interval is not None:
                n_iters = max(n_iters, 2 * tr.iter_interval + 1)  # iter_interval is MAX-reduced: same on every rank
        return {"interval": Detector.report_interval_tracker.iter_interval, "should_stop": trainer.should_stop,
                "logged": ({k: v for d in module.logged[-2:] for k, v in d.items()} if module.logged else None),
                "messages": records,
                "sections": sorted(Detector.custom_sections)}
    finally:
        cb.teardown(trainer, module, "fit")


def route_fault_injection(rank, world, fault, scenario):
    """The in-stream exchange route of the reports replaced by one that misbehaves on the LAST rank only (``wrong_table``,
    ``never_completes``) or on every rank (``error``), behind the generic ``allgather_fn`` hook of the report descriptor
    (include/nvrx_straggler.h): a real working exchange (the peer windows of ranks sharing this GPU) wrapped in a ctypes
    callback that corrupts the gathered table / parks seconds of work behind it / returns an error.  Every rank must drop
    the route together after the checked trial, the reports must run on torch.distributed and be right."""
    import ctypes
    import os

    import torch

    from nvrx_straggler import backend as backend_mod
    from nvrx_straggler import peer_exchange, rccl_direct
    from nvrx_straggler.reporting import ReportGenerator

    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    be = backend_mod.get_backend()
    # how many spin cycles are a second on this box (the spin kernel's clock is not specified)
    for n_spin in (2_000_000, 20_000_000, 40_000_000):   # (the last, longest one is measured with the clocks up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.cuda._sleep(n_spin)
        torch.cuda.synchronize()
        cycles_per_s = n_spin / max(time.perf_counter() - t0, 1e-4)
    state = {"calls": 0, "closed": 0, "aborted": 0}
    last = rank == world - 1

    class FaultyRoute:
        route = f"fault-injected exchange ({fault})"

        def __init__(self, inner):
            self.inner = inner
            proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p)
            self._cb = proto(self._entry)   # what nvrx_report would call between its two kernels
            self.fn_address = ctypes.cast(self._cb, ctypes.c_void_p).value
            self.comm_address = 0

        def _entry(self, send, recv, count, dtype, comm, stream):
            try:
                self.all_gather(send, recv, count, stream)
                return 0
            except Exception:  # noqa: BLE001
                return 5

        def all_gather(self, send_ptr, recv_ptr, count, stream_handle):
            state["calls"] += 1
            if fault == "error":
                raise RuntimeError("injected: the exchange function fails")
            self.inner.all_gather(send_ptr, recv_ptr, count, stream_handle)
            if last and fault == "wrong_table":
                hip.hipMemsetD32Async(recv_ptr, 0x42280000, count, stream_handle)   # 42.0f over rank 0's row of the table
            if last and fault == "never_completes":
                with torch.cuda.stream(be.stream):
                    torch.cuda._sleep(int(4.0 * cycles_per_s))                        # seconds: far beyond the trial's patience (0.5 s)

        def exchange(self, ws, backend):
            self.all_gather(ws.send_ptr, ws.table_ptr, ws.local_ranks * ws.L, backend.stream_handle)
            return ws.table

        def close(self):
            state["closed"] += 1
            self.inner.close()

        def abort(self):
            state["aborted"] += 1
            self.inner.close()

    def fake_create(group=None, device_index=None):
        inner = peer_exchange.create(group, device_index, 1.5)
        assert inner is not None, "the peer windows (the working exchange under the fault) did not come up"
        return FaultyRoute(inner)

    gen = ReportGenerator(scenario["scores_to_compute"], gather_on_rank0=scenario["gather_on_rank0"], node_name=f"node{rank}")
    reports = []
    t_first = None
    with mock.patch.object(rccl_direct, "create", fake_create):
        for i, step in enumerate(scenario["steps"]):
            sec, ker = step[rank]
            t0 = time.perf_counter()
            rep = gen.generate_report(_summ(sec), _summ(ker))
            if i == 0:
                t_first = time.perf_counter() - t0
            reports.append(report_to_plain(rep, scenario.get("thresholds", [0.75])))
    info = dict(gen.exchange_info)
    direct = gen._direct is not None
    gen.close()
    torch.cuda.synchronize()
    return {"reports": reports, "info": info, "direct": direct, "state": state, "first_report_s": t_first}


def detector_mode_agreement(rank, world):
    """Rank 0 believes it times GPU work per kernel, rank 1 per region: the first collective report MIN-reduces the mode
    code and rank 0 follows (``Detector._agree_timing_mode``); the comparison is made once per process group."""
    import logging

    from nvrx_straggler import Detector, ktrace

    records = []

    class _Grab(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())

    log = logging.getLogger("nvrx_straggler.straggler")
    log.addHandler(_Grab())
    log.setLevel(logging.INFO)
    Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=True, node_name=f"host{rank}")
    try:
        switched = []
        if rank == 0:  # the host logic of the switch is what is under test: the profiler swap itself needs the tracer (GPU twin)
            mgr = Detector.cupti_manager  # (built while the mode still says "stamp": no tracer is registered on a CPU host --
            #  the registration's thread check can refuse on a loaded machine, and nothing of the tracer is needed here)
            ktrace._mode, ktrace._mode_note = "kernels", "forced by the test"
            mgr.per_kernel = True
            mgr.switch_to_regions = lambda: (switched.append(1), setattr(mgr, "per_kernel", False), True)[-1]
        for _ in range(3):
            for _ in range(4):
                with Detector.detection_section("s", profile_cuda=False):
                    time.sleep(0.001)
            Detector.generate_report()
        return {"switched": len(switched), "mode": ktrace.timing_mode(), "note": ktrace.mode_note(),
                "log": [m for m in records if "nvrx straggler" in m]}
    finally:
        ktrace._reset_mode_for_tests()
        Detector.shutdown()


def detector_mode_agreement_deferred(rank, world):
    """The same, but rank 0's switch cannot happen at the first report (a profiled region is open there): the comparison
    must NOT be repeated by rank 0 alone (its lone all-reduce would pair with the peers' next collective) -- the
    switch is applied, without any collective, at the first report that finds no region open.  ``dist.all_reduce`` calls with
    an int32 tensor of two elements (the mode code and its negative) are counted on every rank."""
    import torch.distributed as dist

    from nvrx_straggler import Detector, ktrace

    agreements = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        if t.dtype == torch.int32 and t.numel() == 2:
            agreements.append(1)
        return real_all_reduce(t, *a, **k)

    import torch

    dist.all_reduce = counting_all_reduce
    Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=True, node_name=f"host{rank}")
    try:
        switched, refused = [], []
        region_open = [True]
        if rank == 0:
            mgr = Detector.cupti_manager  # (built while the mode still says "stamp": no tracer is registered on a CPU host --
            #  the registration's thread check can refuse on a loaded machine, and nothing of the tracer is needed here)
            ktrace._mode, ktrace._mode_note = "kernels", "forced by the test"
            mgr.per_kernel = True

            def switch():
                if region_open[0]:
                    refused.append(1)
                    return False
                switched.append(1)
                mgr.per_kernel = False
                return True

            mgr.switch_to_regions = switch
        pending_after = []
        for i in range(4):
            for _ in range(4):
                with Detector.detection_section("s", profile_cuda=False):
                    time.sleep(0.001)
            if i == 2:
                region_open[0] = False     # from the third report on, no region is open at report time
            Detector.generate_report()
            pending_after.append(Detector._pending_region_switch is not None)
        return {"agreements": len(agreements), "switched": len(switched), "refused": len(refused), "pending_after": pending_after,
                "mode": ktrace.timing_mode()}
    finally:
        dist.all_reduce = real_all_reduce
        ktrace._reset_mode_for_tests()
        Detector.shutdown()


def detector_trace_budget(rank, world, cost_ms_by_rank, budget_pct, step_ms=12.0, iters=40, profiling_interval=1, dispatches_per_entry=0):
    """Per-kernel mode emulated at the profiler's surface (the tracer itself needs a GPU): ``start()`` of the profiler costs
    this rank ``cost_ms_by_rank[rank]`` -- what tracing the section's kernels costs -- and a step sleeps ``step_ms``.  The loop
    is the reference's (one section per iteration, ``generate_report_if_interval_elapsed`` after it).  Returns what the
    calibration decided and which entries were traced."""
    import logging

    from nvrx_straggler import Detector, ktrace

    traced = []
    records = []

    class _Grab(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())

    log = logging.getLogger("nvrx_straggler.straggler")
    log.addHandler(_Grab())
    log.setLevel(logging.INFO)
    cost = cost_ms_by_rank[rank] / 1e3
    KP = ktrace.KernelTraceProfiler
    saved = (ktrace._mode, ktrace._mode_note, ktrace._setup_error, ktrace.setup, KP._ensure_ready, KP.start, KP.stop)
    ktrace._mode, ktrace._mode_note, ktrace._setup_error = "kernels", "forced by the test", None
    ktrace.setup = lambda *a, **k: None
    KP._live = None
    KP._ensure_ready = lambda self: None

    def start(self, key=""):
        self._started = True
        traced.append(Detector.custom_sections["train_step"].total_entry_cnt - 1)
        if dispatches_per_entry:  # "the section launched this many kernels while traced": counted as enqueued, like the ENQUEUE callback
            ktrace.load().nvrx_ktrace_feed(None, int(dispatches_per_entry), 1)
        time.sleep(cost)

    KP.start = start
    KP.stop = lambda self, *a: (setattr(self, "_started", False), False)[1]
    Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=True, node_name=f"host{rank}",
                        report_time_interval=3600.0, kernel_trace_budget_pct=budget_pct, profiling_interval=profiling_interval)
    try:
        for _ in range(iters):
            with Detector.detection_section("train_step", profile_cuda=True):
                time.sleep(step_ms / 1e3)
            assert Detector.generate_report_if_interval_elapsed() is None
        return {"every": Detector._trace_every, "cost_pct": Detector.kernel_trace_cost_pct, "traced": traced,
                "dispatches": Detector.kernel_trace_dispatches,
                "cpu_samples": len(Detector.custom_sections["train_step"].cpu_elapsed_times),
                "iter_interval": Detector.report_interval_tracker.iter_interval,
                "log": [m for m in records if "budget" in m]}
    finally:
        Detector.shutdown()
        (ktrace._mode, ktrace._mode_note, ktrace._setup_error, ktrace.setup, KP._ensure_ready, KP.start, KP.stop) = saved
        KP._live = None
        ktrace._reset_mode_for_tests()


def detector_c10d_route(rank, world):
    """``NVRX_EXCHANGE=c10d`` in ONE rank's environment (the other one asks for the in-stream RCCL route by name): every rank
    keeps the report's exchange on torch.distributed."""
    os.environ["NVRX_EXCHANGE"] = "c10d" if rank == 1 else "rccl"
    from nvrx_straggler import Detector

    Detector.initialize(scores_to_compute=["relative_perf_scores"], gather_on_rank0=False, node_name=f"host{rank}")
    try:
        out = []
        for _ in range(2):
            for _ in range(6):
                with Detector.detection_section("s", profile_cuda=False):
                    time.sleep(0.002 if rank == 0 else 0.004)
            rep = Detector.generate_report()
            out.append({r: round(v, 2) for r, v in rep.section_relative_perf_scores["s"].items()})
        return {"route": Detector.reporter.exchange_info.get("route", ""), "direct": Detector.reporter._direct is not None, "scores": out}
    finally:
        os.environ.pop("NVRX_EXCHANGE", None)
        Detector.shutdown()


def detector_soak_ranks(rank, world, seconds, seed, gpu=True):
    """Randomised Detector cycles on every rank of a job (tools/soak_mp.py, tests/test_gpu_multiproc.py): synchronous and asynchronous
    generators, gathered on rank 0 or not, a GPU-timed section every rank has, one that comes and goes per rank on a second
    stream, a section only one rank ever has (names the others must learn), reports read at once.  Collective decisions come
    from a generator seeded the same on every rank, local ones from a rank-seeded one; the end of the run is agreed by an
    all-reduce.  Every report is checked for what must hold whatever the timing: who gets a report, NUM of the common
    section, a score entry for every rank.  Returns counts."""
    import faulthandler
    import math

    import nvrx_straggler  # noqa: F401  (per-kernel timing: the tracer registers here, before this process' first HIP call)
    import torch
    import torch.distributed as dist
    from nvrx_straggler import Detector, Statistic, ktrace

    faulthandler.dump_traceback_later(seconds + 90.0, exit=True)  # (a rank that never comes back says where it is)
    if gpu:
        torch.cuda.set_device(0)
    trace_dir = os.environ.get("NVRX_SOAK_TRACE_DIR", "")
    log = None
    if trace_dir:  # every collective this rank issues, in order, to a file of its own (to see where two ranks part ways)
        log = open(os.path.join(trace_dir, f"trace_rank{rank}.log"), "w", buffering=1)
        for name in ("all_reduce", "all_gather", "all_gather_object", "all_gather_into_tensor", "gather", "broadcast", "barrier"):
            orig = getattr(dist, name)

            def wrapped(*a, __orig=orig, __name=name, **k):
                shape = tuple(a[0].shape) if a and hasattr(a[0], "shape") else (len(a[0]) if a and isinstance(a[0], list) else "")
                log.write(f"    {__name} {shape} ...\n")
                out = __orig(*a, **k)
                log.write(f"    {__name} done\n")
                return out

            setattr(dist, name, wrapped)

    def say(msg):
        if log is not None:
            log.write(msg + "\n")
    shared = np.random.default_rng(seed)
    own = np.random.default_rng(seed + 1000 * (rank + 1))
    # (gpu=False: the same flows on the CPU checker backend of the tests -- the generators' protocol is host logic -- with
    #  sections that time no GPU work)
    x = torch.randn(256, 256, device="cuda") if gpu else torch.randn(32, 32)
    side = torch.cuda.Stream() if gpu else None
    t_end = time.time() + seconds
    per_kernel = ktrace.timing_mode() == "kernels"
    counts = {"cycles": 0, "reports": 0, "asynchronous_cycles": 0, "mode": ktrace.timing_mode()}
    while True:
        go = torch.tensor([1.0 if time.time() < t_end else 0.0])
        if world > 1:
            dist.all_reduce(go, op=dist.ReduceOp.MIN)
        if go.item() == 0.0:
            break
        asynchronous = bool(shared.random() < 0.5)
        gather = bool(shared.random() < 0.5)
        scores = ["all", ["relative_perf_scores"], ["individual_perf_scores"]][int(shared.integers(0, 3))]
        by_tracker = bool(shared.random() < 0.25)  # reports when the interval tracker says so (its 16 timed iterations, its all-reduce,
        #                                             the per-kernel tracing budget's calibration riding on it) instead of by hand
        every = int(shared.choice([1, 1, 3])) if by_tracker else 1
        Detector.initialize(scores_to_compute=scores, gather_on_rank0=gather, node_name=f"node{rank}", asynchronous=asynchronous,
                            profiling_interval=every, **({"report_time_interval": 0.002} if by_tracker else {}))
        counts["asynchronous_cycles"] += int(asynchronous)
        counts["tracker_cycles"] = counts.get("tracker_cycles", 0) + int(by_tracker)
        say(f"cycle {counts['cycles']} asynchronous={asynchronous} gather={gather} scores={scores} by_tracker={by_tracker} every={every}")

        def one_step():
            with Detector.detection_section("fwd", profile_cuda=gpu):
                y = x @ x  # noqa: F841
            if own.random() < 0.5:
                if gpu:
                    with torch.cuda.stream(side):
                        with Detector.detection_section("side", profile_cuda=True):
                            z = x + 1  # noqa: F841
                else:
                    with Detector.detection_section("side", profile_cuda=False):
                        z = x + 1  # noqa: F841
            if own.random() < 0.3:
                with Detector.detection_section(f"only_rank{rank}", profile_cuda=False):
                    pass
            with Detector.detection_section("cpu", profile_cuda=False):
                pass

        def check(report, steps):
            if gather and rank != 0:
                assert report is None
                return
            assert report is not None
            if steps is not None:
                assert report.local_section_summaries["fwd"][Statistic.NUM] == steps, (report.local_section_summaries, steps)
            ranks = set(range(world)) if gather else {rank}
            if scores == "all" or "relative_perf_scores" in scores:
                rel = report.section_relative_perf_scores["fwd"]
                assert set(rel) == ranks and all(v > 0.0 and math.isfinite(v) for v in rel.values()), rel
                g = report.gpu_relative_perf_scores
                assert set(g) == ranks, g
                if gpu and steps is not None and not (per_kernel and asynchronous):  # (an asynchronous per-kernel window may hold no kernel samples yet: NaN)
                    assert all(math.isfinite(v) and v > 0.0 for v in g.values()), (
                        g, ktrace.mode_note(), ktrace.counters() if per_kernel else None, report.local_kernel_summaries, steps, asynchronous, gather)
            if scores == "all" or "individual_perf_scores" in scores:
                ind = report.section_individual_perf_scores["fwd"]
                assert set(ind) == ranks and all(v > 0.0 and math.isfinite(v) for v in ind.values()), ind
            report.identify_stragglers()

        try:
            if by_tracker:
                got = 0
                for _i in range(int(shared.integers(40, 120))):
                    one_step()
                    elapsed_before = Detector.report_interval_tracker.iter_interval is not None
                    report = Detector.generate_report_if_interval_elapsed()
                    if report is not None:
                        check(report, None)
                        got += 1
                    del elapsed_before
                counts["reports"] += got
                counts["tracker_reports"] = counts.get("tracker_reports", 0) + got
            else:
                for _ in range(int(shared.integers(2, 7))):
                    steps = int(shared.integers(1, 8))
                    for _s in range(steps):
                        one_step()
                    say(f"  report {counts['reports']} steps={steps} sections={sorted(n for n, c in Detector.custom_sections.items() if len(c.cpu_elapsed_times))}")
                    report = Detector.generate_report()
                    say(f"  report {counts['reports']} returned")
                    counts["reports"] += 1
                    check(report, steps)
        finally:
            Detector.shutdown()
        counts["cycles"] += 1
    if gpu:
        torch.cuda.synchronize()
    faulthandler.cancel_dump_traceback_later()
    return counts
