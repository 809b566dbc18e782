#!/usr/bin/env python3
"""How close is the CPU baseline's PORT (oracle.ref_port_*, what bench.py times on the GPU box, where the reference
tree does not exist) to the REAL reference on the same host?  Build container only (needs /root/reference).

Times, for one rank x 64 sections x 10 000 samples held as Python floats in deques (straggler.py:80-83,343):
  * the real ``Detector._get_section_summaries`` (straggler.py:172-197) + ``ReportGenerator.generate_report``
    (reporting.py:421-554) of the reference, imported from /root/reference/src with the stub native module;
  * the port used by bench.py (oracle.ref_port_section_summaries + RefPortReportGenerator).
Then the whole job both ways -- 8 gloo processes running the REAL reference's report path, and the port's 8-process job
(oracle/port_mp.py) -- in the same run.  Prints the figures; ``--write`` stores them as tests/golden/port_vs_reference.json,
the fixture bench.py quotes next to its cpu_baseline (the reference's Python cannot travel to the GPU box; its ratio can).
"""
import collections
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)

import make_golden  # noqa: E402  (tests/golden: installs the reference with the stub native module)
import synth  # noqa: E402

S, N, REPS = 64, 10_000, 15


def main():
    import torch

    torch.set_num_threads(1)
    straggler = make_golden._install_reference()
    from nvidia_resiliency_ext.attribution.straggler.straggler import CustomSection, Detector

    from oracle import oracle

    x = synth.stress_samples(0, S, N)
    CustomSection.max_elapseds_len = N
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name="n")
    names = [synth.section_name(s) for s in range(S)]
    for s, name in enumerate(names):
        Detector.custom_sections[name] = CustomSection(name=name, location="t")
    t_ref_sum, t_ref_rep = [], []
    for _ in range(REPS):
        for s, name in enumerate(names):
            d = Detector.custom_sections[name].cpu_elapsed_times
            d.clear()
            d.extend(x[s].astype(np.float64).tolist())
        t0 = time.perf_counter()
        summ = Detector._get_section_summaries()
        t1 = time.perf_counter()
        Detector.reporter.generate_report(summ, {})
        t2 = time.perf_counter()
        t_ref_sum.append(t1 - t0)
        t_ref_rep.append(t2 - t1)
    Detector.shutdown()

    deques = {names[s]: collections.deque(x[s].astype(np.float64).tolist(), maxlen=N) for s in range(S)}
    port = oracle.RefPortReportGenerator(1)
    t_port_sum, t_port_rep = [], []
    for _ in range(REPS):
        t0 = time.perf_counter()
        summ = oracle.ref_port_section_summaries(deques)
        t1 = time.perf_counter()
        port.generate_reports([{n: dict(v) for n, v in summ.items()}], [{}])
        t2 = time.perf_counter()
        t_port_sum.append(t1 - t0)
        t_port_rep.append(t2 - t1)
    m = lambda v: float(np.median(v)) * 1e3  # noqa: E731
    print(f"host: {os.cpu_count()} cpus, torch {torch.__version__}, 1 thread")
    print(f"reference : summaries {m(t_ref_sum):7.2f} ms + generate_report {m(t_ref_rep):6.2f} ms = {m(t_ref_sum) + m(t_ref_rep):7.2f} ms")
    print(f"port      : summaries {m(t_port_sum):7.2f} ms + scoring         {m(t_port_rep):6.2f} ms = {m(t_port_sum) + m(t_port_rep):7.2f} ms")
    ratio1 = (m(t_port_sum) + m(t_port_rep)) / (m(t_ref_sum) + m(t_ref_rep))
    print(f"port / reference = {ratio1:.3f}")
    single = {"reference_ms": round(m(t_ref_sum) + m(t_ref_rep), 3), "reference_summaries_ms": round(m(t_ref_sum), 3),
              "reference_generate_report_ms": round(m(t_ref_rep), 3), "port_ms": round(m(t_port_sum) + m(t_port_rep), 3),
              "port_summaries_ms": round(m(t_port_sum), 3), "port_scoring_ms": round(m(t_port_rep), 3), "ratio_port_over_reference": round(ratio1, 4)}

    # the whole job, as bench.py's cpu_baseline times it: 8 gloo processes, one torch thread each -- once with the REAL
    # reference (its Detector._get_section_summaries + ReportGenerator.generate_report, real collectives), once with the port
    job_ref = run_reference_job(8, S, N, reps=12)
    from oracle import port_mp

    job_port = port_mp.run(world=8, sections=S, samples=N, reps=12)
    ratio8 = job_port["report_us"] / job_ref["report_us"]
    print(f"8 gloo ranks: reference {job_ref['report_us'] / 1e3:7.2f} ms per report (summaries {job_ref['summaries_us'] / 1e3:.2f} + generate_report "
          f"{job_ref['generate_report_us'] / 1e3:.2f}), port {job_port['report_us'] / 1e3:7.2f} ms; port / reference = {ratio8:.3f}")
    out = {
        "what": "the CPU baseline's PORT (oracle/port_mp.py, what bench.py can time on a GPU box) against the REAL reference on the same "
                "host, same inputs, same run: 64 sections x 10 000 samples per rank from Python deques",
        "generated_by": "tools/port_vs_reference_timing.py (build container: the only place /root/reference exists; a Python reference "
                        "cannot travel to the GPU box, so the ratio travels instead)",
        "host": {"cpus": os.cpu_count(), "torch": torch.__version__, "torch_threads_per_process": 1},
        "single_rank_no_collectives": single,
        "job_8_gloo_ranks": {"reference_report_us": round(job_ref["report_us"], 1), "reference_summaries_us": round(job_ref["summaries_us"], 1),
                             "reference_generate_report_us": round(job_ref["generate_report_us"], 1),
                             "port_report_us": round(job_port["report_us"], 1), "port_summaries_us": round(job_port["summaries_us"], 1),
                             "port_exchange_scoring_us": round(job_port["exchange_scoring_us"], 1),
                             "ratio_port_over_reference": round(ratio8, 4), "reps": 12},
    }
    if "--write" in sys.argv:
        import json

        path = os.path.join(REPO, "tests", "golden", "port_vs_reference.json")
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        print("wrote", path)


def _reference_worker(rank, world, store, sections, samples, reps):
    """One rank of the REAL reference's report path on gloo (straggler.py:172-197, reporting.py:421-554), unmodified."""
    import json

    import torch
    import torch.distributed as dist

    torch.set_num_threads(1)
    make_golden._install_reference()
    from nvidia_resiliency_ext.attribution.straggler.straggler import CustomSection, Detector

    dist.init_process_group("gloo", init_method=f"file://{store}", world_size=world, rank=rank)
    x = synth.stress_samples(rank, sections, samples)
    CustomSection.max_elapseds_len = samples
    Detector.initialize(scores_to_compute="all", gather_on_rank0=True, node_name=f"n{rank}")
    names = [synth.section_name(s) for s in range(sections)]
    for name in names:
        Detector.custom_sections[name] = CustomSection(name=name, location="t")
    for s, name in enumerate(names):
        Detector.custom_sections[name].cpu_elapsed_times.extend(x[s].astype(np.float64).tolist())
    t_sum, t_rep = [], []
    for _ in range(reps + 1):
        dist.barrier()
        t0 = time.perf_counter()
        summ = Detector._get_section_summaries()
        t1 = time.perf_counter()
        Detector.reporter.generate_report(summ, {})
        t2 = time.perf_counter()
        t_sum.append(t1 - t0)
        t_rep.append(t2 - t1)
    t_sum, t_rep = t_sum[1:], t_rep[1:]   # (the first report exchanges the names: cold)
    res = torch.tensor([np.median(t_sum), np.median(t_rep), np.median(np.asarray(t_sum) + np.asarray(t_rep))], dtype=torch.float64)
    dist.all_reduce(res, op=dist.ReduceOp.MAX)
    if rank == 0:
        print("RESULT " + json.dumps(res.tolist()), flush=True)
    dist.barrier()
    Detector.shutdown()
    dist.destroy_process_group()


def run_reference_job(world, sections, samples, reps):
    import json
    import subprocess
    import tempfile

    with tempfile.NamedTemporaryFile(delete=True) as f:
        store = f.name
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), store, str(sections), str(samples), str(reps)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")]
    if not line:
        raise RuntimeError("reference job: rank 0 produced no result: " + outs[0][1][-600:])
    summaries, report, both = (v * 1e6 for v in json.loads(line[-1][len("RESULT "):]))
    return {"summaries_us": summaries, "generate_report_us": report, "report_us": both}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        _reference_worker(*(int(v) if i != 2 else v for i, v in enumerate(sys.argv[2:8])))
    else:
        main()
