#!/usr/bin/env python3
"""How close is the CPU baseline's PORT (oracle.ref_port_*, what bench.py times on the GPU box, where the reference
tree does not exist) to the REAL reference on the same host?  Build container only (needs /root/reference).

Times, for one rank x 64 sections x 10 000 samples held as Python floats in deques (straggler.py:80-83,343):
  * the real ``Detector._get_section_summaries`` (straggler.py:172-197) + ``ReportGenerator.generate_report``
    (reporting.py:421-554) of the reference, imported from /root/reference/src with the stub native module;
  * the port used by bench.py (oracle.ref_port_section_summaries + RefPortReportGenerator).
Prints both and the ratio; the figure is quoted in DESIGN.md section 6.
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
    print(f"port / reference = {(m(t_port_sum) + m(t_port_rep)) / (m(t_ref_sum) + m(t_ref_rep)):.3f}")


if __name__ == "__main__":
    main()
