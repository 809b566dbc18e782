#!/usr/bin/env python3
"""GPU timeline around a report in the training loop of config #4 (rocprofv3 --kernel-trace of this script, then
`--analyze <csv>`): what a report every 10th step does to the step's GEMMs -- their durations and the gaps between them
in the step that runs beside the report's kernels, against steps without a report."""
import csv
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd")):
    sys.path.insert(0, p)


def run(asynchronous, every):
    import torch

    from nvrx_straggler import Detector

    x = torch.randn(4096, 4096, dtype=torch.bfloat16, device="cuda")
    Detector.initialize(scores_to_compute=["individual_perf_scores"], gather_on_rank0=True, node_name="n", asynchronous=asynchronous)
    held = None
    for step in range(1, 241):
        with Detector.detection_section("train_step", profile_cuda=True):
            y = x
            for _ in range(10):
                y = torch.matmul(x, y)
        if step % every == 0:
            rep = Detector.generate_report()
            if asynchronous:
                if held is not None:
                    held.identify_stragglers()
                held = rep
            else:
                rep.identify_stragglers()
    torch.cuda.synchronize()
    Detector.shutdown()


def analyze(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    gemm = [(s, e) for s, e, n in rows if n.startswith("Cijk") or "gemm" in n.lower()]
    ours = [(s, e, n) for s, e, n in rows if n.startswith("k_") or "k_row_stats" in n or "k_score" in n or "k_scatter" in n or "k_stamp" in n]
    print(f"{len(gemm)} GEMMs, {len(ours)} kernels of ours")
    import statistics

    durs = [e - s for s, e in gemm]
    gaps = [gemm[i + 1][0] - gemm[i][1] for i in range(len(gemm) - 1)]
    print("GEMM duration us: median %.1f  p95 %.1f  max %.1f" % (statistics.median(durs) / 1e3, sorted(durs)[int(.95 * len(durs))] / 1e3, max(durs) / 1e3))
    print("gap between consecutive GEMMs us: median %.2f  p95 %.2f  max %.1f" % (statistics.median(gaps) / 1e3, sorted(gaps)[int(.95 * len(gaps))] / 1e3, max(gaps) / 1e3))
    # per report: the window from its first kernel to its last, and what the GEMMs overlapping / following it look like
    reports = [(s, e) for s, e, n in ours if "k_row_stats" in n]
    for rs, re_ in reports[5:12]:
        near = [(s, e) for s, e in gemm if s > rs - 1_200_000 and s < rs + 1_200_000]
        if len(near) < 3:
            continue
        worst = max(near, key=lambda g: g[1] - g[0])
        gg = [near[i + 1][0] - near[i][1] for i in range(len(near) - 1)]
        names = [(n.split("(")[0][:22], (s - rs) / 1e3, (e - s) / 1e3) for s, e, n in ours if rs - 50_000 < s < rs + 200_000]
        print("report at t: longest GEMM within +-1.2 ms %.1f us, largest gap %.1f us, sum of gaps %.1f us | ours (name, start rel us, dur us): %s"
              % ((worst[1] - worst[0]) / 1e3, max(gg) / 1e3, sum(gg) / 1e3, [(a, round(b, 1), round(c, 1)) for a, b, c in names]))


if __name__ == "__main__":
    if sys.argv[1] == "--analyze":
        analyze(sys.argv[2])
    else:
        run(sys.argv[1] == "async", int(sys.argv[2]))
