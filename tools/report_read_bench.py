#!/usr/bin/env python3
"""Host cost of READING a steady-state report (no GPU needed): a synthetic 8 x 64 result block is wrapped exactly as
``ReportGenerator._report_from_plan`` wraps the pinned block, then ``identify_stragglers()`` (flag-byte path) and the six
mappings are built.  Reference behaviour being matched: reporting.py:535-545 returns populated dicts.

    python tools/report_read_bench.py [--ranks 8] [--sections 64] [--kernels 0] [--reps 2000]
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nvidia-resiliency-ext_amd"))

from nvrx_straggler import reporting  # noqa: E402


def synthetic_source(R, S, K, flagged_rank=None):
    W = 2 + 2 * S
    rows = S + K

    def al(n):
        return (n + 63) // 64 * 64

    off_s = 64
    off_f = off_s + al(R * W * 4)
    off_t = off_f + al(R * W)
    blob = np.zeros(off_t + al(rows * 32), dtype=np.uint8)
    rng = np.random.default_rng(0)
    scores = rng.uniform(0.8, 1.0, (R, W)).astype(np.float32)
    flags = np.zeros((R, W), dtype=np.uint8)
    if flagged_rank is not None:
        scores[flagged_rank] = 0.6
        flags[flagged_rank] = 1
    blob[off_s:off_s + R * W * 4] = scores.view(np.uint8).ravel()
    blob[off_f:off_f + R * W] = flags.ravel()
    st = rng.uniform(1.0, 2.0, (rows, 8)).astype(np.float32)
    st[:, 5] = 10000
    blob[off_t:off_t + rows * 32] = st.view(np.uint8).ravel()
    v = reporting._View()
    v.S = S
    v.ranks = range(R)
    v.names = [f"section_{i:03d}" for i in range(S)]
    v.cols = {n: i for i, n in enumerate(v.names)}
    v.has_rel = v.has_indiv = True
    v.section_rows = {n: i for i, n in enumerate(v.names)}
    v.kernel_rows = {f"kernel_{i:04d}_blk_1_1_1_grid_2_2_2": S + i for i in range(K)}
    v.layout = (off_s, off_f, off_t, R, W, 0, R, rows)
    v.thresholds = (0.75, 0.75, 0.75, 0.75)
    reporting._finish_view(v) if hasattr(reporting, "_finish_view") else None
    rank_to_node = {r: f"node{r}" for r in range(R)}
    return lambda: reporting.Report._from_device(reporting._ScoreSource(v, blob), rank_to_node, 0.02, True, 0)


def timeit(fn, make, reps):
    t = []
    for _ in range(reps):
        rep = make()
        t0 = time.perf_counter_ns()
        fn(rep)
        t.append(time.perf_counter_ns() - t0)
    return float(np.median(t)) / 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--sections", type=int, default=64)
    ap.add_argument("--kernels", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2000)
    a = ap.parse_args()
    for flagged in (None, 3):
        make = synthetic_source(a.ranks, a.sections, a.kernels, flagged)
        out = {"flagged_rank": flagged}
        out["identify_stragglers_us"] = timeit(lambda r: r.identify_stragglers(), make, a.reps)
        for f in sorted(reporting._LAZY_FIELDS):
            out[f + "_us"] = timeit(lambda r, f=f: getattr(r, f), make, a.reps)
        out["all_six_us"] = timeit(lambda r: r._materialise(), make, a.reps)
        out["identify_plus_all_six_us"] = timeit(lambda r: (r.identify_stragglers(), r._materialise()), make, a.reps)
        print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
