"""The score stage as the job grows: `nvrx_score` over a gathered table of R ranks x S sections (K GPU-timed rows),
call -> completion word seen by the host, median of 300 after 30 warm-ups.  R <= 64: `k_score1` (one workgroup);
beyond: `k_colmin` + `k_score` (one workgroup per rank, ticket).  The table is what the all-gather delivers; the
statistics kernel in front of it is per-GPU work and does not grow with R.  Also: reading the flagged set and the
two section-score mappings of such a report on the host.
    python tools/score_scale.py [--S 64] [--K 0]"""
import argparse
import json
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, ".."), os.path.join(here, "..", "nvidia-resiliency-ext_amd")]
import numpy as np
import torch


def table(rng, R, K, S):
    from nvrx_straggler._native import table_len

    KS, L = K + S, table_len(K, S)
    T = np.zeros((R, L), dtype=np.float32)
    med = rng.uniform(1.0, 2.0, (R, KS)).astype(np.float32)
    med[R // 3] *= 1.6  # one straggler
    T[:, :KS] = med
    T[:, KS : 2 * KS] = med * rng.uniform(0.9, 1.0, (R, KS)).astype(np.float32)
    T[:, 2 * KS : 2 * KS + K] = rng.uniform(1, 1000, (R, K)).astype(np.float32)
    T[:, L - 1] = 1.0
    return T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--K", type=int, default=0)
    ap.add_argument("--ranks", type=int, nargs="*", default=[8, 64, 65, 128, 512, 1024, 4096, 16384])
    a = ap.parse_args()
    from nvrx_straggler.backend import get_backend

    be = get_backend()
    rng = np.random.default_rng(7)
    out = []
    for R in a.ranks:
        ws = be.workspace(R, a.K, a.S, R, 0)
        dev = torch.from_numpy(table(rng, R, a.K, a.S)).cuda()
        torch.cuda.synchronize()
        ts = []
        for i in range(330):
            t0 = time.perf_counter_ns()
            be.score(ws, dev, True, True, (0.75, 0.75, 0.75, 0.75))
            ts.append(time.perf_counter_ns() - t0)
        ts = np.array(ts[30:]) * 1e-3
        flagged = int(ws.flags[:, 2 + a.S :].any(axis=1).sum())
        out.append({"R": R, "S": a.S, "K": a.K, "table_KB": round(dev.numel() * 4 / 1024, 1),
                    "result_KB": round((ws.scores.nbytes + ws.flags.nbytes) / 1024, 1),
                    "score_us_median": round(float(np.median(ts)), 2), "score_us_p95": round(float(np.percentile(ts, 95)), 2),
                    "ranks_flagged_relative": flagged})
        print(json.dumps(out[-1]), flush=True)
    return out


if __name__ == "__main__":
    main()
