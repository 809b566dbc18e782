"""CPU baseline with real collectives: the reference's report path restated (oracle.ref_port_*), one process
per rank on gloo.  TEST / BASELINE INFRASTRUCTURE ONLY -- nothing in the product imports this file.

What one timed report does on every rank (reference file:line it restates):

* ``Detector._get_section_summaries`` (straggler.py:172-197): ``torch.tensor(deque)`` + min / max / median /
  mean / std per section, over this rank's Python deques of floats (straggler.py:80-83);
* ``ReportGenerator.generate_report`` steady state (reporting.py:421-554): the "every rank has all names" flag
  all-reduce (C1, name_mapper.py:68-69), the f32 MIN all-reduce of the medians (C4, reporting.py:255-296),
  section scores reference / MED (:196-217), the pack into a (2+2S) f32 tensor and the gather to rank 0 (C5,
  :338-419);
* separately timed: ``all_gather_object`` of the per-rank summary dicts -- the exchange BASELINE.json's wording
  names ("Python all_gather_object + NumPy scoring"); the reference itself only pickles names, on the first report.

``run()`` starts ``world`` child interpreters (gloo, file-store rendezvous), one torch thread each, and returns the per-report times as max over ranks of each rank's median.
"""
from __future__ import annotations

import collections
import os
import sys
import tempfile
import time
import traceback
from typing import Dict

import numpy as np


def _worker(rank: int, world: int, store: str, sections: int, samples: int, reps: int, q) -> None:
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        repo = os.path.dirname(here)
        for p in (repo, os.path.join(repo, "tests", "golden")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch
        import torch.distributed as dist

        import synth
        from oracle import oracle

        torch.set_num_threads(1)
        dist.init_process_group("gloo", init_method=f"file://{store}", world_size=world, rank=rank)
        x = synth.stress_samples(rank, sections, samples)
        names = [synth.section_name(s) for s in range(sections)]
        deques = {names[s]: collections.deque(x[s].astype(np.float64).tolist(), maxlen=samples) for s in range(sections)}
        hist_min: Dict[str, float] = collections.defaultdict(lambda: float("inf"))
        t_sum, t_score, t_obj = [], [], []
        summ = None
        for _ in range(reps):
            dist.barrier()
            t0 = time.perf_counter()
            summ = oracle.ref_port_section_summaries(deques)                      # straggler.py:172-197
            t1 = time.perf_counter()
            flag = torch.tensor([1.0], dtype=torch.float32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)                            # C1
            for n, s in summ.items():                                              # reporting.py:298-314
                hist_min[n] = min(hist_min[n], s["MED"])
            med = torch.full((sections,), -1.0, dtype=torch.float32)               # reporting.py:267-279
            for i, n in enumerate(names):
                med[i] = summ[n]["MED"]
            dist.all_reduce(med, op=dist.ReduceOp.MIN)                             # C4
            ref = {n: (med[i].item() if med[i].item() >= 0 else float("nan")) for i, n in enumerate(names)}
            rel = {n: ref[n] / s["MED"] for n, s in summ.items()}                  # reporting.py:196-217
            ind = {n: hist_min[n] / s["MED"] for n, s in summ.items()}
            packed = torch.full((2 + 2 * sections,), float("nan"), dtype=torch.float32)  # reporting.py:338-360
            for i, n in enumerate(names):
                packed[2 + i] = ind[n]
                packed[2 + sections + i] = rel[n]
            bucket = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
            dist.gather(packed, gather_list=bucket, dst=0)                          # C5
            if rank == 0:
                _ = {n: {r: bucket[r][2 + sections + i].item() for r in range(world)} for i, n in enumerate(names)}
            t2 = time.perf_counter()
            out = [None] * world
            dist.all_gather_object(out, summ)
            t3 = time.perf_counter()
            t_sum.append(t1 - t0)
            t_score.append(t2 - t1)
            t_obj.append(t3 - t2)
        res = torch.tensor([np.median(t_sum), np.median(t_score), np.median(t_obj),
                            np.median(np.asarray(t_sum) + np.asarray(t_score))], dtype=torch.float64)
        dist.all_reduce(res, op=dist.ReduceOp.MAX)
        if rank == 0:
            q.put(("ok", res.tolist()))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:  # noqa: BLE001
        q.put(("error", traceback.format_exc()))


def run(world: int = 8, sections: int = 64, samples: int = 10_000, reps: int = 20, timeout: float = 240.0) -> Dict[str, float]:
    """Per-report microseconds: summaries, exchange + scoring, all_gather_object of the summaries, and the report
    (summaries + exchange + scoring); each the max over ranks of the per-rank median over ``reps`` reports.

    The ranks are plain child interpreters running this file (no dependence on the caller's ``__main__``)."""
    import json
    import subprocess

    with tempfile.NamedTemporaryFile(delete=True) as f:
        store = f.name
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    me = os.path.abspath(__file__)
    procs = [subprocess.Popen([sys.executable, me, "--worker", str(r), str(world), store, str(sections), str(samples), str(reps)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    try:
        deadline = time.time() + timeout
        for p in procs:
            outs.append(p.communicate(timeout=max(1.0, deadline - time.time())))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")]
    if not line:
        raise RuntimeError("rank 0 produced no result: " + (outs[0][1] or "")[-400:])
    status, payload = json.loads(line[-1][len("RESULT "):])
    if status != "ok":
        raise RuntimeError(payload)
    summaries, scoring, obj, report = (v * 1e6 for v in payload)
    return {"summaries_us": summaries, "exchange_scoring_us": scoring, "all_gather_object_us": obj, "report_us": report,
            "ranks": world, "reps": reps}


class _StdoutQueue:
    """What the worker reports through: rank 0's result as one line on stdout."""

    @staticmethod
    def put(item) -> None:
        import json

        print("RESULT " + json.dumps(item), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        r, w, store_path, n_sec, n_samp, n_reps = sys.argv[2:8]
        _worker(int(r), int(w), store_path, int(n_sec), int(n_samp), int(n_reps), _StdoutQueue)
    else:
        print(run())
