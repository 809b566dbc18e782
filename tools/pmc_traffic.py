#!/usr/bin/env python3
"""profiles/pmc_row_stats.json from two PMC passes alone (no --stats pass):
    tools/pmc_traffic.py <tag> <pmc_fetch_dir> <pmc_write_dir>
The same arithmetic as tools/summarize_profiles.py (read side = 2 x FETCH_SIZE x 1024 on gfx950, write side = WRITE_SIZE x
1024, per launch, by launch size), keyed by the sha of k_row_stats' text."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_profiles as sp  # noqa: E402

tag, fetch_dir, write_dir = sys.argv[1:4]
f, w = sp.pmc(fetch_dir, "FETCH_SIZE"), sp.pmc(write_dir, "WRITE_SIZE")
d = {"kernel_source_sha16": sp.kernel_source_sha(), "source": tag, "rows": {},
     "note": "read side = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md HBM); write side = WRITE_SIZE x 1024"}
for (k, wgs), (n, fv) in sorted(f.items()):
    if k != "k_row_stats" or n < 5:
        continue
    wv = w.get((k, wgs), (0, 0.0))[1]
    d["rows"][str(wgs)] = {"hbm_bytes_per_launch": int(2 * fv * 1024 + wv * 1024), "fetch_size_kib_avg": fv,
                           "write_size_kib_avg": wv, "dispatches": n}
    print(f"k_row_stats, {wgs} rows: {d['rows'][str(wgs)]['hbm_bytes_per_launch'] / 1e6:.2f} MB per launch over {n} dispatches")
if d["rows"]:
    json.dump(d, open(os.path.join(sp.REPO, "profiles", "pmc_row_stats.json"), "w"), indent=1)
    print("profiles/pmc_row_stats.json written for sha", d["kernel_source_sha16"])
else:
    print("no k_row_stats dispatches found: nothing written")
