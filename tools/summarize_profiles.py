#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/ into the committed summaries under profiles/.

usage: tools/summarize_profiles.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>] [--local-ranks N]
  <stats_dir>      output of `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py ...`
  <pmc_*_dir>      outputs of separate `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace ...` passes
Writes profiles/<tag>_kernel_stats.csv (verbatim), profiles/<tag>_summary.md and updates
profiles/pmc_row_stats.json (read by bench.py for roofline.traffic).
HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE count units of 1024 B, and on
gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so the read side is
doubled.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"k_row_stats<(\d+), (\d+)>", name)
    if m and m.group(2) != "5":  # <512,5> is the bench shape; other tiles come from the overhead leg
        return f"k_row_stats<{m.group(1)},{m.group(2)}>"
    for k in ("k_row_stats", "k_score", "k_scatter", "k_colmin", "k_send_init", "k_fill_f32", "k_stamp_begin", "k_stamp_end"):
        if k in name:
            return k
    return name[:40]


def pmc(dirname, counter):
    vals = collections.defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in vals.items()}


def main():
    argv = sys.argv[1:]
    local_ranks = 8
    if "--local-ranks" in argv:
        i = argv.index("--local-ranks")
        local_ranks = int(argv[i + 1])
        del argv[i : i + 2]
    tag, stats_dir = argv[0], argv[1]
    out = os.path.join(REPO, "profiles")
    os.makedirs(out, exist_ok=True)
    stats_csv = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats_csv, os.path.join(out, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats_csv)))
    lines = [f"# rocprofv3 summary `{tag}`", "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 200 --warmup 20 "
             f"--no-cpu-baseline --no-overhead` (N=1: {local_ranks} logical ranks x 64 sections x 10000 samples on one MI355X)", "",
             "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                     f"{float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |")
    if len(argv) >= 4:
        f = pmc(argv[2], "FETCH_SIZE")
        w = pmc(argv[3], "WRITE_SIZE")
        lines += ["", "PMC passes (separate runs, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` with `--kernel-trace` only):", "",
                  "| kernel | dispatches | FETCH_SIZE avg (KiB) | read bytes (x2 gfx950 correction) | WRITE_SIZE avg (KiB) |",
                  "|---|---|---|---|---|"]
        for k in sorted(set(f) | set(w)):
            if not k.startswith("k_"):
                continue
            fn, fv = f.get(k, (0, 0.0))
            _, wv = w.get(k, (0, 0.0))
            lines.append(f"| {k} | {fn} | {fv:.1f} | {2 * fv * 1024:.0f} | {wv:.1f} |")
        if "k_row_stats" in f:
            hbm = 2 * f["k_row_stats"][1] * 1024 + w.get("k_row_stats", (0, 0.0))[1] * 1024
            path = os.path.join(out, "pmc_row_stats.json")
            d = json.load(open(path)) if os.path.exists(path) else {}
            d[str(local_ranks)] = {"hbm_bytes_per_launch": int(hbm), "fetch_size_kib_avg": f["k_row_stats"][1],
                                   "write_size_kib_avg": w.get("k_row_stats", (0, 0.0))[1], "source": tag,
                                   "note": "read side = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md HBM)"}
            json.dump(d, open(path, "w"), indent=1)
            lines += ["", f"k_row_stats HBM traffic per launch: {hbm/1e6:.2f} MB "
                          f"(algorithmic {local_ranks * 64 * 10000 * 4 / 1e6:.2f} MB)"]
    open(os.path.join(out, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
