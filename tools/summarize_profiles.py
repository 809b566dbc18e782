#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/ into the committed summaries under profiles/.

usage: tools/summarize_profiles.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]
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
    m = re.search(r"k_row_stats<(\d+), (\d+)", name)
    if m and m.group(2) != "5":  # <512,5> is the bench shape; other tiles come from the overhead leg
        return f"k_row_stats<{m.group(1)},{m.group(2)}>"
    for k in ("k_row_stats", "k_score1", "k_score", "k_peer_allgather", "k_scatter", "k_colmin", "k_send_init", "k_fill_f32",
              "k_stamp_begin", "k_stamp_end", "k_append_rows"):
        if k in name:
            return k
    return name[:40]


def kernel_source_sha():
    """The same hash bench.py computes: the text of k_row_stats only (its header comment up to the next kernel's)."""
    import hashlib

    with open(os.path.join(REPO, "nvidia-resiliency-ext_amd", "csrc", "nvrx_straggler.hip"), "rb") as f:
        text = f.read()
    a, b = text.find(b"// k_row_stats: one workgroup per timing row."), text.find(b"// k_scatter:")
    return hashlib.sha256(text[a:b] if 0 <= a < b else text).hexdigest()[:16]


def pmc(dirname, counter):
    """{(kernel, rows): (dispatches, mean counter value)}; rows = workgroups of the dispatch (one per timing row for
    k_row_stats), so the bench shape (512 rows) and the N=8 per-GPU shape (64 rows) stay apart."""
    vals = collections.defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                wgs = int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1)
                vals[(short(row["Kernel_Name"]), wgs)].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in vals.items()}


def main():
    argv = sys.argv[1:]
    tag, stats_dir = argv[0], argv[1]
    out = os.path.join(REPO, "profiles")
    os.makedirs(out, exist_ok=True)
    stats_csv = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats_csv, os.path.join(out, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats_csv)))
    cmd = open(os.path.join(stats_dir, "command.txt")).read().strip() if os.path.exists(os.path.join(stats_dir, "command.txt")) else \
        "python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-overhead"
    lines = [f"# rocprofv3 summary `{tag}`", "",
             f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}` (N=1: 8 logical ranks x 64 sections x "
             "10000 samples on one MI355X; kernel source sha " + kernel_source_sha() + ")", "",
             "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
    # (k_stamp_begin is one argument-free instantiation per stamp slot: 64 kernel names, one line here)
    merged, order = {}, []
    for r in rows:
        k = short(r["Name"])
        key = k if k == "k_stamp_begin" else r["Name"]
        calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
        if key not in merged:
            merged[key] = [k, 0, 0.0, float("inf"), 0.0, 0.0]
            order.append(key)
        m = merged[key]
        m[1] += calls
        m[2] += total
        m[3] = min(m[3], float(r["MinNs"]))
        m[4] = max(m[4], float(r["MaxNs"]))
        m[5] += float(r["Percentage"])
    for key in sorted(order, key=lambda q: -merged[q][2]):
        k, calls, total, mn, mx, pct = merged[key]
        lines.append(f"| {k} | {calls} | {total/max(calls,1)/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {pct:.1f} |")
    # the same trace split by launch size: k_row_stats runs at the bench shape (512 rows), at the N=8 per-GPU shape (64
    # rows) and, in the Detector legs, on a handful of rows; the stats table above averages over all of them
    by_size = collections.defaultdict(list)
    for path in glob.glob(os.path.join(stats_dir, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            if k.startswith("k_row_stats") or k.startswith("k_score") or k.startswith("k_peer"):
                wgs = int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1)
                by_size[(k, wgs)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if by_size:
        lines += ["", "Same trace, by launch size (workgroups = timing rows for k_row_stats):", "",
                  "| kernel | workgroups | calls | avg us | median us | min us |", "|---|---|---|---|---|---|"]
        for (k, wgs), v in sorted(by_size.items()):
            if len(v) >= 20:
                v = sorted(v)
                lines.append(f"| {k} | {wgs} | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {v[len(v)//2]/1e3:.2f} | {v[0]/1e3:.2f} |")
    if len(argv) >= 4:
        f = pmc(argv[2], "FETCH_SIZE")
        w = pmc(argv[3], "WRITE_SIZE")
        lines += ["", "PMC passes (separate runs, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` with `--kernel-trace` only; one line per "
                      "kernel and launch size):", "",
                  "| kernel | workgroups | dispatches | FETCH_SIZE avg (KiB) | read bytes (x2 gfx950 correction) | WRITE_SIZE avg (KiB) |",
                  "|---|---|---|---|---|---|"]
        for k in sorted(set(f) | set(w)):
            if not k[0].startswith("k_"):
                continue
            fn, fv = f.get(k, (0, 0.0))
            _, wv = w.get(k, (0, 0.0))
            lines.append(f"| {k[0]} | {k[1]} | {fn} | {fv:.1f} | {2 * fv * 1024:.0f} | {wv:.1f} |")
        d = {"kernel_source_sha16": kernel_source_sha(), "source": tag, "rows": {},
             "note": "read side = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md HBM); write side = WRITE_SIZE x 1024"}
        for (k, wgs), (n, fv) in sorted(f.items()):
            if k != "k_row_stats" or n < 5:
                continue
            wv = w.get((k, wgs), (0, 0.0))[1]
            hbm = 2 * fv * 1024 + wv * 1024
            d["rows"][str(wgs)] = {"hbm_bytes_per_launch": int(hbm), "fetch_size_kib_avg": fv, "write_size_kib_avg": wv, "dispatches": n}
            lines += ["", f"k_row_stats, {wgs} rows: HBM traffic per launch {hbm/1e6:.2f} MB (algorithmic {wgs * 10000 * 4 / 1e6:.2f} MB, "
                          f"ratio {hbm / (wgs * 10000 * 4):.3f})"]
        if d["rows"]:
            json.dump(d, open(os.path.join(out, "pmc_row_stats.json"), "w"), indent=1)
    open(os.path.join(out, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
