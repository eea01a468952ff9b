#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into small text summaries for profiles/.

  python scripts/summarize_prof.py <rocprof output dir> [more dirs...] > profiles/rNN_summary.txt

Only this repo's kernels (namespace mgl_sw_dev) are listed individually; everything else
(torch's input-generation kernels) is lumped into "other"."""
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    for ns in ("mgl_sw_dev::", "mgl_ph_dev::"):
        if ns in name:
            return name.split(ns)[1].split("(")[0]
    return "other"


def kernel_stats(path):
    rows = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            s = rows[k]
            s[0] += int(r["Calls"])
            s[1] += float(r["TotalDurationNs"])
            s[2] = min(s[2], float(r["MinNs"]))
            s[3] = max(s[3], float(r["MaxNs"]))
    print(f"## kernel stats: {path}")
    print(f"{'kernel':28s} {'calls':>7s} {'total_ms':>12s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s}")
    for k, (c, t, mn, mx) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:28s} {c:7d} {t / 1e6:12.3f} {t / c / 1e6:10.4f} {mn / 1e6:10.4f} {mx / 1e6:10.4f}")
    print()


def counters(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    meta = {}
    dur = defaultdict(lambda: [0, 0.0])
    seen = set()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if k != "other":
                meta[k] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"],
                           r["SGPR_Count"], r["Scratch_Size"])
            key = (r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d = dur[k]
                d[0] += 1
                d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print(f"## counters: {path}")
    for k in sorted(acc):
        if k == "other":
            continue
        wg, lds, vgpr, agpr, sgpr, scr = meta[k]
        n, t = dur[k]
        print(f"{k}: dispatches={n} avg_ms={t / n / 1e6:.4f} workgroup={wg} lds_bytes={lds} vgpr={vgpr} agpr={agpr} "
              f"sgpr={sgpr} scratch={scr}")
        for c, (cnt, tot) in sorted(acc[k].items()):
            print(f"    {c:28s} sum={tot:18.1f}  per_dispatch={tot / cnt:16.2f}")
    print()


def main():
    for d in sys.argv[1:]:
        for p in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
            kernel_stats(p)
        for p in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
            counters(p)


if __name__ == "__main__":
    main()
