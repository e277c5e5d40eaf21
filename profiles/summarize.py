#!/usr/bin/env python3
"""Condenses rocprofv3 output (profiles/collect.sh) into the two summaries kept under profiles/:
   <out>/kernel_stats.csv  per kernel: calls, total / average / min / max duration (ns), share of GPU time
   <out>/pmc.csv           per kernel: launches and the per-launch average of every collected counter
Kernel names are reduced to the function name (template arguments dropped), so the rows read like the source."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name.strip().strip('"'))
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"<.*", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.split("::")[-1].split(" ")[-1]


def main(src, dst):
    # --- kernel time statistics, from the kernel trace (start / end timestamps per dispatch)
    durs = defaultdict(list)
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in durs.values()) or 1
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "percent"])
        for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / total, 3)])
    # --- counters: one row per (dispatch, counter) in *counter_collection.csv
    sums = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k, c = short(r["Kernel_Name"]), r["Counter_Name"]
            sums[k][c] += float(r["Counter_Value"])
            disp[k][c].add((f, r["Dispatch_Id"]))
    counters = sorted({c for k in sums for c in sums[k]})
    with open(os.path.join(dst, "pmc.csv"), "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["kernel", "launches"] + [c + "_avg" for c in counters] + ["valu_insts_per_wave"])
        for k in sorted(sums, key=lambda k: -sums[k].get("SQ_INSTS_VALU", 0.0)):
            n = max((len(disp[k][c]) for c in disp[k]), default=0)
            avg = {c: (sums[k][c] / len(disp[k][c]) if disp[k][c] else "") for c in counters}
            ipw = (avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]) if avg.get("SQ_WAVES") and avg.get("SQ_INSTS_VALU") not in ("", None) else ""
            w.writerow([k, n] + [round(avg[c], 3) if avg[c] != "" else "" for c in counters] + [round(ipw, 3) if ipw != "" else ""])
    print("wrote", os.path.join(dst, "kernel_stats.csv"), "and", os.path.join(dst, "pmc.csv"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
