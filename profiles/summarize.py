#!/usr/bin/env python
"""Condense rocprofv3 CSV output into small per-kernel summaries (kept under profiles/).

usage: summarize.py stats <dir> <out.csv>      -- kernel-trace: calls, total/avg/min/max ns per kernel
       summarize.py pmc   <dir> <out.csv>      -- counter collection: per-kernel mean of each counter
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name if len(name) <= 110 else name[:107] + "..."


def stats(d, out):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in rows.values()) or 1
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"])
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([short(k), len(v), sum(v), sum(v) // len(v), min(v), max(v), "%.2f" % (100.0 * sum(v) / tot)])


def pmc(d, out):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + ["mean_" + c for c in counters])
        for k in sorted(acc, key=lambda k: -max(len(v) for v in acc[k].values())):
            n = max(len(v) for v in acc[k].values())
            w.writerow([short(k), n] + ["%.6g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in counters])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
