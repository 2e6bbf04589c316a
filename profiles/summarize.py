#!/usr/bin/env python
"""Condense rocprofv3 CSV output into small per-kernel summaries (kept under profiles/).

usage: summarize.py stats <dir> <out.csv>      -- kernel-trace: calls, total/avg/min/max ns per kernel
       summarize.py pmc   <dir> <out.csv>      -- counter collection: per-kernel mean of each counter
       summarize.py timeline <dir> <out.csv>   -- kernel-trace: every dispatch of the LAST rollout-to-rollout
                                                  interval (one iteration) with its start offset, duration
                                                  and the idle gap before it
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


def timeline(d, out):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    starts = [i for i, e in enumerate(ev) if "rollout_" in e[2]]
    if len(starts) < 2:
        return
    # the last complete iteration the bench timed: second-to-last rollout .. last rollout
    i0, i1 = starts[-2], starts[-1]
    t0 = ev[i0][0]
    busy = idle = 0
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["start_us", "dur_us", "gap_before_us", "kernel"])
        prev_end = t0
        for s_, e_, k in ev[i0:i1]:
            gap = max(0, s_ - prev_end)
            w.writerow(["%.1f" % ((s_ - t0) / 1e3), "%.1f" % ((e_ - s_) / 1e3), "%.1f" % (gap / 1e3), short(k)[:70]])
            busy += e_ - s_
            idle += gap
            prev_end = max(prev_end, e_)
        w.writerow(["%.1f" % ((ev[i1][0] - t0) / 1e3), "", "%.1f" % (max(0, ev[i1][0] - prev_end) / 1e3), "next rollout"])
        w.writerow(["# busy_us=%.1f idle_us=%.1f" % (busy / 1e3, (idle + max(0, ev[i1][0] - prev_end)) / 1e3)])


def pmc(d, out):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    # the kernel durations of the SAME pass (counter collection slows a launch: clocks must be derived from these)
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + ["mean_" + c for c in counters] + ["mean_ns_this_pass"])
        for k in sorted(acc, key=lambda k: -max(len(v) for v in acc[k].values())):
            n = max(len(v) for v in acc[k].values())
            w.writerow([short(k), n] + ["%.6g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in counters]
                       + ["%d" % (sum(dur[k]) // len(dur[k])) if dur.get(k) else ""])


def traffic(fetch_csv, write_csv, out_json, workload="swimmer4096_trpo", n_envs=4096, tag="", sq_csv=None, sq2_csv=None):
    """HBM bytes per launch of the rollout kernel from the FETCH_SIZE / WRITE_SIZE summaries (KB).
    Reads are 4 B/lane plane loads (not the 16 B/lane streams MI355X_MICROARCH.md's x2 correction
    was calibrated on) and are < 1 % of the total here, so they are taken as reported."""
    import json

    def pick(path, col):
        for r in csv.DictReader(open(path)):
            if "rollout_" in r["kernel"]:
                return float(r[col]) * 1024.0
        return None
    rd, wr = pick(fetch_csv, "mean_FETCH_SIZE"), pick(write_csv, "mean_WRITE_SIZE")
    rec = {}
    if os.path.exists(out_json):
        rec = json.load(open(out_json))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec[workload] = dict(n_envs=int(n_envs), rollout_bytes_per_launch=rd + wr, fetch_bytes=rd, write_bytes=wr,
                         kernel_source_hash=bench.kernel_source_hash(),
                         source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (%s)" % tag)
    if sq_csv:
        # the compute axis of the same kernel from its own counters (SQ pass): vector instructions issued, wavefronts,
        # and GRBM_GUI_ACTIVE (summed over the 8 XCDs) for the clock the launch actually ran at
        for r in csv.DictReader(open(sq_csv)):
            if "rollout_" in r["kernel"]:
                rec[workload].update(rollout_insts_valu=float(r["mean_SQ_INSTS_VALU"]),
                                     rollout_waves=float(r["mean_SQ_WAVES"]),
                                     rollout_gui_active=float(r["mean_GRBM_GUI_ACTIVE"]),
                                     rollout_active_inst_valu=float(r["mean_SQ_ACTIVE_INST_VALU"]),
                                     rollout_wave_cycles=float(r["mean_SQ_WAVE_CYCLES"]))
                break
    if sq2_csv:
        # second SQ pass: scalar instructions, matrix instructions and the matrix pipe's busy cycles, time parked at
        # s_waitcnt -- what the issue-slot accounting of a lone wavefront needs beside the vector instructions
        for r in csv.DictReader(open(sq2_csv)):
            if "rollout_" in r["kernel"]:
                for key, col in (("rollout_insts_salu", "mean_SQ_INSTS_SALU"), ("rollout_insts_mfma", "mean_SQ_INSTS_MFMA"),
                                 ("rollout_mfma_busy_cycles", "mean_SQ_VALU_MFMA_BUSY_CYCLES"),
                                 ("rollout_wait_any", "mean_SQ_WAIT_ANY"), ("rollout_insts_vmem", "mean_SQ_INSTS_VMEM"),
                                 ("rollout_insts_smem", "mean_SQ_INSTS_SMEM"),
                                 ("rollout_active_inst_any", "mean_SQ_ACTIVE_INST_ANY")):
                    if r.get(col):
                        rec[workload][key] = float(r[col])
                break
    json.dump(rec, open(out_json, "w"), indent=1, sort_keys=True)


def step_kernels(sq_csv, out_json, n_envs, tag=""):
    """Vector-instruction counters of the per-step boundary kernels (tools/step_kernel_roofline.py under an SQ pass) ->
    pmc_traffic.json["step_kernels"][<env class>]: instructions and wavefronts per launch of ``n_envs`` envs, and the
    clock of that pass -- bench.py's roofline_step_kernel prices its live launch time on the vector axis with them."""
    import json
    import re
    rec = json.load(open(out_json)) if os.path.exists(out_json) else {}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {}
    for r in csv.DictReader(open(sq_csv)):
        m = re.search(r"vecenv_step_kernel<rl::(\w+)>", r["kernel"])
        if not m or not r.get("mean_SQ_INSTS_VALU"):
            continue
        out[m.group(1)] = dict(n_envs=int(n_envs), insts_valu=float(r["mean_SQ_INSTS_VALU"]),
                               waves=float(r["mean_SQ_WAVES"]), gui_active=float(r["mean_GRBM_GUI_ACTIVE"]),
                               active_inst_valu=float(r["mean_SQ_ACTIVE_INST_VALU"]),
                               wave_cycles=float(r["mean_SQ_WAVE_CYCLES"]),
                               ns_this_pass=float(r["mean_ns_this_pass"]) if r.get("mean_ns_this_pass") else None)
    rec["step_kernels"] = dict(kernels=out, kernel_source_hash=bench.kernel_source_hash(),
                               source="rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES / GRBM_GUI_ACTIVE over "
                                      "tools/step_kernel_roofline.py, a builder-run pass (%s)" % tag)
    json.dump(rec, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:])
    elif sys.argv[1] == "stepkernels":
        step_kernels(*sys.argv[2:])
    else:
        {"stats": stats, "pmc": pmc, "timeline": timeline}[sys.argv[1]](sys.argv[2], sys.argv[3])
