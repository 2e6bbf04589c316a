#!/usr/bin/env python3
"""tools/isa_attribution.py <file.hip> <kernel-name-substring> [--flags ...]  -- static instruction profile of the
largest loop of a kernel by SOURCE FILE (and top source lines): compiles for gfx950 with -gline-tables-only, walks the
.loc directives of the assembly.  What the PMC counters cannot say: which part of a fused kernel (policy / physics /
RNG / record) the issue slots of one env-step belong to.  Inner loops (reset paths, Philox rounds) are listed apart:
they are inside the loop's text but not on every iteration's path.

    python tools/isa_attribution.py rllab_amd/csrc/env_kernels.hip rollout_two_leg_quad_kernelINS_11HalfCheetahELi64 \\
        -mllvm -amdgpu-sched-strategy=max-ilp
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cost(op, line):
    if op.startswith("v_mfma"):
        return 32
    if op in ("v_rcp_f32", "v_exp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"):
        return 8
    if op.startswith("v_pk_"):
        return 5
    if op == "s_nop":
        return 4 * (int(line.split()[1]) + 1)
    return 4


def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    out = os.path.join(tempfile.gettempdir(), "isa_attr_%s.s" % os.path.basename(src))
    if not (os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on",
                               "-fno-slp-vectorize", "-gline-tables-only", "--cuda-device-only", "-S", "-o", out, src] + extra)
    lines = open(out).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    st = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l))
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = lines[st:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    loops.sort(key=lambda x: x[1] - x[0], reverse=True)
    lo, hi = loops[0]
    inner = [(a, b) for a, b in loops[1:] if a >= lo and b <= hi]
    cur = ("?", 0)
    by_file, cyc, by_line = collections.Counter(), collections.Counter(), collections.Counter()
    for i in range(lo, hi + 1):
        l = body[i]
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not l.startswith("\t") or l.strip().startswith(";") or l.strip().startswith("."):
            continue
        if any(a <= i <= b for a, b in inner):
            by_file["(inner loops: reset / Philox rounds)"] += 1
            continue
        op = l.split()[0]
        by_file[cur[0]] += 1
        cyc[cur[0]] += cost(op, l)
        by_line["%s:%d" % cur] += 1
    rec = {"kernel": lines[st].split(":")[0], "loop_instructions": sum(by_file.values()),
           "by_source_file": {f: {"instructions": n, "issue_cycles_lone_wavefront": cyc[f]} for f, n in by_file.most_common()},
           "top_source_lines": dict(by_line.most_common(25)),
           "note": "static: both sides of wave-uniform branches are counted; 4 cycles per issue slot, packed f32 5, "
                   "transcendental 8, matrix instruction 32, s_nop N = N + 1 slots (tools/ubench/valu_latency.hip)"}
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
