#!/usr/bin/env python3
"""profiles/make_index.py -- regenerate profiles/INDEX.md: every tracked file under profiles/ with the commit that last
wrote it and the command that produces it.  Run from the repo root after committing new evidence:
    python profiles/make_index.py && git add profiles/INDEX.md
The command column comes from the table below (first matching pattern wins); a file no pattern matches is listed as
"(no recipe recorded)" so that it shows up in review."""
import fnmatch
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# pattern -> (what it is, command that writes it on the GPU box)
RECIPES = [
    ("pmc_traffic.json", "HBM bytes + vector-instruction counters of the rollout kernels per workload, stamped with sha256(csrc/*); bench.py reads it",
     "profiles/run_profile.sh <tag> (traffic step: profiles/summarize.py traffic ...)"),
    ("*_bench_line.json", "the default bench line of that tree", "python bench.py"),
    ("*_bench_under_rocprof.log", "bench stdout under the kernel-trace pass", "profiles/run_profile.sh <tag>"),
    ("*_c5_kernel_stats.csv", "kernel stats of C5's per-GPU shard", "profiles/run_profile.sh <tag> cheetah1024_trpo_gae"),
    ("*_c5_pmc_*.csv", "PMC passes of C5's per-GPU shard (one counter group per pass)", "profiles/run_profile.sh <tag> cheetah1024_trpo_gae"),
    ("*_c2_*.csv", "kernel stats / PMC passes of C2 (Cartpole 4096 envs, VPG)", "profiles/run_profile.sh <tag> cartpole4096_vpg"),
    ("r06_csplit_*.csv", "SQ counter passes / kernel stats of csplit_fvp_kernel on (100, 50, 25) and (128, 128) at 2.048 M samples", "tools/exp/r06_call8.sh"),
    ("r06_c7_csplit_ab.txt", "cooperative product per library: round start / one instruction stream per wavefront (off) / compile-time shapes + k-slices", "tools/exp/r06_call7.sh"),
    ("r06_c9_csplit_ab.txt", "cooperative product per library: k-slices + compile-time shapes / round start / + late fetch", "tools/exp/r06_call9.sh"),
    ("r06_c10_csplit_ab.txt", "cooperative product: the library of the round's start, timed in the call that ran the parity tests and the wide bench lines on the final one", "tools/exp/r06_call10.sh"),
    ("r06_f16_split.txt", "device facts of the two-way f16 split: subnormal inputs honoured, accuracy of five product forms against float64 over ten magnitude regimes, the five-instruction split bit for bit, instruction rates", "tools/ubench/f16_split.hip (tools/exp/r06_call15.sh)"),
    ("r06_c14_half*.txt", "what a two-part / three-term product would buy: RL_ABL_HALF timing ablation of the bf16 split kernels (wrong results)", "tools/exp/r06_call14.sh"),
    ("r06_c16_splith_ab*.txt", "fvp() per library: bf16 three-way / f16 two-way / ablations and variants of policy_splith_kernels.hip", "tools/exp/r06_call16.sh"),
    ("r06_c18_splith_check.txt", "two-way f16 / three-way bf16 / f32 products against float64 over shapes, direction and observation scales, with timings", "tools/exp/fvp_splith_check.py (tools/exp/r06_call18.sh)"),
    ("r06_c17_bench_*.json", "headline and C5 bench lines, f16 two-way (default) and bf16 three-way (RLLAB_FVP_SPLIT=5) interleaved in one call", "tools/exp/r06_call17.sh"),
    ("r06_c20_splith_ab_wide.txt", "fvp() of (20, 6) on (32, 32), 512 k samples: bf16 three-way against f16 two-way at one wavefront per SIMD", "tools/exp/r06_call20.sh"),
    ("r06_splith_*.csv", "SQ counter passes / FETCH_SIZE / kernel stats of the bf16 and the f16 split products, (32, 32) at 2.048 M and (64, 64) at 512 k samples", "tools/exp/r06_call19.sh"),
    ("curves/r06_splith_*", "learning under the f16 two-way and the bf16 three-way split products, same seeds; 1500-iteration soak", "tools/exp/r06_splith_curves.sh"),
    ("r06_c11_policy_time.txt", "wide loss / gradient / product passes with (.orig) and without (lib_before_wpf) the one-tile-ahead prefetch", "tools/exp/r06_call11.sh"),
    ("r06_wide_kernel_bench.txt", "loss / gradient / product passes of the wide nets at the final sources", "python tools/kernel_bench.py --configs ..."),
    ("*_split_kernel_stats.csv", "kernel stats of the three Fisher-vector-product kernels back to back", "tools/prof_split.sh"),
    ("*_split_pmc_*.csv", "PMC passes of the product kernels", "tools/prof_split.sh"),
    ("*_split64_*.csv", "kernel stats / PMC passes of the 64-unit split product", "tools/prof_split.sh 64"),
    ("*_wide*_pmc_*.csv", "PMC passes of the wide / deep kernels", "tools/prof_wide.sh"),
    ("*_wide*_stats.csv", "kernel stats of the wide / deep kernels", "tools/prof_wide.sh"),
    ("*_fvp_fixed_cost_stats.csv", "50 cached products back to back per batch size", "python tools/exp/fvp_fixed_cost.py under rocprofv3 --kernel-trace --stats"),
    ("*_step_kernel_roofline.jsonl", "rl_vecenv_step at 4 M envs, HIP-event timing", "python tools/step_kernel_roofline.py"),
    ("*_step_kernel_stats.csv", "the same under rocprofv3 --kernel-trace --stats", "profiles/run_profile.sh <tag>"),
    ("*_timeline.csv", "one iteration's dispatch timeline (start, duration, idle gap)", "profiles/run_profile.sh <tag> (summarize.py timeline)"),
    ("*_kernel_stats.csv", "rocprofv3 --kernel-trace --stats summary of `bench.py --steps 5 --warmup 2`", "profiles/run_profile.sh <tag>"),
    ("*_cheetah_pmc_*.csv", "PMC passes of the HalfCheetah shard", "profiles/run_profile.sh <tag> (cheetah leg)"),
    ("*_pmc_fetch_size.csv", "FETCH_SIZE pass (own run)", "profiles/run_profile.sh <tag>"),
    ("*_pmc_write_size.csv", "WRITE_SIZE pass (own run)", "profiles/run_profile.sh <tag>"),
    ("*_pmc_sq*.csv", "SQ_* counter passes (own runs)", "profiles/run_profile.sh <tag>"),
    ("*_bench_*.json", "side bench line: workload / size / switch in the file name", "python bench.py --workload ... | --n-envs ... | --hidden ... (tools/exp/r0N_final_bench.sh)"),
    ("*_resource_usage.txt", "-Rpass-analysis=kernel-resource-usage lines (registers, spills, LDS) of the named kernels", "tools/resource_usage.sh"),
    ("*_preflight*.json", "multi-GPU pre-flight record", "python tools/preflight_multigpu.py"),
    ("*_notes.md", "the round's measurements, experiments and dead ends in prose", "(written by hand from the files of that round)"),
    ("*_fallback_probe_*.txt", "which constructor options stay on the fused path, and the cost of an iteration (512 envs x 100 steps)",
     "python tools/exp/fallback_probe.py / tools/exp/fallback_probe_envs.py"),
    ("curves/r05_*", "tabular training logs of round 5's learning-curve runs", "tools/exp/r05_curves.sh, tools/exp/r05_call13.sh"),
    ("curves/*", "tabular training logs of the learning-curve runs", "tools/learning_curves.sh, tools/exp/r03_*curves*.sh"),
    ("run_profile.sh", "the profile recipe", "-"),
    ("summarize.py", "condenses rocprofv3 output directories into the CSVs kept here", "-"),
    ("make_index.py", "writes this index", "-"),
    ("INDEX.md", "this file", "python profiles/make_index.py"),
]

ROUND_SETS = """\
One evidence set per round (older intermediate sets `r01a … r01l` were pruned in round 4; `git log -- profiles/` has them):

| round | set | kernel-source state |
|---|---|---|
| 1 | `r01_kernel_stats.csv` (first build, VALU design) and `r01m_*` (end of round 1), `r01n_bench_line.json` | the commits in the table below |
| 2 | `r02_*` | end of round 2 |
| 3 | `r03_*` (headline), `r03_split_*` (product kernels), `r03_wide_*` | HEAD of round 3 = `8bd729a`; `pmc_traffic.json` carries the sha256 of `rllab_amd/csrc/*` it was taken at |
| 4 | `r04_*` | see the stamp in `pmc_traffic.json` and `r04_notes.md` |
| 5 | `r05_*` (headline, `r05_c5_*` C5's shard, `r05_split_*` product kernels) | one run of `tools/exp/r05_final_a.sh` at the final kernel sources (stamp in `pmc_traffic.json`); `r05_notes.md` |
| 6 | `r06_*` (headline, `r06_c5_*`, `r06_c2_*`, `r06_split16_*`, `r06_csplit_*`) | one run of `tools/exp/r06_final2_all.sh` at the final kernel sources (stamp in `pmc_traffic.json`); `r06_notes.md` |
"""


def main():
    files = subprocess.check_output(["git", "ls-files", "profiles"], cwd=ROOT, text=True).split()
    rows = []
    for f in sorted(files):
        rel = f[len("profiles/"):]
        log = subprocess.check_output(["git", "log", "-1", "--format=%h %ad", "--date=short", "--", f], cwd=ROOT, text=True).strip()
        what, cmd = "(no recipe recorded)", ""
        for pat, w, c in RECIPES:
            if fnmatch.fnmatch(rel, pat):
                what, cmd = w, c
                break
        rows.append((rel, log or "(uncommitted)", what, cmd))
    with open(os.path.join(ROOT, "profiles", "INDEX.md"), "w") as fh:
        fh.write("# profiles/ index (generated by `python profiles/make_index.py`)\n\n")
        fh.write(ROUND_SETS + "\n")
        fh.write("| file | last written by (commit, date) | what | command |\n|---|---|---|---|\n")
        for r in rows:
            fh.write("| `%s` | %s | %s | `%s` |\n" % r)
    print("profiles/INDEX.md: %d files" % len(rows))


if __name__ == "__main__":
    main()
