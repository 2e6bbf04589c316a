#!/bin/bash
# Profile recipe (run on the GPU box through gpurun):  profiles/run_profile.sh <tag> [workload] [file prefix]
#   kernel-trace/stats pass and SEPARATE --pmc passes of the same bench command, condensed
#   into gpurun_out/<prefix>_*.csv by profiles/summarize.py (raw traces are too large to keep).
#   e.g.  profiles/run_profile.sh r04                                    (headline: swimmer4096_trpo -> r04_*)
#         profiles/run_profile.sh r04 cheetah1024_trpo_gae r04_c5        (C5's per-GPU shard -> r04_c5_*)
#         profiles/run_profile.sh r04 cartpole4096_vpg r04_c2            (C2 -> r04_c2_*)
ROUND=${1:-r04}
WORKLOAD=${2:-swimmer4096_trpo}
TAG=${3:-$ROUND}
case $WORKLOAD in cheetah1024_trpo_gae) NENVS=1024;; *) NENVS=4096;; esac
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=/tmp/prof_$TAG
rm -rf $P && mkdir -p $P gpurun_out
BENCH="python bench.py --workload $WORKLOAD --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $BENCH > gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
python profiles/summarize.py stats $P/stats gpurun_out/${TAG}_kernel_stats.csv
head -12 gpurun_out/${TAG}_kernel_stats.csv
BENCH2="python bench.py --workload $WORKLOAD --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/fetch gpurun_out/${TAG}_pmc_fetch_size.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/write gpurun_out/${TAG}_pmc_write_size.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $P/sq -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq gpurun_out/${TAG}_pmc_sq.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $P/sq2 -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq2 gpurun_out/${TAG}_pmc_sq2.csv
# one pmc_traffic.json for all workloads: start from the tracked copy, replace this workload's record
[ -f gpurun_out/pmc_traffic.json ] || cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python profiles/summarize.py traffic gpurun_out/${TAG}_pmc_fetch_size.csv gpurun_out/${TAG}_pmc_write_size.csv gpurun_out/pmc_traffic.json $WORKLOAD $NENVS ${ROUND} gpurun_out/${TAG}_pmc_sq.csv gpurun_out/${TAG}_pmc_sq2.csv
head -6 gpurun_out/${TAG}_pmc_fetch_size.csv gpurun_out/${TAG}_pmc_write_size.csv gpurun_out/${TAG}_pmc_sq.csv gpurun_out/${TAG}_pmc_sq2.csv
# one iteration's dispatch timeline (start, duration, idle gap before each kernel)
python profiles/summarize.py timeline $P/stats gpurun_out/${TAG}_timeline.csv
if [ "$WORKLOAD" != "swimmer4096_trpo" ]; then ls -la gpurun_out; exit 0; fi
# the per-step VecEnv boundary kernel at a chip-filling size: the kernel the HBM roofline applies to
python tools/step_kernel_roofline.py 2>&1 | grep "^{" > gpurun_out/${TAG}_step_kernel_roofline.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $P/step -- python tools/step_kernel_roofline.py > /dev/null 2>&1
python profiles/summarize.py stats $P/step gpurun_out/${TAG}_step_kernel_stats.csv
# ... and its vector axis: instructions per launch from the SQ counters (bench.py prices roofline_step_kernel with them)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $P/stepsq -- python tools/step_kernel_roofline.py --steps 3 --warmup 1 > /dev/null 2>&1
python profiles/summarize.py pmc $P/stepsq gpurun_out/${TAG}_step_kernel_pmc_sq.csv
python profiles/summarize.py stepkernels gpurun_out/${TAG}_step_kernel_pmc_sq.csv gpurun_out/pmc_traffic.json $((1 << 22)) ${ROUND}
tail -2 gpurun_out/${TAG}_bench_under_rocprof.log
ls -la gpurun_out
