#!/bin/bash
# Round-1 profile recipe (run on the GPU box through gpurun):
#   kernel-trace/stats pass and SEPARATE --pmc passes of the same bench command, condensed
#   into gpurun_out/r01_*.csv by profiles/summarize.py (raw traces are too large to keep).
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=/tmp/prof_r01
rm -rf $P && mkdir -p $P gpurun_out
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $BENCH > gpurun_out/r01_bench_under_rocprof.log 2>&1
python profiles/summarize.py stats $P/stats gpurun_out/r01_kernel_stats.csv
head -12 gpurun_out/r01_kernel_stats.csv
BENCH2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/fetch gpurun_out/r01_pmc_fetch_size.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/write gpurun_out/r01_pmc_write_size.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $P/sq -- $BENCH2 > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq gpurun_out/r01_pmc_sq.csv
head -5 gpurun_out/r01_pmc_fetch_size.csv gpurun_out/r01_pmc_write_size.csv gpurun_out/r01_pmc_sq.csv
tail -2 gpurun_out/r01_bench_under_rocprof.log
ls -la gpurun_out
