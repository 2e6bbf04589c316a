#!/bin/bash
# tools/prof_wide.sh <tag>: kernel stats + SQ counters of the wide-net update passes (GPU box, through gpurun)
TAG=${1:-r03w}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P gpurun_out
CMD="python tools/kernel_bench.py --configs 13,2,128-128,2048000;13,2,100-50-25,2048000"
$CMD > gpurun_out/${TAG}_kernel_bench.jsonl 2> gpurun_out/${TAG}_kernel_bench.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $P/sq -- $CMD > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq gpurun_out/${TAG}_pmc_sq.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/sq2 -- $CMD > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq2 gpurun_out/${TAG}_pmc_sq2.csv
cat gpurun_out/${TAG}_kernel_bench.jsonl; grep wide gpurun_out/${TAG}_pmc_sq.csv gpurun_out/${TAG}_pmc_sq2.csv; head -1 gpurun_out/${TAG}_pmc_sq.csv gpurun_out/${TAG}_pmc_sq2.csv
