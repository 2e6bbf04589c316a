#!/bin/bash
# tools/prof_pmc.sh <tag> <kernel-name-substring> <command ...>: kernel stats + SQ / LDS / HBM counter passes (one counter
# group per pass, rocprofv3 --pmc) of any command, condensed to gpurun_out/<tag>_{kernel_stats,pmc_sq,pmc_sq2,pmc_sq3,
# pmc_fetch_size,pmc_write_size}.csv; prints the rows of the named kernel.  GPU box, through gpurun.
TAG=$1; KERN=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- "$@" > /dev/null 2>&1
python profiles/summarize.py stats $P/stats gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $P/sq -- "$@" > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq gpurun_out/${TAG}_pmc_sq.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $P/sq2 -- "$@" > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq2 gpurun_out/${TAG}_pmc_sq2.csv
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $P/sq3 -- "$@" > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq3 gpurun_out/${TAG}_pmc_sq3.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -- "$@" > /dev/null 2>&1
python profiles/summarize.py pmc $P/fetch gpurun_out/${TAG}_pmc_fetch_size.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -- "$@" > /dev/null 2>&1
python profiles/summarize.py pmc $P/write gpurun_out/${TAG}_pmc_write_size.csv
grep -h "$KERN\|^kernel" gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_pmc_sq.csv gpurun_out/${TAG}_pmc_sq2.csv gpurun_out/${TAG}_pmc_sq3.csv gpurun_out/${TAG}_pmc_fetch_size.csv gpurun_out/${TAG}_pmc_write_size.csv | cut -c1-500
