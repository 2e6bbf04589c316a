#!/bin/bash
# round 6, call 19: SQ counters and HBM bytes of the two-way f16 split products next to the three-way bf16 ones
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; P=/tmp/prof_r06h; rm -rf $P; mkdir -p $P
CMD="python tools/exp/fvp_splith_pmc.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $P/sq -- $CMD > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq $O/r06_splith_pmc_sq.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/sq2 -- $CMD > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq2 $O/r06_splith_pmc_sq2.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $CMD > /dev/null 2>&1
python profiles/summarize.py pmc $P/fetch $O/r06_splith_pmc_fetch_size.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $CMD > /dev/null 2>&1
python profiles/summarize.py stats $P/stats $O/r06_splith_kernel_stats.csv
grep -h "fvp_split\|^kernel" $O/r06_splith_pmc_sq.csv $O/r06_splith_pmc_sq2.csv $O/r06_splith_pmc_fetch_size.csv $O/r06_splith_kernel_stats.csv | cut -c1-400
