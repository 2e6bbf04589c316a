#!/bin/bash
# round 6, call 8: SQ counters of csplit_fvp_kernel (k-slices + compile-time shapes) on (100, 50, 25) and (128, 128)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; P=/tmp/prof_r06c; rm -rf $P; mkdir -p $P
CMD="python tools/exp/csplit_pmc.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $P/sq -- $CMD > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq $O/r06_csplit_pmc_sq.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/sq2 -- $CMD > /dev/null 2>&1
python profiles/summarize.py pmc $P/sq2 $O/r06_csplit_pmc_sq2.csv
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $P/sq3 -- $CMD > $O/r06_csplit_pmc_sq3.log 2>&1
python profiles/summarize.py pmc $P/sq3 $O/r06_csplit_pmc_sq3.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- $CMD > /dev/null 2>&1
python profiles/summarize.py stats $P/stats $O/r06_csplit_kernel_stats.csv
grep -h "csplit_fvp\|^kernel" $O/r06_csplit_pmc_sq.csv $O/r06_csplit_pmc_sq2.csv $O/r06_csplit_pmc_sq3.csv $O/r06_csplit_kernel_stats.csv | cut -c1-500
tail -3 $O/r06_csplit_pmc_sq3.log
