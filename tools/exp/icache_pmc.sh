#!/bin/bash
# instruction-cache and branch counters of the rollout kernels (C5's and the headline's): gpurun_out/<tag>_pmc_icache.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "r04_c5:--workload cheetah1024_trpo_gae" "r04:"; do
  tag=${spec%%:*}; args=${spec#*:}
  P=/tmp/prof_$tag; rm -rf $P
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d $P/ic -- python $R/bench.py $args --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/profiles/summarize.py pmc $P/ic $R/gpurun_out/${tag}_pmc_icache.csv
  grep -i "rollout" $R/gpurun_out/${tag}_pmc_icache.csv | cut -c1-400
  head -1 $R/gpurun_out/${tag}_pmc_icache.csv
done
