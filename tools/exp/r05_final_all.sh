#!/bin/bash
# the round's evidence in one GPU call at the final kernel sources: r05_final_a.sh (GPU suite, headline + C5 profiles, bench
# lines, split-product counters), C2's profile and line against the fresh stamp, r05_final_b.sh (side lines)
cd $GRAFT_REPO_ROOT
bash tools/exp/r05_final_a.sh
bash profiles/run_profile.sh r05 cartpole4096_vpg r05_c2 > gpurun_out/r05_profile_c2.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --workload cartpole4096_vpg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_cartpole4096_vpg.json
bash tools/exp/r05_final_b.sh
