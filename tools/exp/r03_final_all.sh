#!/bin/bash
# end-of-round evidence in one GPU call: profile (stamps pmc_traffic.json with the kernel-source hash), then the bench
# lines against that stamp, then the wide-net side lines
cd $GRAFT_REPO_ROOT
bash profiles/run_profile.sh r03 > gpurun_out/r03_profile.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
bash tools/exp/r03_final_bench.sh
for h in "100,50,25" "128,128"; do
  tag=$(echo $h | tr ',' '_')
  python bench.py --hidden $h --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_swimmer4096_hidden_$tag.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_bench_swimmer4096_hidden_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phase_ms"].items()}, d.get("roofline_mfma",{}).get("frac"), d.get("roofline_mfma",{}).get("avg_launch_ms"))
PY
