#!/bin/bash
# round 5: learning curves of what changed this round (GPU box) -> gpurun_out/curves/r05_*.csv + r05_summary.txt
#   the device-decided line search against the host-decided one (same seeds), the split64 Fisher-vector product against the
#   f32 kernel on C5's shapes, one-hidden-layer / rectify policies, running normalisation and position_only on the fused path
mkdir -p gpurun_out/curves
S=gpurun_out/curves/r05_summary.txt
: > $S
run() {   # tag, env-prefix..., -- args
  tag=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done
  shift
  t0=$(date +%s.%N)
  env $envs timeout 300 python examples/run_trpo.py "$@" --quiet --csv gpurun_out/curves/r05_$tag.csv 2>&1 | grep -v amdgpu.ids | tail -1
  t1=$(date +%s.%N)
  python - <<PY >> $S
import csv
r=list(csv.DictReader(open("gpurun_out/curves/r05_$tag.csv")))
bt=[float(x.get("backtrack_iters", x.get("BacktrackItr", 0)) or 0) for x in r] if r else []
print("%-34s iters %3d wall %5.1f s  AverageReturn first / mean of last 5: %9.3f / %9.3f   max MeanKL %.5f" % ("$tag", len(r), $t1 - $t0, float(r[0]["AverageReturn"]), sum(float(x["AverageReturn"]) for x in r[-5:])/5, max(float(x["MeanKL"]) for x in r)))
PY
}
run swimmer_device_ls -- --env swimmer --n-envs 4096 --n-itr 100
run swimmer_host_ls RLLAB_DEVICE_LINE_SEARCH=0 -- --env swimmer --n-envs 4096 --n-itr 100
run cheetah_split64 -- --env half_cheetah --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97
run cheetah_f32_fvp RLLAB_FVP_SPLIT=0 -- --env half_cheetah --n-envs 1024 --n-itr 100 --hidden 64 --gae-lambda 0.97
run swimmer_one_layer_32 -- --env swimmer --n-envs 4096 --n-itr 60 --hidden 32 --one-hidden-layer
run swimmer_relu_32_32 -- --env swimmer --n-envs 4096 --n-itr 60 --nonlinearity relu
run swimmer_normalize_obs -- --env swimmer --n-envs 4096 --n-itr 60 --normalize-obs
run cheetah_normalize_obs_reward -- --env half_cheetah --n-envs 1024 --n-itr 60 --hidden 64 --gae-lambda 0.97 --normalize-obs --normalize-reward
run cartpole_position_only -- --env cartpole --n-envs 1024 --n-itr 40 --position-only
cat $S
