#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phase_ms'].items()})"; }
for pf in 1 0; do
  export RLLAB_UPDATE_PREFETCH=$pf
  echo "== prefetch $pf"
  echo -n "n16384 w2: "; run --n-envs 16384 --steps 5 --warmup 2
  echo -n "n16384 w8: "; run --n-envs 16384 --steps 5 --warmup 8
  echo -n "h128 w3: "; run --hidden 128,128 --steps 10 --warmup 3
  echo -n "c5 w3: "; run --workload cheetah1024_trpo_gae --steps 10 --warmup 3
  echo -n "headline: "; run --steps 20 --warmup 5
done
