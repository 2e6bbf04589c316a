#!/bin/bash
# the headline rollout's workgroup shape at HEAD: 256 wavefronts as workgroups of 1 / 2 / 4 (RLLAB_ROLLOUT_WPB), same box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for w in 0 1 2 4; do
  if [ $w = 0 ]; then e=""; else e="RLLAB_ROLLOUT_WPB=$w"; fi
  env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('WPB $w', round(d['ms_per_step'],3), round(d['phase_ms']['sample'],3))"
done
done
