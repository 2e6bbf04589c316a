cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -5
for v in "" "RLLAB_PEER_ALLREDUCE=1"; do
  env RLLAB_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['phase_ms'], d['collectives_per_iter'], d['peer_reductions_per_iter'], d['collective_ms_per_iter'])"
done
