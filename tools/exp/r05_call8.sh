#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3.json
python - <<PY
import json
d=json.loads(open("$O/bench_c3.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), d["phase_ms"])
PY
