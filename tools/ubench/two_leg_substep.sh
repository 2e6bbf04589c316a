#!/bin/bash
# instruction histogram of the one-body-per-lane sub-step loop (compiles here, no GPU needed)
set -e
cd "$(dirname "$0")"
OUT=${TMPDIR:-/tmp}/two_leg_substep.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
  --cuda-device-only -S -o $OUT two_leg_substep.hip
python3 - $OUT <<'PY'
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
st = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*substep_kernel\S*:', l)][0]
end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
body = lines[st:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
for i, l in enumerate(body):
    m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = [x for x in body[labels[m.group(1)]:i + 1] if x.startswith('\t') and x.strip()[0] not in ';.']
        ops = [x.split()[0] for x in seg]
        print('loop: %d instrs, %d vector, %d scalar, %d nop, %d with a lane move (%d of them bare v_mov_b32_dpp)' % (
            len(ops), sum(o.startswith('v_') for o in ops), sum(o.startswith('s_') for o in ops), ops.count('s_nop'),
            sum('quad_perm' in x or 'row_ror' in x for x in seg), ops.count('v_mov_b32_dpp')))
        print(collections.Counter(ops).most_common(12))
PY
