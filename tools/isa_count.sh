#!/bin/bash
# tools/isa_count.sh <file.hip> <mangled-kernel-substring> : compile for gfx950, print instruction
# histogram of every loop body (between a .LBB label and the backward branch to it) in that kernel.
set -e
SRC=$(realpath $1); PAT=$2; OUT=${3:-/tmp/isa_count}
mkdir -p $OUT && cd $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fno-slp-vectorize --cuda-device-only -S -o k.s $SRC
python3 - "$PAT" <<'PY'
import re, sys
pat = sys.argv[1]
lines = open('k.s').read().split('\n')
# find function
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*%s\S*:' % re.escape(pat), l)]
for st in start:
    name = lines[st].split(':')[0]
    end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
    body = lines[st:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    print(name, 'lines', len(body))
    for i, l in enumerate(body):
        m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = [x.split()[0] for x in body[labels[m.group(1)]:i + 1]
                   if x.startswith('\t') and not x.strip().startswith(';') and not x.strip().startswith('.')]
            v = sum(1 for x in seg if x.startswith('v_'))
            s = sum(1 for x in seg if x.startswith('s_'))
            print('  loop %s: %d instrs (%d vector, %d scalar, %d mfma, %d nop, %d dpp-lines)' % (
                m.group(1), len(seg), v, s, sum('mfma' in x for x in seg), sum(x == 's_nop' for x in seg),
                sum('quad_perm' in x or 'row_' in x for x in body[labels[m.group(1)]:i + 1])))
PY
