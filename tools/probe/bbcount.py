#!/usr/bin/env python3
"""Development tool: per-basic-block instruction-class counts of an AMDGPU assembly listing (hipcc -S --cuda-device-only).
    python tools/probe/bbcount.py file.s [min_total]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
blocks = []
cur = ['entry', {}]


def add(k):
    cur[1][k] = cur[1].get(k, 0) + 1


for l in lines:
    s = l.strip()
    m = re.match(r'^(\.LBB\S+):', s)
    if m:
        blocks.append(cur)
        cur = [m.group(1), {}]
        continue
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    op = s.split()[0]
    if op.startswith('v_mfma'):
        add('mfma')
    elif 'dpp' in s and op.startswith('v_'):
        add('dpp')
    elif op.startswith('v_'):
        add('valu')
    elif op.startswith('ds_') or op.startswith('global_') or op.startswith('buffer_'):
        add(op)
    elif op == 's_barrier':
        add('barrier')
    elif op == 's_waitcnt':
        add('waitcnt')
    elif op.startswith('s_cbranch') or op == 's_branch':
        add('branch')
    elif op.startswith('s_'):
        add('salu')
blocks.append(cur)
for n, c in blocks:
    tot = sum(c.values())
    if tot >= mn:
        print(n, tot, c)
