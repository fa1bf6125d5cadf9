#!/usr/bin/env python3
"""Basic blocks of one kernel in a hipcc -S listing: vector / scalar / LDS / global instruction counts and branch targets.
usage: asm_blocks.py file.s mangled-name-substring [--sum FROM TO]   (sum: VALU over the blocks FROM..TO in listing order)"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(src) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ': ' in l)][0]
end = [i for i in range(start, len(src)) if src[i].strip().startswith('s_endpgm')][0]
blocks = []
cur = {'name': 'entry', 'v': 0, 's': 0, 'ds': 0, 'g': 0, 'br': [], 'bar': 0}
blocks.append(cur)
for ln in src[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        cur = {'name': m.group(1), 'v': 0, 's': 0, 'ds': 0, 'g': 0, 'br': [], 'bar': 0}
        blocks.append(cur)
        continue
    t = ln.strip().split()
    if not t or t[0].startswith(';'):
        continue
    op = t[0]
    if op.startswith('v_'): cur['v'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith('global_'): cur['g'] += 1
    elif op == 's_barrier': cur['bar'] += 1
    elif op.startswith('s_cbranch') or op == 's_branch':
        cur['br'].append(t[1]); cur['s'] += 1
    elif op.startswith('s_'): cur['s'] += 1
if '--sum' in sys.argv:
    i = sys.argv.index('--sum')
    a, b = sys.argv[i + 1], sys.argv[i + 2]
    names = [x['name'].split('_')[-1] for x in blocks]
    ia, ib = names.index(a), names.index(b)
    print('VALU', sum(x['v'] for x in blocks[ia:ib + 1]), 'scalar', sum(x['s'] for x in blocks[ia:ib + 1]), 'LDS', sum(x['ds'] for x in blocks[ia:ib + 1]),
          'global', sum(x['g'] for x in blocks[ia:ib + 1]))
else:
    for b in blocks:
        print(b['name'], 'v', b['v'], 's', b['s'], 'ds', b['ds'], 'g', b['g'], 'bar', b['bar'], '->', ','.join(x.split('_')[-1] for x in b['br']))
