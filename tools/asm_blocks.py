#!/usr/bin/env python3
"""Developer tool: split a kernel's gfx950 assembly (hipcc -S --cuda-device-only) into basic blocks and
count instruction classes per block.  usage: asm_blocks.py file.s kernel_symbol_substring"""
import re, sys
src, sym = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sym in l and l.rstrip().endswith(':') or (l.startswith('_Z') and sym in l and ':' in l))
blocks = []; cur = {'name': 'entry', 'line': start, 'v': 0, 's': 0, 'lds': 0, 'vm': 0, 'sm': 0, 'br': [], 'wait': 0}
for i in range(start + 1, len(lines)):
    l = lines[i].strip()
    if l.startswith('.Lfunc_end'): break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = {'name': m.group(1), 'line': i, 'v': 0, 's': 0, 'lds': 0, 'vm': 0, 'sm': 0, 'br': [], 'wait': 0}
        continue
    if l.startswith('; %bb.'):
        blocks.append(cur); cur = {'name': l[2:].split(':')[0], 'line': i, 'v': 0, 's': 0, 'lds': 0, 'vm': 0, 'sm': 0, 'br': [], 'wait': 0}
        continue
    if not l or l.startswith(';') or l.startswith('.'): continue
    op = l.split()[0]
    if op.startswith('v_'): cur['v'] += 1
    elif op.startswith('ds_'): cur['lds'] += 1
    elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): cur['vm'] += 1
    elif op.startswith('s_load') or op.startswith('s_memrealtime') or op.startswith('s_memtime'): cur['sm'] += 1
    elif op.startswith('s_waitcnt'): cur['wait'] += 1
    elif op.startswith('s_cbranch') or op == 's_branch':
        cur['s'] += 1; cur['br'].append(op.replace('s_cbranch_', '') + '->' + l.split()[1])
    elif op.startswith('s_'): cur['s'] += 1
blocks.append(cur)
for b in blocks:
    print(f"{b['line']+1:6d} {b['name']:12s} V={b['v']:3d} S={b['s']:3d} LDS={b['lds']:3d} VM={b['vm']:2d} SM={b['sm']:2d} W={b['wait']:2d}  {' '.join(b['br'])}")
