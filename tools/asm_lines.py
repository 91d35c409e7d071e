#!/usr/bin/env python3
"""Developer tool: static instruction counts per SOURCE LINE of a kernel (hipcc -S --cuda-device-only -gline-tables-only).
usage: asm_lines.py file.s kernel_symbol_substring source_file [min_count]"""
import collections, re, sys
asm, sym, srcf = sys.argv[1], sys.argv[2], sys.argv[3]
mn = int(sys.argv[4]) if len(sys.argv) > 4 else 6
lines = open(asm).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sym in l and ':' in l)
cur = None
cv, cs, cl, cm = (collections.Counter() for _ in range(4))
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'): break
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if op.startswith('v_'): cv[cur] += 1
    elif op.startswith('ds_'): cl[cur] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cm[cur] += 1
    elif op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop')): cs[cur] += 1
src = open(srcf).read().split('\n')
keys = sorted(set(cv) | set(cs) | set(cl) | set(cm), key=lambda k: (k[0], k[1]) if k else (0, 0))
print("file line  VALU SALU  LDS VMEM  source")
for k in keys:
    if k is None: continue
    if cv[k] + cs[k] + cl[k] + cm[k] >= mn:
        text = src[k[1] - 1].strip()[:110] if 0 < k[1] <= len(src) else ''
        print(f"{k[0]:3d} {k[1]:5d} {cv[k]:5d} {cs[k]:4d} {cl[k]:4d} {cm[k]:4d}  {text}")
print("total", sum(cv.values()), sum(cs.values()), sum(cl.values()), sum(cm.values()))
