#!/usr/bin/env python3
"""One character per instruction for every basic block of a kernel that holds MFMAs (M mfma, v valu, d ds_read, w ds_write,
L lds-dma, g global/buffer load, s store, W s_waitcnt, B barrier, . scalar/other):  isa_trace.py FILE.s MANGLED_SUBSTRING"""
import re
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.splitlines() if ':' in l and key in l.split(':')[0] and not l.startswith(('.', ' ', '\t', ';'))]
for name in names:
    a = s.index('\n' + name + ':')
    b = s.index('.Lfunc_end', a)
    cur, seq, blocks = 'entry', [], []
    for l in s[a:b].splitlines():
        t = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            blocks.append((cur, ''.join(seq)))
            cur, seq = m.group(1), []
            continue
        if not t or t.startswith((';', '.')):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'): c = 'M'
        elif op.startswith('ds_read'): c = 'd'
        elif op.startswith('ds_write'): c = 'w'
        elif 'lds' in t and op.startswith('buffer_load'): c = 'L'
        elif op.startswith(('global_load', 'buffer_load')): c = 'g'
        elif op.startswith(('global_store', 'buffer_store')): c = 's'
        elif op == 's_waitcnt': c = 'W'
        elif op == 's_barrier': c = 'B'
        elif op.startswith('v_'): c = 'v'
        else: c = '.'
        seq.append(c)
    blocks.append((cur, ''.join(seq)))
    print(name)
    for lab, q in blocks:
        if 'M' in q:
            print(f'  {lab} ({len(q)}): {q}')
