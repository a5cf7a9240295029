#!/usr/bin/env python3
"""Instruction histogram and the vmcnt waits of one kernel in a -save-temps .s file:  isa_stats.py FILE.s MANGLED_SUBSTRING"""
import collections
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.splitlines() if ':' in l and key in l.split(':')[0] and not l.startswith(('.', ' ', '\t', ';'))]
for name in names:
    a = s.index('\n' + name + ':')
    b = s.index('.Lfunc_end', a)
    body = s[a:b].splitlines()
    c = collections.Counter()
    for l in body:
        t = l.strip().split()
        if t and not t[0].startswith(('.', ';')) and not t[0].endswith(':'):
            c[t[0]] += 1
    print(name, len(body), 'lines')
    print('  ' + ', '.join(f'{k} {v}' for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30)))
    print('  vmcnt waits:', collections.Counter(l.strip() for l in body if 's_waitcnt' in l and 'vmcnt' in l))
