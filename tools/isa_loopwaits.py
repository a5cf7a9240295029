#!/usr/bin/env python3
"""vmcnt waits of a kernel listed per basic block, with the block's loop annotation and what precedes / follows:
   isa_loopwaits.py FILE.s MANGLED_SUBSTRING"""
import re
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.splitlines() if ':' in l and key in l.split(':')[0] and not l.startswith(('.', ' ', '\t', ';'))]
for name in names:
    a = s.index('\n' + name + ':')
    b = s.index('.Lfunc_end', a)
    body = s[a:b].splitlines()
    cur, note = 'entry', ''
    for i, l in enumerate(body):
        t = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', t)
        if m:
            cur, note = m.group(1), (m.group(2) or '')
        elif 's_waitcnt' in t and 'vmcnt' in t:
            inloop = 'Loop' in note
            nxt = next((x.strip() for x in body[i + 1:i + 4] if x.strip() and not x.strip().startswith(';')), '')
            print(f"{cur:12s} {'LOOP' if inloop else '    '} {t:28s} -> {nxt[:60]}")
