"""Instruction mix of a kernel's ISA between consecutive s_barrier instructions (code-layout order):
    hipcc ... -S --cuda-device-only -o k.s file.hip ; python tools/isa_segments.py k.s <mangled-name-substring>"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
start = None
for mm in re.finditer(r'^([A-Za-z_]\S*):', txt, re.M):
    if pat in mm.group(1):
        start = mm
        break
end = txt.index('s_endpgm', start.end())
print(start.group(1))
keys = ['n', 'scr_st', 'scr_ld', 'mfma', 'ds_r', 'ds_w', 'vmem_ld', 'vmem_st', 'valu', 'salu', 'waitcnt']
cur = dict.fromkeys(keys, 0)
seg = []
for l in txt[start.end():end].split('\n'):
    t = l.strip()
    if not t or t[0] in ';.' or t.endswith(':'):
        continue
    cur['n'] += 1
    if 'scratch_store' in t: cur['scr_st'] += 1
    elif 'scratch_load' in t: cur['scr_ld'] += 1
    elif 'v_mfma' in t: cur['mfma'] += 1
    elif t.startswith(('ds_read', 'ds_load')): cur['ds_r'] += 1
    elif t.startswith(('ds_write', 'ds_store')): cur['ds_w'] += 1
    elif t.startswith(('buffer_load', 'global_load')): cur['vmem_ld'] += 1
    elif t.startswith(('buffer_store', 'global_store')): cur['vmem_st'] += 1
    elif t.startswith('s_waitcnt'): cur['waitcnt'] += 1
    elif t.startswith('v_'): cur['valu'] += 1
    elif t.startswith('s_'): cur['salu'] += 1
    if t.startswith('s_barrier'):
        seg.append(cur)
        cur = dict.fromkeys(keys, 0)
seg.append(cur)
for i, c in enumerate(seg):
    print(i, ' '.join(f"{k}={v}" for k, v in c.items() if v))
