"""Instruction-class census of every kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
python tools/isa_count.py file.s [name-filter]   -- counts are static (whole kernel), loops are not weighted."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z[\w]+):[^\n]*\n(.*?)^\s+s_endpgm', s, re.M | re.S):
    name, body = m.group(1), m.group(2)
    if flt not in name: continue
    c = Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'): continue
        i = l.split()[0]
        if i.startswith('v_mfma'): c['mfma'] += 1
        elif i.startswith('v_'): c['valu'] += 1; c[i] += 0
        elif i.startswith('s_waitcnt'): c['wait'] += 1
        elif i.startswith('s_nop'): c['nop'] += 1
        elif i.startswith('s_'): c['salu'] += 1
        elif i.startswith('ds_'): c['lds'] += 1
        elif i.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): c['vmem'] += 1
    vg = re.search(r'\.vgpr_count:\s+(\d+)', s[m.end():m.end() + 20000])
    print(name[:80], {k: v for k, v in c.items() if v})
