"""Prints, for every kernel in a hipcc -S listing, the instruction pattern of its MFMA-heaviest basic block:
M = MFMA, r = ds_read_b128, t = ds_read_b64_tr_b16, e = v_exp, . = other VALU, s = SALU, |n = s_waitcnt lgkmcnt(n), |v = vmcnt wait,
B = s_barrier, D = LDS-DMA.  Usage: python tools/isa_pattern.py file.s [kernel-substring]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
filt = sys.argv[2] if len(sys.argv) > 2 else ''
ends = [i for i, l in enumerate(s) if 's_endpgm' in l]
prev = 0
for e in ends:
    body = s[prev:e]
    prev = e + 1
    name = next((l for l in body if re.match(r'^_Z\w+:', l)), '?')
    if filt not in name:
        continue
    blocks, cur = [], []
    for l in body:
        if re.match(r'^\.LBB', l):
            blocks.append(cur); cur = [l]
        else:
            cur.append(l)
            if 's_cbranch' in l or 's_branch' in l:
                blocks.append(cur); cur = []
    blocks.append(cur)
    big = max(blocks, key=lambda b: sum('v_mfma' in x for x in b))
    seq = []
    for l in big:
        t = l.split()
        if not t:
            continue
        op = t[0]
        if op.startswith('v_mfma'): seq.append('M')
        elif op.startswith('ds_read_b128'): seq.append('r')
        elif op.startswith('ds_read_b64_tr'): seq.append('t')
        elif op.startswith('ds_'): seq.append('d')
        elif op.startswith('s_waitcnt'):
            m = re.search(r'lgkmcnt\((\d+)\)', l)
            seq.append('|' + (m.group(1) if m else 'v'))
        elif op.startswith('s_barrier'): seq.append('B')
        elif op.startswith('global_load_lds') or (op.startswith('buffer_load') and ' lds' in l): seq.append('D')
        elif op.startswith('global_') or op.startswith('buffer_'): seq.append('G')
        elif op.startswith('v_exp'): seq.append('e')
        elif op.startswith('v_'): seq.append('.')
        elif op.startswith('s_nop'): seq.append('n')
        elif op.startswith('s_'): seq.append('s')
    print(name, 'block lines', len(big), 'mfma', sum('v_mfma' in x for x in big))
    print(''.join(seq))
    print(Counter(l.split()[0] for l in big if l.split() and not l.split()[0].startswith(';')).most_common(14))
