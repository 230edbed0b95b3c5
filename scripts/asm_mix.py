#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing.
usage: asm_mix.py file.s <mangled-name-substring> [min_block_size]"""
import collections, re, sys
src = open(sys.argv[1]).read().split('\n')
needle = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = next(i for i, l in enumerate(src) if l.startswith('_Z') and needle in l and l.rstrip().split(':')[0].endswith('E') or (l.startswith('_Z') and needle in l and ':' in l))
end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
lines = src[start:end]
def cls(k):
    if k.startswith('v_pk_'): return 'v_pk'
    if k.startswith('v_mfma'): return 'mfma'
    if k.startswith('v_'): return 'valu'
    if k.startswith('s_waitcnt'): return 'waitcnt'
    if k.startswith('s_nop'): return 'nop'
    if k.startswith('s_cbranch') or k.startswith('s_branch'): return 'branch'
    if k.startswith('s_'): return 'salu'
    if k.startswith('ds_'): return 'lds'
    if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    return 'other'
pos = [i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)] + [len(lines)]
tot = collections.Counter()
for a, b in zip(pos[:-1], pos[1:]):
    ins = [l.strip().split()[0] for l in lines[a:b] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    g = collections.Counter(cls(k) for k in ins)
    if len(ins) >= minsz:
        print(lines[a].split(';')[0].strip(), lines[a].split(';')[1].strip() if ';' in lines[a] else '', len(ins), dict(g))
        if '-v' in sys.argv: print('   ', collections.Counter(ins).most_common(25))
for l in lines[-40:]:
    pass
meta = [l for l in src[end:end+80] if 'vgpr_count' in l or 'sgpr_count' in l or 'private_segment' in l]
print(len(lines), 'lines;', len(pos) - 1, 'blocks')
