"""Instruction-class string of a kernel's large basic blocks (M mfma, v valu, p permlane, d LDS, L buffer load / LDS-DMA, S store,
w waitcnt, B barrier, n nop, s scalar): how the compiler interleaved an epilogue with the matrix instructions.
usage: python tools/asm_pattern.py build/api.s <mangled kernel name> [min block size]"""
import sys
src, name = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
blocks, cur = [], ['entry', []]
blocks.append(cur)
for l in lines[start + 1:]:
    if 's_endpgm' in l:
        break
    if l.startswith('.LBB'):
        cur = [l.split(':')[0], []]
        blocks.append(cur)
    elif l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        cur[1].append(l.strip().split()[0])
def cls(op):
    for p, c in (('v_mfma', 'M'), ('ds_', 'd'), ('buffer_load', 'L'), ('buffer_store', 'S'), ('v_permlane', 'p'), ('v_accvgpr', 'a'), ('v_', 'v'),
                 ('s_waitcnt', 'w'), ('s_barrier', 'B'), ('s_nop', 'n')):
        if op.startswith(p):
            return c
    return 's'
for nm, ops in blocks:
    if len(ops) >= minlen:
        s = ''.join(cls(o) for o in ops)
        print(nm, len(ops), 'instructions:', {c: s.count(c) for c in sorted(set(s))})
        for i in range(0, len(s), 160):
            print('  ' + s[i:i + 160])
