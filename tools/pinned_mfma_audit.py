"""Inline-assembly MFMAs are invisible to hipcc's hazard tables (cdna_hip_programming.md 5.7 item 2).  This scan of the device assembly
checks, for every v_mfma_scale inside an ASMSTART/ASMEND pair, that (a) none of the two instructions in front of it is a vector
instruction writing one of its VGPR operands (two wait states), (b) where its C operand is not its destination, no vector instruction
writes C within the seven states behind it (the MFMA is still reading it), and (c) the kernel has no scratch.
usage: python tools/pinned_mfma_audit.py [file.s] [kernel-name-regex]   (without a file: compiles csrc/api.hip to assembly first)
Exit code 1 on a finding (a CPU test runs it on the product build's assembly)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly_file(tmp):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _device_asm import assembly_path          # compiled once into build/, reused while no source is newer
    return assembly_path(False)


def regs(tok):
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r'([va])(\d+)', tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


def audit(path, pat):
    lines = open(path).read().split('\n')
    findings, kernels = [], 0
    i = 0
    while i < len(lines):
        m = re.match(r'^(_Z\w+):', lines[i])
        if not m or not re.search(pat, m.group(1)):
            i += 1
            continue
        name, body, j = m.group(1), [], i + 1
        while j < len(lines) and 's_endpgm' not in lines[j]:
            body.append(lines[j])
            j += 1
        i = j
        ins, in_asm = [], False
        for l in body:
            t = l.strip()
            if t.startswith(';;#ASMSTART'):
                in_asm = True
            elif t.startswith(';;#ASMEND'):
                in_asm = False
            elif l.startswith('\t') and t and not t.startswith(';') and not t.startswith('.'):
                ins.append((t, in_asm))
        pinned = [k for k, (t, a) in enumerate(ins) if a and t.startswith('v_mfma')]
        if not pinned:
            continue
        kernels += 1
        if any(t.startswith('scratch_') for t, _ in ins):
            findings.append('%s: scratch instructions' % name)
        for k in pinned:
            ops = [o.strip() for o in ins[k][0].split(None, 1)[1].split(',')]
            vregs = set()
            for o in ops[:6]:
                c, r = regs(o.split()[0])
                if c == 'v':
                    vregs |= r
            dreg, creg = regs(ops[0].split()[0])[1], regs(ops[3].split()[0])[1]
            if creg and creg != dreg:
                fwd, seen = k + 1, 0
                while fwd < len(ins) and seen < 7:
                    t = ins[fwd][0]
                    fwd += 1
                    if t.startswith('s_nop'):
                        seen += 1 + int(t.split()[1])
                        continue
                    seen += 1
                    if t.startswith('v_') and not t.startswith('v_mfma') and not t.startswith('v_cmp'):
                        c, r = regs(t.split(None, 1)[1].split(',')[0].strip())
                        if c == 'v' and r & creg:
                            findings.append('%s: "%s" overwrites the C operand of a pinned MFMA %d state(s) behind it' % (name, t, seen))
            back, seen = k - 1, 0
            while back >= 0 and seen < 2:
                t = ins[back][0]
                back -= 1
                if t.startswith('s_nop'):
                    seen += 1 + int(t.split()[1])
                    continue
                seen += 1
                if t.startswith('v_') and not t.startswith('v_mfma') and not t.startswith('v_cmp'):
                    c, r = regs(t.split(None, 1)[1].split(',')[0].strip())
                    if c == 'v' and r & vregs:
                        findings.append('%s: "%s" writes an operand of the pinned MFMA %d instruction(s) behind it' % (name, t, seen))
    return kernels, findings


if __name__ == '__main__':
    args = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        path = args.pop(0) if args and args[0].endswith('.s') else assembly_file(tmp)
        n, f = audit(path, args[0] if args else '.')
    print('%d kernel(s) with pinned MFMAs, %d finding(s)' % (n, len(f)))
    for x in f:
        print('  ' + x)
    sys.exit(1 if f else 0)
