#!/usr/bin/env python3
"""Scan the device assembly of csrc/api.hip for the gfx950 store-data hazard (csrc/gemm_ws.cuh, store_b128_settled): a
buffer_store_dwordx3/x4 whose data registers are overwritten by one of the next two instructions.  LLVM's hazard recogniser does not
protect MUBUF stores whose soffset is an SGPR; on gfx950 such a store was seen sending the overwritten values for lanes 12-15 / 44-47.
usage: tools/store_hazard_scan.py [--variants] [file.s]   (without a file: compiles csrc/api.hip to assembly first; --variants: with
-DCP_VARIANTS, i.e. the tools-only build that also carries the superseded kernels whose timings DESIGN.md quotes).  Exit code 1 = hazards found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly() -> str:
    args = [a for a in sys.argv[1:] if a != "--variants"]
    if args:
        return open(args[0]).read()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _device_asm import assembly_path          # compiled once into build/, reused while no source is newer
    return open(assembly_path("--variants" in sys.argv)).read()


def regs(tok: str):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(text: str):
    hits = []
    kernel = "?"
    lines = [l for l in text.split("\n")]
    code = []                                   # (kernel, instruction text)
    for l in lines:
        t = l.strip()
        if t.endswith(":") and t.startswith("_Z") or (t.startswith("_Z") and ":" in t and not l.startswith("\t")):
            kernel = t.split(":")[0]
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code.append((kernel, t.split(";")[0].strip()))
    for i, (k, ins) in enumerate(code):
        m = re.match(r"buffer_store_dwordx[34]\s+(v\[\d+:\d+\])", ins)
        if not m:
            continue
        data = regs(m.group(1))
        for nxt_k, nxt in code[i + 1:i + 3]:
            if nxt.startswith("s_nop"):
                break                            # wait states behind the store: protected
            if nxt.startswith("v_") and not nxt.startswith("v_cmp"):
                dst = nxt.split()[1].rstrip(",") if len(nxt.split()) > 1 else ""
                written = regs(dst)
                # v_permlane*_swap and v_swap write their second operand too
                if nxt.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")):
                    written |= regs(nxt.split()[2].rstrip(",")) if len(nxt.split()) > 2 else set()
                if written & data:
                    hits.append((k, ins, nxt))
                    break
    return hits


if __name__ == "__main__":
    found = scan(assembly())
    for k, st, nx in found:
        print(f"{k}: {st}   <-   {nx}")
    print(f"{len(found)} unprotected 96/128-bit buffer stores followed by a write of their data registers")
    sys.exit(1 if found else 0)
