#!/usr/bin/env python3
"""Micro-benchmark of one fc-layer GEMM launch (cp_debug_gemm) with ablations, on random data.
usage: python tools/gemm_bench.py [M] [K] [F]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 167936
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
F = int(sys.argv[3]) if len(sys.argv) > 3 else 512
lib = _lib.load()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(F, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
R = torch.randn(M, F, device=dev, generator=g).to(torch.bfloat16)
C = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
slabs = torch.empty(64 * K * F + 1024, device=dev)
bias = torch.zeros(F, device=dev)
partials = torch.empty(4 * (M // 128 + 8) * F, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(kind, dbg, iters=20, with_r=True):
    out = slabs if kind == 2 else C
    w = R if kind == 2 else W           # wgrad: Y operand is [M][F]
    rp = R.data_ptr() if with_r else 0
    for _ in range(3):
        _lib.check(lib.cp_debug_gemm(1, kind, M, K, F, A.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                     rp, partials.data_ptr(), dbg, st), "cp_debug_gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.cp_debug_gemm(1, kind, M, K, F, A.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr(), rp,
                          partials.data_ptr(), dbg, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


flops = 2.0 * M * K * F
names = {0: "fwd", 1: "dgrad", 2: "wgrad"}
for kind in (0, 1, 2):
    for dbg, what in ((0, "full"), (2, "no epilogue"), (1, "no MFMA"), (3, "loads+barriers only"), (6, "MFMA+LDS reads only"),
                      (4, "no staging loads")):
        if kind == 2 and dbg:
            continue
        us = run(kind, dbg | (16 if dbg else 0))      # ablations exist in the one-tile-per-block kernel only
        print(f"{names[kind]:6s} {what:22s} {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s-equivalent")


print(f"fwd    persistent, 256-row tiles (dbg 32) {run(0, 32):8.1f} us")
print(f"dgrad  persistent, 256-row tiles, no statistics (dbg 32) {run(1, 32, with_r=False):8.1f} us")
print(f"fwd    one tile per block, direct stores (dbg 16) {run(0, 16):8.1f} us")
print(f"fwd    one tile per block, staged epilogue (dbg 24) {run(0, 24):8.1f} us")
print(f"dgrad  no statistics (R = null): persistent {run(1, 0, with_r=False):8.1f} us")
print(f"dgrad  no statistics, one tile per block: direct {run(1, 16, with_r=False):8.1f} us, staged {run(1, 24, with_r=False):8.1f} us")
# numerical check of the forward against torch (bf16 inputs, f32 accumulate)
lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), R.data_ptr(), partials.data_ptr(), 0, st)
ref = torch.relu(A[:4096].float() @ W.float().t())
print("fwd max err vs torch:", float((C[:4096].float() - ref).abs().max()), "of", float(ref.abs().max()))

bias.copy_(torch.randn(F, device=dev) * 0.1)
for dbg in (0, 32, 24):
    C.zero_()
    partials.zero_()
    lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), R.data_ptr(), partials.data_ptr(), dbg, st)
    tiles = (M + 255) // 256
    p = partials[: tiles * 2 * F].view(tiles, 2, F).double().sum(0)
    Cf = C.double()
    ref = torch.relu(A[-3000:].float() @ W.float().t() + bias)
    print(f"dbg {dbg}: tail-row max err {float((C[-3000:].float() - ref).abs().max()):.4f}; "
          f"sum err {float((p[0] - Cf.sum(0)).abs().max()):.3e} of {float(Cf.sum(0).abs().max()):.3e}; "
          f"sumsq err {float((p[1] - (Cf * Cf).sum(0)).abs().max()):.3e} of {float((Cf * Cf).sum(0).max()):.3e}")
    del Cf
Cd = torch.empty_like(C)
lib.cp_debug_gemm(1, 1, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), 0, st)
lib.cp_debug_gemm(1, 1, M, K, F, A.data_ptr(), W.data_ptr(), Cd.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), 24, st)
print("dgrad persistent == staged:", bool((C == Cd).all()))
