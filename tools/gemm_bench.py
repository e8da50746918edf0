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


def run(kind, dbg, iters=20):
    out = slabs if kind == 2 else C
    w = R if kind == 2 else W           # wgrad: Y operand is [M][F]
    for _ in range(3):
        _lib.check(lib.cp_debug_gemm(1, kind, M, K, F, A.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                     R.data_ptr(), partials.data_ptr(), dbg, st), "cp_debug_gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.cp_debug_gemm(1, kind, M, K, F, A.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr(), R.data_ptr(),
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
        us = run(kind, dbg)
        print(f"{names[kind]:6s} {what:22s} {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s-equivalent")
# numerical check of the forward against torch (bf16 inputs, f32 accumulate)
lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), R.data_ptr(), partials.data_ptr(), 0, st)
ref = torch.relu(A[:4096].float() @ W.float().t())
print("fwd max err vs torch:", float((C[:4096].float() - ref).abs().max()), "of", float(ref.abs().max()))
