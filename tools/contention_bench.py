#!/usr/bin/env python3
"""How a persistent one-block-per-CU GEMM launch behaves when another stream's kernel holds some CUs (as an RCCL
collective does during multi-GPU steps): time of the fc forward launch alone and beside `hog` workgroups that each
occupy a CU.  usage: python tools/contention_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd import _lib

M, K, F = 167936, 512, 512
lib = _lib.load()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(F, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
C = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(F, device=dev)
partials = torch.zeros(4 * (M // 128 + 8) * F, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def gemm(dbg=0):
    _lib.check(lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), dbg,
                                 main.cuda_stream), "cp_debug_gemm")


for hog in (0, 8, 16, 32):
    for dbg, name in ((128, "persistent, static tiles"), (64, "persistent, dynamic tiles"), (24, "one tile per block")):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            if hog:
                _lib.check(lib.cp_debug_hog(hog, 3000, side.cuda_stream), "cp_debug_hog")
                torch.cuda._sleep(200000)                   # let the hog settle on its CUs first
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(4):
                gemm(dbg)
            e1.record(main)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4 * 1e3)
        print(f"{hog:3d} CUs held elsewhere, {name:26s}: {sorted(ts)[2]:7.1f} us per launch")
