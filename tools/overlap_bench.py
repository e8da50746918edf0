#!/usr/bin/env python3
"""Does an HBM-bound elementwise pass hide behind a power-bound GEMM launch on another stream?  Times a GEMM launch
(weight gradient, weight-stationary forward, weight-stationary data gradient) and a torch elementwise pass of
bn_relu_bwd's traffic (read 2 x 172 MB, write 172 MB) serially and on two streams.  usage: python tools/overlap_bench.py"""
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
R = torch.randn(M, F, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(F, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
C = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
slabs = torch.empty(64 * 512 * 512, device=dev)
bias = torch.zeros(F, device=dev)
coef = torch.randn(3 * 512, device=dev)
partials = torch.zeros(4 * (M // 32 + 8) * F, device=dev)
e1_, e2_, e3_ = (torch.randn(M, F, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def gemm(kind):
    if kind == "wgrad":
        _lib.check(lib.cp_debug_gemm(1, 2, M, 512, 512, A.data_ptr(), R.data_ptr(), slabs.data_ptr(), 0, 0, partials.data_ptr(), 0, main.cuda_stream), "wgrad")
    elif kind == "fwd":
        _lib.check(lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), 128,
                                     main.cuda_stream), "fwd")


def elementwise(stream):
    with torch.cuda.stream(stream):
        torch.add(e1_, e2_, out=e3_)


def timed(fn, reps=6):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(reps):
            fn()
        e1.record(main)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


def both_serial(kind):
    gemm(kind)
    elementwise(main)


def both_overlap(kind):
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    elementwise(side)
    gemm(kind)
    ev2 = torch.cuda.Event()
    ev2.record(side)
    main.wait_event(ev2)


for _ in range(3):
    gemm("wgrad"); gemm("fwd"); elementwise(main)
print("elementwise alone   %7.1f us" % timed(lambda: elementwise(main)))
for kind in ("wgrad", "fwd"):
    print("%-6s alone        %7.1f us" % (kind, timed(lambda: gemm(kind))))
    print("%-6s + pass serial %7.1f us" % (kind, timed(lambda: both_serial(kind))))
    print("%-6s + pass 2 strm %7.1f us" % (kind, timed(lambda: both_overlap(kind))))
