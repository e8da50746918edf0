#!/usr/bin/env python3
"""Time of the global-negatives reduction (cp_global_negatives: G and H over the gathered z of the global batch) at the row
counts of 1, 2, 4 and 8 ranks x 4096 groups.  usage: python tools/gneg_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd.engine import Engine

T = 41
e = Engine(adabn=False, dtype="bf16", dp_emg=0.0, device="cuda", seed=1)
e.init_parameters(5)
for world in (1, 2, 4, 8):
    n = world * 4096 * T
    z = torch.randn(n, 16, device="cuda")
    labels = torch.arange(T).repeat(n // T).cuda()
    for _ in range(3):
        gh = e.global_negatives(z, labels)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gh = e.global_negatives(z, labels)
    e1.record()
    torch.cuda.synchronize()
    print(f"{world} ranks x 4096 groups = {n:8d} rows: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us   G[0] {float(gh[0, 0]):.4f} H[0] {float(gh[1, 0]):.6f}")
