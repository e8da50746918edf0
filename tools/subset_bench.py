#!/usr/bin/env python3
"""Time the README's class-subset experiment on the device (cp_subset_vote) against the numpy oracle.
usage: python tools/subset_bench.py [groups] [trials]      (defaults: the reference's test split, 48 groups, 144 trials)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from contrastiveprosthetics_amd import engine as E
from contrastiveprosthetics_amd.results import random_subsets
from oracle import eval_cpu as ev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 144
V, T = 25, 41
rng = np.random.default_rng(0)
logits = rng.standard_normal((B * V, T, T)).astype(np.float32)
masks = random_subsets(range(2, T + 1), trials, 0)
lg, lab, mk = torch.from_numpy(logits).cuda(), torch.arange(T).cuda(), torch.from_numpy(masks).cuda()
for _ in range(2):
    c = E.subset_vote(lg, lab, B, V, mk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    c = E.subset_vote(lg, lab, B, V, mk)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
n = masks.shape[0]
print(f"device: {n} subsets x {B} groups x {V} samples in {ms:.3f} ms  ({n * B * V * T / ms / 1e6:.1f} G row-decisions/s)")
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 10.0 and k < n:
    ref, _ = ev.subset_vote(logits, np.arange(T), B, V, masks[k])
    assert np.array_equal(ref, c[k].cpu().numpy()), k
    k += 1
dt = time.perf_counter() - t0
print(f"numpy oracle: {k} subsets in {dt:.2f} s -> {dt / k * n:.1f} s for all {n} (x{dt / k * n / (ms / 1e3):.0f}); all {k} agree")
