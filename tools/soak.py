#!/usr/bin/env python3
"""Soak run: many training steps of the full-size step on synthetic class-structured data, twice from the same seed -- the loss must stay
finite and fall, and the two runs must end with bit-identical weights (every reduction of the step has a fixed order).
usage: python tools/soak.py [dtype=bf16] [groups=4096] [steps=1500]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd.engine import Engine

T = 41
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0635, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.3817)
DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
D = 4200


def run():
    g = torch.Generator().manual_seed(0)
    table = (torch.randn(T, 1, 12, generator=g) + 0.8 * torch.randn(T, D, 12, generator=g)).reshape(T * D, 12).cuda()
    emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).cuda()
    labels = torch.arange(T).repeat(B).cuda()
    e = Engine(adabn=False, dtype=DT, dp_emg=BEST["dp_emg"], device="cuda", seed=7)
    e.init_parameters(3)
    losses = []
    t0 = time.time()
    for s in range(STEPS):
        p = torch.randperm(D, generator=g)[:B].cuda()
        x = e.gather(table, emg_rand, p, 1)
        z = e.encoder_forward(x, training=True)
        out, _, _ = e.head(z, labels, 1, want_grad=True)
        e.encoder_backward(x)
        e.adam_step(BEST)
        if s % 250 == 0 or s == STEPS - 1:
            losses.append(float(out[0]))
            print(f"  step {s:5d} loss {losses[-1]:.5f}  ({time.time() - t0:.1f} s)", flush=True)
    torch.cuda.synchronize()
    w = e.values.flat.detach().cpu()
    assert torch.isfinite(w).all(), "non-finite weights"
    return losses, hashlib.sha256(w.numpy().tobytes()).hexdigest()


l1, h1 = run()
l2, h2 = run()
assert all(x == x and abs(x) < 1e3 for x in l1), l1
assert l1[-1] < l1[0] - 0.5, (l1[0], l1[-1])
print(f"{DT} {B} groups x {STEPS} steps: loss {l1[0]:.4f} -> {l1[-1]:.4f}; weights sha256 {h1[:16]} / {h2[:16]}: {'bit-identical' if h1 == h2 else 'DIFFERENT'}")
assert l1 == l2 and h1 == h2, "the two runs differ"
