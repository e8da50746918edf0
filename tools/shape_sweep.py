#!/usr/bin/env python3
"""bf16 against f32 over odd batch sizes (1 group .. ragged strips and tiles): embeddings and gradients must agree to bf16
noise.  A guard for the strip / tile / prefetch edge cases of the conv and GEMM kernels.  usage: python tools/shape_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd.engine import Engine

T = 41
bad = 0
for groups in (1, 2, 3, 5, 13, 17, 100, 389, 1001, 4099):
    n = groups * T
    g = torch.Generator().manual_seed(groups)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + 0.5 * torch.randn(groups, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(groups).cuda()
    res = {}
    for dtype in ("f32", "bf16"):
        e = Engine(adabn=False, dtype=dtype, dp_emg=0.0, device="cuda", seed=9)
        e.init_parameters(4)
        e.grads.flat.zero_()
        z = e.encoder_forward(x, training=True)
        out, pred, _ = e.head(z, labels, 1, want_grad=True)
        e.encoder_backward(x)
        torch.cuda.synchronize()
        res[dtype] = (z.double().cpu(), e.grads.flat.double().cpu(), float(out[0]))
    za, ga, la = res["f32"]
    zb, gb, lb = res["bf16"]
    zc = float((za * zb).sum() / (za.norm() * zb.norm()))
    gc = float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-30))
    ok = zc > 0.995 and gc > (0.95 if groups < 100 else 0.99) and abs(la - lb) < 2e-2 and torch.isfinite(gb).all()
    bad += not ok
    print(f"{groups:5d} groups ({n:6d} windows): z cos {zc:.5f}  grad cos {gc:.5f}  loss {la:.5f} / {lb:.5f}  {'ok' if ok else 'MISMATCH'}")
sys.exit(1 if bad else 0)
