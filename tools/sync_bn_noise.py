#!/usr/bin/env python3
"""Noise floor for the synchronised-BatchNorm equivalence test (tests/test_gpu_global_batch.py): how far do the f32 gradients
of ONE rank move when (a) the statistics go through the sync path with an identity hook (same numbers, summed as one f32 row
instead of f64 over the partial rows) and (b) the 48 groups are merely permuted (same set of rows, another summation order)?
Measured on MI355X: both move single gradient tensors by 2e-3 .. 2e-2 of their maximum (a few ReLUs sit within an ulp of
zero at random initialisation and flip), which is the tolerance the 2-rank test has to leave.  usage: python tools/sync_bn_noise.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/contrastiveprosthetics_amd") else os.environ["GRAFT_REPO_ROOT"])
from contrastiveprosthetics_amd.engine import Engine
T, B = 41, 48
g = torch.Generator().manual_seed(5)
x_all = (torch.randn(T, 12, generator=g)[None] + torch.randn(B, T, 12, generator=g)).reshape(B * T, 12).cuda()
def run(x, hook, scale_world=1):
    eng = Engine(adabn=False, dtype="f32", dp_emg=0.0, device="cuda", seed=3)
    eng.init_parameters(11)
    if hook: eng.set_sync_bn(lambda t: t, scale_world)
    labels = torch.arange(T).repeat(x.shape[0] // T).cuda()
    eng.grads.flat.zero_()
    z = eng.encoder_forward(x, training=True)
    out, pred, _ = eng.head(z, labels, 1, want_grad=True)
    eng.encoder_backward(x)
    torch.cuda.synchronize()
    eng.set_sync_bn(None)
    return z.clone(), eng.grads.flat.clone(), eng.grads.offsets
z0, g0, off = run(x_all, False)
z1, g1, _ = run(x_all, True)
print("hook(identity, world 1) vs none: z", float((z0-z1).abs().max()), "grads max rel", float((g0-g1).abs().max()/g0.abs().max()))
for k,(o,n) in off.items():
    a,b = g1[o:o+n], g0[o:o+n]
    if float(b.abs().max())>0:
        r = float((a-b).abs().max()/b.abs().max())
        if r > 1e-5: print("  ", k, r)
# permuted rows (groups shuffled): same set, different order -> rounding only
perm = torch.randperm(B, generator=g)
xp = x_all.reshape(B, T, 12)[perm.cuda()].reshape(B*T, 12).contiguous()
z2, g2, _ = run(xp, False)
print("permuted groups vs original (no sync): grads:")
for k,(o,n) in off.items():
    a,b = g2[o:o+n], g0[o:o+n]
    if float(b.abs().max())>0:
        r = float((a-b).abs().max()/b.abs().max())
        if r > 1e-5: print("  ", k, r)
