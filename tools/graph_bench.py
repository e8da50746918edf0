#!/usr/bin/env python3
"""Steps per second of the training step at the reference's own batch sizes, issued call by call and replayed as one HIP
graph (engine.GraphStep).  usage: python tools/graph_bench.py [batch ...]   (default 8 32 64 128)
CP_GB_DTYPE=f32|bf16 (default bf16); CP_GB_NO_SMALL=1: batches of <= 64 groups on the large-batch kernels (cp_config.options, CP_OPT_NO_SMALL)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd.engine import Engine, GraphStep

T = 41
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0635, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.0)
D = 1800
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, 1, 12, generator=g) + torch.randn(T, D, 12, generator=g)).reshape(T * D, 12).cuda()
emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).cuda()
DT = os.environ.get("CP_GB_DTYPE", "bf16")
for B in [int(a) for a in sys.argv[1:]] or [8, 32, 64, 128]:
    perms = [torch.randperm(D, generator=g)[:B].cuda() for _ in range(64)]
    labels = torch.arange(T).repeat(B).cuda()
    res = {}
    for mode in ("calls", "graph"):
        e = Engine(adabn=False, dtype=DT, dp_emg=BEST["dp_emg"], device="cuda", seed=1)
        e.options["no_small"] = 1 if os.environ.get("CP_GB_NO_SMALL") else 0
        e.init_parameters(2)
        gs = GraphStep(e, table, emg_rand, B, BEST) if mode == "graph" else None

        def step(p):
            if gs is not None:
                return gs.step(p)
            x = e.gather(table, emg_rand, p, 1)
            z = e.encoder_forward(x, training=True)
            out, _, _ = e.head(z, labels, 1, want_grad=True)
            e.encoder_backward(x)
            e.adam_step(BEST)
            return out
        for p in perms[:8]:
            step(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            for p in perms:
                step(p)
            torch.cuda.synchronize()
            n += len(perms)
        res[mode] = (time.perf_counter() - t0) / n
    print(f"{DT}{' (large-batch kernels)' if os.environ.get('CP_GB_NO_SMALL') else ''} batch {B:4d} groups ({B * T:6d} windows): {res['calls'] * 1e3:6.3f} ms/step call by call, {res['graph'] * 1e3:6.3f} ms/step "
          f"as one graph  (x{res['calls'] / res['graph']:.2f};  {B * T / res['graph'] / 1e6:.2f} M windows/s)")
