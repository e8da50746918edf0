#!/usr/bin/env python3
"""A few training steps at a small batch for rocprofv3 --kernel-trace (tools/step_kernels.py reads the trace):
   rocprofv3 --kernel-trace -d gpurun_out/sm_trace -- python3 tools/small_step_trace.py 8 bf16 [no_small]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd.engine import Engine

T = 41
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0635, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DT = sys.argv[2] if len(sys.argv) > 2 else "bf16"
D = 1800
g = torch.Generator().manual_seed(0)
table = (torch.randn(T, 1, 12, generator=g) + torch.randn(T, D, 12, generator=g)).reshape(T * D, 12).cuda()
emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).cuda()
labels = torch.arange(T).repeat(B).cuda()
e = Engine(adabn=False, dtype=DT, dp_emg=BEST["dp_emg"], device="cuda", seed=1)
e.options["no_small"] = 1 if "no_small" in sys.argv else 0
e.init_parameters(2)
for s in range(12):
    p = torch.randperm(D, generator=g)[:B].cuda()
    x = e.gather(table, emg_rand, p, 1)
    z = e.encoder_forward(x, training=True)
    out, _, _ = e.head(z, labels, 1, want_grad=True)
    e.encoder_backward(x)
    e.adam_step(BEST)
torch.cuda.synchronize()
