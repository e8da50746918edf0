import os, sys
sys.path.insert(0, os.getcwd())
import torch
from contrastiveprosthetics_amd.engine import Engine
T = 41
for groups, sched in ((8192, 0), (8192, 1), (12001, 0)):
    n = groups * T
    g = torch.Generator().manual_seed(3)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(groups, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(groups).cuda()
    e = Engine(adabn=False, dtype="bf16", dp_emg=0.0635, device="cuda", seed=123)
    e.tile_schedule = sched
    e.init_parameters(5)
    e.grads.flat.zero_()
    z = e.encoder_forward(x, training=True)
    out, pred, _ = e.head(z, labels, 1, want_grad=True)
    e.encoder_backward(x)
    torch.cuda.synchronize()
    fin = all(torch.isfinite(v).all() for k, v in e.grads.views.items() if k.startswith("emg_net."))
    print(groups, "groups", n, "windows, schedule", sched, "loss", float(out[0]), "acc", float(out[1]) / n, "finite grads", fin,
          "|g|", float(e.grads.flat.norm()))
    e.tile_schedule = 0
    del e, x, z
    torch.cuda.empty_cache()
