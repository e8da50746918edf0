"""How many stored activations are > 0 on one kernel path and == 0 on the other (f32, same inputs and weights): the explanation of
tools/fuzz_small_vs_large.py's erratic rows (profiles/r04_relu_mask_flips.txt).  usage: python tools/relu_mask_flips.py"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from contrastiveprosthetics_amd.engine import Engine
T = 41
def acts(groups, no_small):
    n = groups * T
    g = torch.Generator().manual_seed(17)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(groups, T, 12, generator=g)).reshape(n, 12).cuda()
    e = Engine(adabn=False, dtype="f32", dp_emg=0.0635, device="cuda", seed=321)
    e.init_parameters(7)
    gg = torch.Generator().manual_seed(5)
    for k in e.specs:
        v = e.values.views[k]
        if k.startswith("emg_net.") and v.dim() == 1 and v.numel() in (64, 512) and (".bn" in k or "conv_emg.2" in k or "conv_emg.5" in k or k.split(".")[-2] in ("2", "5", "8", "11", "15", "19", "23")):
            v.copy_((1.0 + 0.2 * torch.randn(v.shape, generator=gg) if k.endswith("weight") else 0.1 * torch.randn(v.shape, generator=gg)).cuda())
    e.options["no_small"] = 1 if no_small else 0
    e.encoder_forward(x, training=True)
    torch.cuda.synchronize()
    return [e.debug_activation(l).clone() for l in range(1, 9)]
for groups in (5, 31, 32, 17):
    a, b = acts(groups, False), acts(groups, True)
    flips = [int(((u > 0) != (v > 0)).sum()) for u, v in zip(a, b)]
    mx = [float((u - v).abs().max()) for u, v in zip(a, b)]
    print(groups, "groups: ReLU-mask mismatches per layer 1..8:", flips, " max |activation difference|:", ["%.1e" % m for m in mx])
