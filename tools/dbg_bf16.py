import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import ref_cpu as oc
from tests.test_gpu_parity import *

for adabn in (True,):
    B = 16
    sd = oc.init_state_dict(31, 16, adabn)
    EMG = randn(404, (B, T, 1, 1, 12))
    label = torch.arange(T).repeat(B)
    taps = {}
    m = oc.OracleModel(sd, BEST, adabn=adabn)
    logits_ref = m.forward(EMG, torch.zeros(B, T, 20), label, taps)
    for dtype in ("f32", "bf16"):
        e = make_engine(sd, adabn, dtype)
        out, pred, logits = run_step(e, EMG, label)
        print(dtype)
        for l in range(9):
            got = to_ref_layout(e.debug_activation(l), l)
            ref = taps[f"r{l}"]
            rel = float((got - ref).norm() / ref.norm())
            print(f"  layer {l}: rel L2 err {rel:.3e}  max abs {float((got-ref).abs().max()):.3e} (ref rms {float(ref.pow(2).mean().sqrt()):.3e})")
        z = None
        d = (logits.cpu() - logits_ref)
        srt = torch.sort(logits_ref, dim=-1, descending=True)[0]
        print(f"  logits: max {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}; ref top2 margin median {float((srt[...,0]-srt[...,1]).median()):.3e}; argmax agree {float((pred.cpu()==logits_ref.argmax(-1)).float().mean()):.4f}; loss {out[0].item():.6f} ref {m.loss_vectorized(logits_ref,label).item():.6f}")
