#!/usr/bin/env python3
"""SURVEY 8d acceptance check of the CPU baseline: the oracle's full training step (oracle/ref_cpu.py, the "port" that
bench.py times on the GPU box's host cores) against the REFERENCE ITSELF imported on CPU (same shim as tools/make_golden.py),
same weights, same synthetic batches, same thread count, in the build container.  The port is accepted as a stand-in for
the reference's CPU path only if its step time is within +-10 % of the reference's at B=8 (BASELINE config 1) and B=64.

    python tools/time_port_vs_reference.py [--threads 8] [--steps 20]     -> one JSON line; paste the ratios into DESIGN.md

A step on both sides = code/train.py:95-108: forward, loopy loss (with the per-group accuracy bookkeeping), l2, zero_grad,
backward, two Adam steps; stock BN (--no_adabn, config 1), dp_emg = 0.0635 (best key), fp32.
"""
import argparse
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0635, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.3817)
T = 41


def interleaved_medians(fns, steps, rounds=4, warm=3):
    """Median step time of each callable, measured in alternating rounds inside this one process (a single block per side
    made the side that ran first look 20 % slower at B=8: allocator and oneDNN primitive caches warm up during it)."""
    for f in fns:
        for i in range(warm):
            f(i)
    ts = [[] for _ in fns]
    per = max(1, steps // rounds)
    for r in range(rounds):
        for k, f in enumerate(fns):
            for i in range(per):
                t0 = time.perf_counter()
                f(warm + r * per + i)
                ts[k].append(time.perf_counter() - t0)
    return [float(np.median(t)) for t in ts]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from make_golden import import_reference
    from oracle import ref_cpu as oc
    rutils, rmodels, rload = import_reference()
    out = {"threads": a.threads, "steps": a.steps, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")}
    for B in (8, 64):
        g = torch.Generator().manual_seed(B)
        batches = [torch.randn(B, T, 1, 1, 12, generator=g) for _ in range(a.steps + 3)]
        label = torch.arange(T).repeat(B)
        glove = torch.zeros(B, T, 20)
        sd = oc.init_state_dict(0, 16, adabn=False)
        # the port
        m = oc.OracleModel(sd, BEST, adabn=False, requires_grad=True)
        m.set_train()
        opts = m.make_optimizers()
        port_step = lambda i: m.train_step(batches[i % len(batches)], glove, label, opts)
        # the reference (code/models.py Model + the loop body of code/train.py:95-108)
        ref = rmodels.Model(params=dict(BEST), train_model=True, adabn=False, prediction=False, glove=False, device="cpu").to(torch.float32)
        ref.load_state_dict(sd, strict=True)
        ref.set_train()
        oe = torch.optim.Adam(ref.emg_net.parameters(), lr=BEST["lr_emg"], weight_decay=0)
        og = torch.optim.Adam(ref.glove_net.parameters(), lr=BEST["lr_glove"], weight_decay=0)

        def ref_step(i):
            logits = ref.forward(batches[i % len(batches)], glove, label)
            loss = ref.loss(logits, label)
            loss.item()
            loss = loss + ref.l2()
            oe.zero_grad(set_to_none=True)
            og.zero_grad(set_to_none=True)
            loss.backward()
            oe.step()
            og.step()
        t_port, t_ref = interleaved_medians([port_step, ref_step], a.steps)
        out[f"B{B}"] = dict(port_ms=1e3 * t_port, reference_ms=1e3 * t_ref, port_over_reference=t_port / t_ref,
                            port_windows_per_s=B * T / t_port, reference_windows_per_s=B * T / t_ref,
                            within_10_percent=bool(abs(t_port / t_ref - 1) <= 0.10))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
