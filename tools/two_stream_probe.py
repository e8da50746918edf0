"""Can the ~40 latency-bound small launches of a large-batch step be filled with other work?  Two independent engines (two replicas of the
bench's step) on ONE MI355X, each on its own stream, against one engine alone: if the idle CUs beside the finalisers / folds /
reductions were usable, two concurrent steps would finish in less than twice one step.   python tools/two_stream_probe.py [bf16|fp8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastiveprosthetics_amd.engine import Engine
T, B = 41, 4096
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0635, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.3817)
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
D = 20000
table = (torch.randn(T, 1, 12, generator=g) + torch.randn(T, D, 12, generator=g)).reshape(T * D, 12).to(dev)
emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).to(dev)
labels = torch.arange(T).repeat(B).to(dev)
perms = [torch.randperm(D, generator=g)[:B].to(dev) for _ in range(8)]
engs = []
for i in range(2):
    e = Engine(adabn=False, dtype=dt, dp_emg=0.0635, device=dev, seed=i)
    e.init_parameters(42)
    e.workspace(B * T)
    engs.append(e)
streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
def step(e, i):
    x = e.gather(table, emg_rand, perms[i % 8], 1)
    z = e.encoder_forward(x, training=True)
    e.head(z, labels, 1, want_grad=True)
    e.encoder_backward(x)
    e.adam_step(BEST)
def run(n_eng, steps=20):
    for k in range(n_eng):
        with torch.cuda.stream(streams[k]):
            for i in range(3):
                step(engs[k], i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for k in range(n_eng):
            with torch.cuda.stream(streams[k]):
                step(engs[k], i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    one = run(1)
    two = run(2)
    print(f"{dt}: one engine {one:.3f} ms/step; two engines on two streams {two:.3f} ms per PAIR of steps = {two / 2:.3f} per step ({2 * one / two:.3f}x the throughput)")
