import torch, sys
sys.path.insert(0, "/root/repo")
from contrastiveprosthetics_amd.engine import Engine
from contrastiveprosthetics_amd import _lib
T=41; B=4096; N=T*B
g = torch.Generator().manual_seed(6)
mu = torch.randn(T, 12, generator=g)
x = (mu[None] + torch.randn(B, T, 12, generator=g)).reshape(N, 12).cuda()
labels = torch.arange(T).repeat(B).cuda()
e = Engine(adabn=False, dtype="fp8", dp_emg=0.0635, device="cuda", seed=1000)
e.init_parameters(5)
tap = torch.zeros(9, N, 768, dtype=torch.bfloat16, device="cuda")
lib=_lib.load()
for it in range(3):
    e.step_count=0
    e.grads.flat.zero_()
    if it==2: e.grad_tap = tap
    z = e.encoder_forward(x, training=True)
    e.head(z, labels, 1, want_grad=True)
    e.encoder_backward(x)
    torch.cuda.synchronize()
e.grad_tap = None
ex = e.fp8_scale_exponents()
print("exps act", ex[1:12].tolist(), "grad", ex[16:25].tolist(), "gb", ex[32:41].tolist())
LIN=(0,3,6,9,13,17,21)
for L in (8,7,6,5,4):
    i=L-2
    gy = tap[L].reshape(-1)[:N*512].reshape(N,512).float()
    got = e.grads.views[f"emg_net.linear.{LIN[i]}.bias"]
    ref = gy.sum(0)
    l1 = gy.abs().sum(0)
    zero = float((gy==0).float().mean())
    amax = float(gy.abs().max()); med = float(gy.abs().median())
    sc = 2.0**int(ex[16+L])
    print(f"L={L} fc{i+1}_b: max|got-ref|/max|ref| {float((got-ref).abs().max()/ref.abs().max()):.3f}  |got-ref|/L1 max {float(((got-ref).abs()/l1).max()):.5f}  |ref|/L1 median {float((ref.abs()/l1).median()):.5f} zeros {zero:.3f} amax*scale {amax*sc:.1f} median*scale {med*sc:.5f}")
