"""fc1 forward (K = 768): gemm_ws16n_kernel (one wave owns 32 features x all k, round 4) against gemm_ws16k_kernel (k split over wave
pairs; cp_debug_gemm dbg bit 1024), alternating in one process: correctness of both against an fp32 recomputation, then the times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastiveprosthetics_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
K, F = 768, 512
g = torch.Generator(device="cuda").manual_seed(0)
for M in (41 * 7, 40057, 167936):
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(F, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(F, device=dev, generator=g) * 0.1
    ref = torch.relu(A.float() @ W.float().t() + bias)
    for dbg in (0, 1024):
        C = torch.full((M, F), float("nan"), device=dev, dtype=torch.bfloat16)
        partials = torch.zeros(4 * (M // 32 + 600) * F, device=dev)
        _lib.check(lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), dbg, st), "gemm")
        torch.cuda.synchronize()
        err = float((C.float() - ref).abs().max() / ref.abs().max())
        rows = 128 if dbg else 64
        tiles = (M + (32 if dbg else 48) - 1) // (32 if dbg else 48)
        nrows = min(tiles, 64) * (2 if dbg else 1)
        ps = partials[:nrows * 2 * F].reshape(nrows, 2, F).double().sum(0)
        cb = C.float().double()
        e1 = float((ps[0] - ref.double().sum(0)).abs().max() / ref.double().sum(0).abs().max())
        e2 = float((ps[1] - (ref.double() ** 2).sum(0)).abs().max() / (ref.double() ** 2).sum(0).abs().max())
        print(f"M={M} dbg={dbg}: max|C-ref|/max {err:.2e} (bf16 rounding 3.9e-3), column sums {e1:.2e}, sums of squares {e2:.2e}, finite {bool(torch.isfinite(C.float()).all())}")
def t(dbg, iters=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), dbg, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rep in range(4):
    print(f"M={M}: ws16n {t(0):.1f} us   ws16k {t(1024):.1f} us")
