#!/usr/bin/env python3
"""Weight-stationary forward GEMM (gemm_ws.cuh) against the tile-staged persistent kernel (gemm_nt256p.cuh) through
cp_debug_gemm: same outputs (bit for bit: same k order), column sums equal to rounding, interleaved timing in one process.
usage: python tools/ws_bench.py [M] [F]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contrastiveprosthetics_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 167936
F = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = 512
lib = _lib.load()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(F, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(F, device=dev, generator=g) * 0.1
st = torch.cuda.current_stream().cuda_stream
rows = (M + 63) // 64 + 8


def run(dbg):
    C = torch.zeros(M, F, device=dev, dtype=torch.bfloat16)
    partials = torch.zeros(rows * 2 * F, device=dev)
    _lib.check(lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), dbg, st), "gemm")
    torch.cuda.synchronize()
    return C, partials.view(rows, 2, F).double().sum(0)


Cw, pw = run(0)
Co, po = run(256)
C32, p32 = run(512)
d16 = (Cw.float() - Co.float()).abs()
print("16x16x32 form vs tile-staged kernel: differing elements %d of %d, max |diff| %.4g (bf16 ulp at the max value: %.4g)" %
      (int((Cw != Co).sum()), Cw.numel(), float(d16.max()), float(Co.float().abs().max()) * 2 ** -8))
print("32x32x16 form == tile-staged kernel, bit for bit:", bool(torch.equal(C32, Co)))
del d16, C32
ref = torch.relu(A[:8192].float() @ W.float().t() + bias)
print("ws  vs torch (first 8192 rows): max err", float((Cw[:8192].float() - ref).abs().max()), "of", float(ref.abs().max()))
tail = torch.relu(A[-200:].float() @ W.float().t() + bias)
print("ws  vs torch (last 200 rows):   max err", float((Cw[-200:].float() - tail).abs().max()))
print("ws == tile-staged kernel, bit for bit:", bool(torch.equal(Cw, Co)), " mismatching elements:", int((Cw != Co).sum()))
if not torch.equal(Cw, Co):
    bad = (Cw != Co).nonzero()
    rows_b, cols_b = bad[:, 0].unique(), bad[:, 1].unique()
    print("  mismatching rows %d..%d (%d), cols %d..%d (%d); NaN in ws: %d, in old: %d" % (int(rows_b.min()), int(rows_b.max()), rows_b.numel(),
          int(cols_b.min()), int(cols_b.max()), cols_b.numel(), int(torch.isnan(Cw.float()).sum()), int(torch.isnan(Co.float()).sum())))
    print("  cols:", cols_b.tolist()[:40], "...")
    print("  rows//64 (tiles):", (rows_b // 64).unique().tolist()[:40], " rows%64:", (rows_b % 64).unique().tolist()[:64])
    print("  sample ws", Cw[rows_b[0], cols_b[:4]].tolist(), "old", Co[rows_b[0], cols_b[:4]].tolist())
Cf = Cw.double()
print("sum   : ws %.3e  old %.3e  (max |col| %.3e)" % (float((pw[0] - Cf.sum(0)).abs().max()), float((po[0] - Cf.sum(0)).abs().max()), float(Cf.sum(0).abs().max())))
print("sumsq : ws %.3e  old %.3e  (max %.3e)" % (float((pw[1] - (Cf * Cf).sum(0)).abs().max()), float((po[1] - (Cf * Cf).sum(0)).abs().max()), float((Cf * Cf).sum(0).max())))
del Cf

C = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
partials = torch.empty(rows * 2 * F, device=dev)


def timed(dbg, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), dbg, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if os.environ.get("WS_STAMP"):
    # a -DCP_VARIANTS -DWS_STAMP build (tools/ws_stamps.sh): the 32x32x16 form (dbg 512) carries the stamps
    partials.zero_()
    lib.cp_debug_gemm(1, 0, M, K, F, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), 0, partials.data_ptr(), 512, st)
    torch.cuda.synchronize()
    stv = partials[200 * 2 * F: 200 * 2 * F + 2 * 2 * 3 * 24].view(torch.int64).cpu().tolist()
    cyc, rt = stv[0::2], stv[1::2]
    nt = sum(1 for i in range(24) if cyc[3 * i])
    print("in-kernel stamps, block 0 wave 0: tile: k loop cycles (us, GHz) | closing wait + barrier cycles | gap to the next tile")
    for i in range(nt):
        nxt = cyc[3 * i + 3] - cyc[3 * i + 2] if i + 1 < nt else 0
        kc, kus = cyc[3 * i + 1] - cyc[3 * i], (rt[3 * i + 1] - rt[3 * i]) / 100.0
        print("  tile %2d: %6d (%5.2f us, %4.2f GHz) | %6d | %6d" % (i, kc, kus, kc / kus / 1e3 if kus > 0 else 0.0, cyc[3 * i + 2] - cyc[3 * i + 1], nxt))
    tc, tus = cyc[3 * nt - 1] - cyc[0], (rt[3 * nt - 1] - rt[0]) / 100.0
    print("  whole: %d cycles in %.2f us from the first stamp to the last = %.2f GHz" % (tc, tus, tc / tus / 1e3))
    for _ in range(2):
        timed(512, 5)
    v = sorted(timed(512) for _ in range(6))
    print("  launch (32x32x16 form, this build): median %.1f us  min %.1f  max %.1f" % (v[3], v[0], v[-1]))
    sys.exit(0)
for _ in range(2):
    timed(0, 5), timed(512, 5), timed(256, 5)
res = {0: [], 512: [], 256: []}
for rnd in range(6):
    for dbg in (0, 512, 256):
        res[dbg].append(timed(dbg))
flops = 2.0 * M * K * F
for dbg, name in ((0, "weight-stationary 16x16x32"), (512, "weight-stationary 32x32x16"), (256, "tile-staged persistent")):
    v = sorted(res[dbg])
    print(f"{name:28s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}   {flops / v[len(v) // 2] / 1e6:6.0f} TFLOP/s  "
          f"{(M * (K + F) * 2) / v[len(v) // 2] / 1e3:6.0f} GB/s")
