#!/usr/bin/env python3
"""Where a small-batch forward launch spends its time: needs a library built with -DSM_STAMP (CPNATIVE_LIB=build/libcp_smstamp.so);
prints, per fc layer, workgroup 0's 100 MHz timestamps: prologue (statistics), k loop, epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrastiveprosthetics_amd.engine import Engine
T = 41
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = Engine(adabn=False, dtype=sys.argv[2] if len(sys.argv) > 2 else "bf16", dp_emg=0.0635, device="cuda", seed=1)
e.init_parameters(2)
x = torch.randn(B * T, 12, device="cuda")
for _ in range(5):
    z = e.encoder_forward(x, training=True)
torch.cuda.synchronize()
nb = e.lib.cp_workspace_bytes(B * T, e.dtype, e.dp_emg)
# the accumulator block sits 2 x 2 x 768 x 4 bytes before the end of the workspace (sync_loc, sync_glob follow it)
ws = e._ws
acc_off = nb - 2 * (2 * 768 * 4) - 18 * 2 * 768 * 8
acc_off -= acc_off % 256
acc = ws[acc_off:acc_off + 18 * 2 * 768 * 8].view(torch.int64).view(18, 2 * 768).cpu()
for L in range(2, 9):
    st = acc[L, 1100:1104].tolist()
    if st[0] == 0:
        print("no stamps (library without -DSM_STAMP, or the offset guess is off)"); break
    print(f"fc{L - 1}: loads issued + prologue {(st[1] - st[0]) / 100:.2f} us, k loop {(st[2] - st[1]) / 100:.2f} us, epilogue {(st[3] - st[2]) / 100:.2f} us")
    ls = acc[L, 1110:1130].tolist()
    if ls[0]:
        names = ["issue", "mid", "store0", "sync", "mma0", "sync", "store1", "sync", "mma1", "sync"]
        t0 = st[0]
        print("      loop stamps (us from kernel start): " + "  ".join(f"{n} {(v - t0) / 100:.2f}" for n, v in zip(names, ls) if v))
