#!/usr/bin/env python3
"""Throughput of the raw-sEMG preprocessing at the size of the reference's data set (46 people x 41 stimuli x 6
repetitions = 11,316 segments of 2010 x 12 float32 = 1.09 GB) on the device, against the host path the
reference runs (scipy.signal.lfilter + scipy.ndimage.uniform_filter1d per channel, code/utils.py:137-156).
usage: python tools/preprocess_bench.py [segments]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from contrastiveprosthetics_amd import preprocess as P

S = int(sys.argv[1]) if len(sys.argv) > 1 else 46 * 41 * 6
g = torch.Generator(device="cuda").manual_seed(0)
raw = (torch.randn(S, P.SEGMENT_LEN, 12, device="cuda", generator=g) * 2e-5).contiguous()
b, a = P.butter_bandpass()
for _ in range(2):
    seg = P.preprocess_segments(raw, b, a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    seg = P.preprocess_segments(raw, b, a)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
mean, std = P.emg_stats(seg)            # warm-up (first launch loads the code objects)
e0.record()
mean, std = P.emg_stats(seg)
P.normalize_(seg, mean, std)
e1.record()
torch.cuda.synchronize()
ms2 = e0.elapsed_time(e1)
gb = raw.numel() * 4 / 1e9
print(f"device: {S} segments ({gb:.2f} GB raw) filter+rms+sample {ms:.2f} ms ({gb / ms * 1e3:.0f} GB/s of raw input, "
      f"{S * P.SEGMENT_LEN * 12 / ms / 1e6:.1f} G samples/s); statistics + normalise {ms2:.2f} ms")
try:
    from scipy import signal
    from scipy.ndimage import uniform_filter1d
except ImportError:
    sys.exit(0)
host = raw[:64].cpu().numpy()
t0 = time.perf_counter()
n = 0
for r in host:
    x = r * np.float32(1024)
    xt = x.T
    for i in range(12):
        xt[i] = signal.lfilter(b, a, xt[i])
    out = np.transpose([np.sqrt(uniform_filter1d(np.square(t), size=11, mode="nearest"))[5:-5] for t in x.T])[P.time_mask()]
    n += 1
dt = time.perf_counter() - t0
ref = seg  # normalised by now; compare un-normalised values of the sample instead
chk = P.preprocess_segments(raw[:64].contiguous(), b, a)[n - 1].cpu().numpy()
print(f"host (SciPy, 1 thread): {n} segments in {dt * 1e3:.1f} ms -> {dt / n * S:.1f} s for all {S} "
      f"(x{dt / n * S / (ms / 1e3):.0f}); last segment identical: {np.array_equal(chk, out)}")
