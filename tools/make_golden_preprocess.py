#!/usr/bin/env python3
"""tests/golden/preprocess.npz: the reference's raw-sEMG preprocessing (code/load.py:102-114, code/utils.py:79-156)
run by the reference's OWN functions (utils.filter, utils.rms, utils.RunningStats, imported read-only as in
tools/make_golden.py) on seeded synthetic float32 segments shaped like one (stimulus, repetition) slice of a
Ninapro recording: 2010 samples x 12 channels at 2 kHz.  Build container only (needs /root/reference + scipy)."""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

import make_golden as mg


def synthetic_segments(seed, S, L=2010, C=12):
    """float32, volt-scale (1e-5) noise + a few tones per channel; what scipy.io.loadmat hands load.py for `emg`."""
    rng = np.random.default_rng(seed)
    t = np.arange(L)[None, :, None] / 2000.0
    f = rng.uniform(5, 900, size=(S, 1, C))
    x = 2e-5 * rng.standard_normal((S, L, C)) + 1e-5 * np.sin(2 * np.pi * f * t) + 3e-6 * rng.standard_normal((S, 1, C))
    return x.astype(np.float32)


def main():
    rutils, _, _ = mg.import_reference()
    import constants as rc
    from scipy import signal
    S, seed = 6, 77
    raw = synthetic_segments(seed, S)
    time_mask = np.arange(0, rc.TOTAL_WINDOW_SIZE, rc.FACTOR, dtype=np.uint8)        # code/load.py:115 (wraps mod 256)
    stats = rutils.RunningStats("/tmp/_cp_golden_emg_", complete=False)
    segs = []
    for s in range(S):
        emg_ = raw[s].copy()                                                         # load.py:102 slice of the recording
        emg_ = rutils.filter(emg_ * 2 ** 10, (20, 450), butterworth_order=4, btype="bandpass")   # load.py:105
        emg_ = rutils.rms(emg_)                                                      # load.py:107
        emg_ = rutils.torchize(emg_[time_mask])                                      # load.py:109
        stats.push(emg_)                                                             # load.py:141
        segs.append(emg_)
    EMG = torch.stack(segs)
    mean, std = stats.mean_std()                                                     # load.py:144
    norm = stats.normalize(EMG)                                                      # load.py:148
    b, a = signal.butter(4, [20 / 1000, 450 / 1000], btype="bandpass")
    np.savez_compressed(os.path.join(mg.OUT, "preprocess.npz"), seed=seed, S=S, time_mask=time_mask.astype(np.int64),
                        seg=EMG.numpy(), seg_dtype=str(EMG.dtype), mean=mean.numpy(), std=std.numpy(), norm=norm.numpy(),
                        butter_b=b, butter_a=a, rms_window=rc.RMS_WINDOW, window_edge=rc.WINDOW_EDGE)
    print("seg", EMG.shape, EMG.dtype, "mean", mean[:3], "std", std[:3], "time_mask head", time_mask[:16], "uniq", len(set(time_mask.tolist())))
    for f in ("/tmp/_cp_golden_emg_mean.npy", "/tmp/_cp_golden_emg_std.npy"):
        if os.path.exists(f):
            os.remove(f)


if __name__ == "__main__":
    main()
