"""CPU oracle for the raw-sEMG preprocessing -- TEST INFRASTRUCTURE ONLY.

numpy restatement of what the reference does to one (stimulus, repetition) slice of a Ninapro recording
(code/load.py:95-110) with the helpers of code/utils.py:137-156, and of its running normalisation statistics
(code/utils.py:79-135, code/load.py:116,141-148).  The signal processing itself lives in a third-party
dependency of the reference, SciPy (unpinned there; 1.15 in this image): `scipy.signal.butter`,
`scipy.signal.lfilter` and `scipy.ndimage.uniform_filter1d`.  Their published algorithms are restated here
(analog Butterworth prototype -> band-pass -> bilinear transform; direct-form-II-transposed difference equation;
running-sum recurrence with 'nearest' edge extension) so that the device kernel can be checked step by step.

Parity status: PINNED -- tests/golden/preprocess.npz is produced by the reference's own utils.filter / utils.rms /
utils.RunningStats (tools/make_golden_preprocess.py) on seeded float32 segments, and tests/test_oracle_preprocess.py
requires this restatement to reproduce it bit for bit (segments) / to 1e-6 (statistics, which the reference
accumulates sequentially in float32).

dtype note (decides the rounding points): `scipy.io.loadmat` returns Ninapro's `emg` as float32.  The reference
multiplies by 2**10 (still float32), and utils.filter writes lfilter's float64 output back INTO that float32
array row by row (utils.py:148-149), so the filtered signal is rounded to float32; np.square keeps float32;
uniform_filter1d accumulates in float64 and returns float32; np.sqrt keeps float32.
"""
from __future__ import annotations

import numpy as np

HZ = 2000                       # code/constants.py:58
FACTOR = 20                     # code/constants.py:60  (2000 Hz -> 100 Hz)
RMS_WINDOW = 11                 # code/constants.py:68
WINDOW_EDGE = (RMS_WINDOW - 1) // 2
TOTAL_WINDOW_SIZE = HZ          # code/constants.py:71
SEGMENT_LEN = TOTAL_WINDOW_SIZE + 2 * WINDOW_EDGE       # load.py:102 -> 2010 raw samples per segment
GAIN = 2.0 ** 10                # load.py:105


def time_mask() -> np.ndarray:
    """code/load.py:115: np.arange(0, 2000, 20, dtype=np.uint8) -- the uint8 wraps modulo 256, so the 100 kept
    samples all come from the first 256 of the 2000 (and only 64 of them are distinct).  Reproduced as is."""
    return (np.arange(0, TOTAL_WINDOW_SIZE, FACTOR) % 256).astype(np.int64)


def butter_bandpass(order: int = 4, low_hz: float = 20.0, high_hz: float = 450.0, fs: float = HZ):
    """scipy.signal.butter(order, [low, high]/nyquist, 'bandpass') -> (b, a), length 2*order+1 each.
    Analog prototype poles on the unit circle, lp2bp at the pre-warped band, bilinear transform (fs = 2)."""
    wn = np.array([low_hz, high_hz], dtype=np.float64) / (fs / 2.0)
    m = np.arange(-order + 1, order, 2)
    p = -np.exp(1j * np.pi * m / (2 * order))                 # buttap
    k = 1.0
    fs2 = 2.0
    warped = 2 * fs2 * np.tan(np.pi * wn / fs2)
    bw, wo = warped[1] - warped[0], np.sqrt(warped[0] * warped[1])
    # lp2bp_zpk
    p_lp = p * bw / 2
    p_bp = np.concatenate((p_lp + np.sqrt(p_lp ** 2 - wo ** 2 + 0j), p_lp - np.sqrt(p_lp ** 2 - wo ** 2 + 0j)))
    z_bp = np.zeros(order)
    k_bp = k * bw ** order
    # bilinear_zpk
    fs2x = 2.0 * fs2
    z_z = (fs2x + z_bp) / (fs2x - z_bp)
    p_z = (fs2x + p_bp) / (fs2x - p_bp)
    z_z = np.append(z_z, -np.ones(len(p_bp) - len(z_bp)))
    k_z = k_bp * np.real(np.prod(fs2x - z_bp) / np.prod(fs2x - p_bp))
    b = k_z * np.real(np.poly(z_z))
    a = np.real(np.poly(p_z))
    return b, a


def lfilter_df2t(b: np.ndarray, a: np.ndarray, x: np.ndarray) -> np.ndarray:
    """scipy.signal.lfilter(b, a, x) along axis 0 in float64 (zero initial state), the difference equation
    exactly as SciPy's C loop evaluates it: y = z0 + b0*x;  z_i = (z_{i+1} + x*b_{i+1}) - y*a_{i+1}."""
    b = np.asarray(b, np.float64) / a[0]
    a = np.asarray(a, np.float64) / a[0]
    n = len(b)
    x = np.asarray(x, np.float64)
    z = np.zeros((n - 1,) + x.shape[1:], np.float64)
    y = np.empty_like(x)
    for t in range(x.shape[0]):
        xt = x[t]
        yt = z[0] + b[0] * xt
        for i in range(n - 2):
            z[i] = (z[i + 1] + xt * b[i + 1]) - yt * a[i + 1]
        z[n - 2] = xt * b[n - 1] - yt * a[n - 1]
        y[t] = yt
    return y


def uniform_filter1d_nearest(x: np.ndarray, size: int) -> np.ndarray:
    """scipy.ndimage.uniform_filter1d(x, size, axis=0, mode='nearest'): float64 running mean over the line
    extended by size//2 copies of its end values, with SciPy's recurrence: a running SUM  tmp += in[l+size-1] - in[l-1]
    and out[l] = tmp / size."""
    half = size // 2
    ext = np.concatenate((np.repeat(x[:1], half, 0), x, np.repeat(x[-1:], size - 1 - half, 0))).astype(np.float64)
    out = np.empty(x.shape, np.float64)
    tmp = np.zeros(x.shape[1:], np.float64)
    for l in range(size):
        tmp = tmp + ext[l]
    out[0] = tmp / float(size)
    for l in range(1, x.shape[0]):
        tmp = tmp + (ext[l + size - 1] - ext[l - 1])
        out[l] = tmp / float(size)
    return out.astype(x.dtype)


def preprocess_segment(raw: np.ndarray, b=None, a=None) -> np.ndarray:
    """One segment (2010, 12) float32 -> (100, 12) float32: load.py:102-109 (get_stim_rep after the slice)."""
    if b is None:
        b, a = butter_bandpass()
    assert raw.dtype == np.float32 and raw.shape[0] == SEGMENT_LEN
    x = raw * np.float32(GAIN)                                            # float32
    y = lfilter_df2t(b, a, x).astype(np.float32)                          # utils.filter: written back into float32
    sq = np.square(y)                                                     # float32
    r = np.sqrt(uniform_filter1d_nearest(sq, RMS_WINDOW))[WINDOW_EDGE:-WINDOW_EDGE]      # utils.moving_rms
    return r[time_mask()]


def running_stats(segments: np.ndarray, complete: bool = False):
    """utils.RunningStats over segments (S,100,12): the statistic is taken over the per-segment means
    (push does X.mean(0)); mean = their average, std = their sample standard deviation (n-1).
    (The reference updates both sequentially in float32; this is the float64 closed form.)"""
    m = segments.astype(np.float64).mean(1)                                # (S, 12)
    mean = m.mean(0)
    var = ((m - mean) ** 2).sum(0) / (m.shape[0] - 1)
    if complete:
        return np.float64(mean.mean()), np.float64(np.sqrt(var.mean()))
    return mean, np.sqrt(var)
