"""oracle/preprocess_cpu.py against tests/golden/preprocess.npz (the reference's utils.filter / utils.rms /
utils.RunningStats run on seeded float32 segments, tools/make_golden_preprocess.py)."""
import os
import sys

import numpy as np

from oracle import preprocess_cpu as pp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def raw_segments(seed, S):
    """same generator as tools/make_golden_preprocess.py (kept here so the GPU box needs no tools import)"""
    rng = np.random.default_rng(seed)
    L, C = 2010, 12
    t = np.arange(L)[None, :, None] / 2000.0
    f = rng.uniform(5, 900, size=(S, 1, C))
    x = 2e-5 * rng.standard_normal((S, L, C)) + 1e-5 * np.sin(2 * np.pi * f * t) + 3e-6 * rng.standard_normal((S, 1, C))
    return x.astype(np.float32)


def test_time_mask_quirk(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    tm = pp.time_mask()
    assert np.array_equal(tm, g["time_mask"])
    assert tm.max() < 256 and len(set(tm.tolist())) == 64 and tm[13] == 4          # 260 wrapped to 4
    assert (int(g["rms_window"]), int(g["window_edge"])) == (pp.RMS_WINDOW, pp.WINDOW_EDGE)


def test_butterworth_design(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    b, a = pp.butter_bandpass()
    np.testing.assert_allclose(b, g["butter_b"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(a, g["butter_a"], rtol=1e-9, atol=1e-15)


def test_segments_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    raw = raw_segments(int(g["seed"]), int(g["S"]))
    # the filter coefficients SciPy produced: the comparison is about the recurrences and rounding points
    seg = np.stack([pp.preprocess_segment(r, g["butter_b"], g["butter_a"]) for r in raw])
    assert seg.dtype == np.float32 and seg.shape == (6, 100, 12)
    assert np.array_equal(seg, g["seg"])
    # with the oracle's own design the result may move by an ulp at most
    seg2 = np.stack([pp.preprocess_segment(r) for r in raw])
    np.testing.assert_allclose(seg2, g["seg"], rtol=3e-7)


def test_running_stats(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    mean, std = pp.running_stats(g["seg"])
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-6)
    np.testing.assert_allclose(std, g["std"], rtol=1e-5)
    np.testing.assert_allclose((g["seg"] - mean) / std, g["norm"], rtol=2e-4, atol=2e-4)
