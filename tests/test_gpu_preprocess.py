"""SURVEY.md 8f row f3 on the GPU: raw sEMG -> preprocessed, normalised segments through the C ABI, against
the fixture produced by the reference's own utils.filter / utils.rms / utils.RunningStats and against the numpy
oracle on synthetic recordings."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_cpu as pp
from test_oracle_preprocess import raw_segments

pytestmark = pytest.mark.gpu


def test_segments_bit_exact_vs_reference_fixture(golden_dir):
    from contrastiveprosthetics_amd import preprocess as P
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    raw = torch.from_numpy(raw_segments(int(g["seed"]), int(g["S"]))).cuda()
    seg = P.preprocess_segments(raw, g["butter_b"], g["butter_a"])          # SciPy's coefficients: isolates the kernel
    assert seg.dtype == torch.float32 and tuple(seg.shape) == (6, 100, 12)
    assert np.array_equal(seg.cpu().numpy(), g["seg"])                      # integer-grade parity on a float pipeline
    seg2 = P.preprocess_segments(raw)                                       # own filter design
    np.testing.assert_allclose(seg2.cpu().numpy(), g["seg"], rtol=3e-7)
    mean, std = P.emg_stats(seg)
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=1e-6)    # reference: sequential float32 Welford
    np.testing.assert_allclose(std.cpu().numpy(), g["std"], rtol=1e-5)
    P.normalize_(seg, mean, std)
    np.testing.assert_allclose(seg.cpu().numpy(), g["norm"], rtol=2e-4, atol=2e-4)


def test_many_segments_other_masks_and_lengths():
    from contrastiveprosthetics_amd import preprocess as P
    rng = np.random.default_rng(3)
    raw = raw_segments(11, 40)
    b, a = pp.butter_bandpass()
    ref = np.stack([pp.preprocess_segment(r, b, a) for r in raw])
    got = P.preprocess_segments(torch.from_numpy(raw).cuda(), b, a).cpu().numpy()
    assert np.array_equal(got, ref)
    # an un-wrapped mask (what the reference presumably intended) and the extreme positions of the RMS series
    keep = np.concatenate((np.arange(0, 2000, 20), [1999, 0, 1999]))
    full = np.stack([np.sqrt(pp.uniform_filter1d_nearest(np.square(pp.lfilter_df2t(b, a, r * np.float32(1024)).astype(np.float32)), 11))[5:-5]
                     for r in raw[:5]])
    got = P.preprocess_segments(torch.from_numpy(raw[:5]).cuda(), b, a, keep=keep).cpu().numpy()
    assert np.array_equal(got, full[:, keep])
    with pytest.raises(Exception):
        P.preprocess_segments(torch.from_numpy(raw[:1]).cuda(), b, a, keep=[2000])          # outside the RMS series
    # statistics over a subset, per channel and "complete"
    seg = torch.from_numpy(ref).cuda()
    use = torch.from_numpy((rng.random(40) < 0.6).astype(np.uint8))
    mean, std = P.emg_stats(seg, use)
    m_ref, s_ref = pp.running_stats(ref[use.numpy().astype(bool)])
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref, rtol=1e-6)
    np.testing.assert_allclose(std.cpu().numpy(), s_ref, rtol=1e-6)
    mean_c, std_c = P.emg_stats(seg, use, complete=True)
    mc, sc = pp.running_stats(ref[use.numpy().astype(bool)], complete=True)
    np.testing.assert_allclose(mean_c.cpu().numpy(), np.full(12, mc), rtol=1e-6)
    np.testing.assert_allclose(std_c.cpu().numpy(), np.full(12, sc), rtol=1e-6)


def synthetic_recording(rng, stimuli, reps=6, seg=2300, gap=150):
    """A fake exercise file: runs of `seg` samples per (stimulus, repetition) separated by rest, as restimulus /
    rerepetition label them; float32 emg."""
    emg, st, rp = [], [], []
    for r in range(1, reps + 1):
        for s in stimuli:
            n = seg + int(rng.integers(0, 50))
            emg.append(rng.standard_normal((n, 12)).astype(np.float32) * 2e-5 * (1 + 0.1 * s))
            st.append(np.full(n, s)); rp.append(np.full(n, r))
            emg.append(rng.standard_normal((gap, 12)).astype(np.float32) * 5e-6)            # rest between movements
            st.append(np.zeros(gap, dtype=int)); rp.append(np.full(gap, r))
        # enough labelled rest for the "stimulus 0" slices
        emg.append(rng.standard_normal((2100, 12)).astype(np.float32) * 5e-6)
        st.append(np.zeros(2100, dtype=int)); rp.append(np.full(2100, r))
    return np.concatenate(emg), np.concatenate(st).reshape(-1, 1), np.concatenate(rp).reshape(-1, 1)


def test_build_emg_tensor_matches_oracle_pipeline():
    from contrastiveprosthetics_amd import preprocess as P
    rng = np.random.default_rng(5)
    recordings = [(synthetic_recording(rng, range(1, 18)), synthetic_recording(rng, range(18, 41))) for _ in range(2)]
    EMG, mean, std = P.build_emg_tensor(recordings, train_people=[0, 1], train_reps=[0, 2, 3, 5])
    assert tuple(EMG.shape) == (2, 41, 6, 100, 12) and EMG.dtype == torch.float32
    b, a = pp.butter_bandpass()
    ref = np.empty((2, 41, 6, 100, 12), dtype=np.float32)
    for i, person in enumerate(recordings):
        for rep in range(6):
            for stim in range(41):
                emg, st, rp = person[0 if stim <= 17 else 1]
                idx = np.flatnonzero((st.reshape(-1) == stim) & (rp.reshape(-1) == rep + 1))[:2010]
                ref[i, stim, rep] = pp.preprocess_segment(emg[idx], b, a)
    m_ref, s_ref = pp.running_stats(ref[:, :, [0, 2, 3, 5]].reshape(-1, 100, 12))
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref, rtol=1e-6)
    np.testing.assert_allclose(std.cpu().numpy(), s_ref, rtol=1e-6)
    norm_ref = ((ref - m_ref.astype(np.float32)) / s_ref.astype(np.float32)).astype(np.float32)
    np.testing.assert_allclose(EMG.cpu().numpy(), norm_ref, rtol=1e-5, atol=1e-5)


def test_other_filter_orders_and_windows_generic_kernel():
    """anything but the reference's 9 coefficients / 11-sample window takes the general kernel"""
    from contrastiveprosthetics_amd import preprocess as P
    raw = raw_segments(21, 7)
    for order, win in ((2, 7), (3, 11), (4, 5)):
        b, a = pp.butter_bandpass(order=order, low_hz=30.0, high_hz=400.0)
        half = win // 2
        L = 600 + 2 * half
        keep = np.arange(0, 600, 7)
        ref = np.stack([np.sqrt(pp.uniform_filter1d_nearest(np.square(pp.lfilter_df2t(b, a, r[:L] * np.float32(1024)).astype(np.float32)), win))[half:L - half][keep]
                        for r in raw])
        got = P.preprocess_segments(torch.from_numpy(np.ascontiguousarray(raw[:, :L])).cuda(), b, a, keep=keep, rms_window=win)
        assert np.array_equal(got.cpu().numpy(), ref), (order, win)
