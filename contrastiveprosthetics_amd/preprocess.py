"""Raw Ninapro sEMG -> the `emg.pt` tensor of the reference, on the GPU.

Mirrors `DB23.load_dataset` / `DB23.get_stim_rep` (code/load.py:85-155) and the helpers `filter`, `rms`,
`RunningStats` (code/utils.py:79-156): for every (person, stimulus, repetition) the first 2010 samples of the
recording where `restimulus == stimulus` and `rerepetition == repetition` are amplified by 2**10, band-passed
(4th-order Butterworth, 20-450 Hz, `scipy.signal.lfilter`), reduced to a moving RMS over 11 samples and sampled
at `time_mask`; the training slices give the normalisation statistics; the normalised tensor
(people, 41, 6, 100, 12) is what `torch.save(..., 'data/emg.pt')` stores.

Here the slicing stays on the host (index arithmetic on the label columns) and everything per sample runs in
three kernels over ALL segments at once (`cp_preprocess_emg`, `cp_emg_stats`, `cp_emg_normalize`).  The filter
design is restated in numpy (no SciPy needed at run time).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .constants import EMG_DIM, FACTOR, Hz, MAX_REPS, MAX_TASKS, RMS_WINDOW, TASK_DIST, TOTAL_WINDOW_SIZE, WINDOW_EDGE

SEGMENT_LEN = TOTAL_WINDOW_SIZE + 2 * WINDOW_EDGE          # code/load.py:102
GAIN = 2.0 ** 10                                           # code/load.py:105


def time_mask() -> np.ndarray:
    """code/load.py:115 -- `np.arange(0, 2000, 20, dtype=np.uint8)`: the uint8 wraps modulo 256 (reference quirk,
    kept: the stored data set was made this way, so the 100 samples of a repetition cover its first 128 ms)."""
    return np.arange(0, TOTAL_WINDOW_SIZE, FACTOR).astype(np.uint8).astype(np.int32)


def butter_bandpass(order: int = 4, low_hz: float = 20.0, high_hz: float = 450.0, fs: float = Hz):
    """(b, a) of `scipy.signal.butter(order, (low, high) / nyquist, btype="bandpass")` (code/utils.py:137-146):
    Butterworth analog prototype, low-pass -> band-pass at the pre-warped edges, bilinear transform."""
    wn = np.array([low_hz, high_hz], dtype=np.float64) / (fs / 2.0)
    p = -np.exp(1j * np.pi * np.arange(-order + 1, order, 2) / (2 * order))
    fs2 = 2.0
    warped = 2 * fs2 * np.tan(np.pi * wn / fs2)
    bw, wo = warped[1] - warped[0], np.sqrt(warped[0] * warped[1])
    p_lp = p * bw / 2
    root = np.sqrt(p_lp ** 2 - wo ** 2 + 0j)
    p_bp = np.concatenate((p_lp + root, p_lp - root))
    z_bp = np.zeros(order)
    k_bp = bw ** order
    fs2x = 2.0 * fs2
    z_z = np.append((fs2x + z_bp) / (fs2x - z_bp), -np.ones(len(p_bp) - len(z_bp)))
    p_z = (fs2x + p_bp) / (fs2x - p_bp)
    k_z = k_bp * np.real(np.prod(fs2x - z_bp) / np.prod(fs2x - p_bp))
    return k_z * np.real(np.poly(z_z)), np.real(np.poly(p_z))


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(t: torch.Tensor):
    if t.device.type != "cuda":
        raise _lib.CpNativeError("contrastiveprosthetics_amd runs on an MI355X (device 'cuda') only; no CPU path")


def preprocess_segments(raw: torch.Tensor, b=None, a=None, keep=None, rms_window: int = RMS_WINDOW,
                        gain: float = GAIN) -> torch.Tensor:
    """raw (S, seg_len, 12) f32 on the GPU -> (S, len(keep), 12) f32: filter(emg*2**10) -> rms -> [time_mask]."""
    lib = _lib.load()
    _need_gpu(raw)
    assert raw.dtype == torch.float32 and raw.is_contiguous() and raw.dim() == 3 and raw.shape[2] == EMG_DIM
    if b is None:
        b, a = butter_bandpass()
    keep = time_mask() if keep is None else np.asarray(keep, dtype=np.int32)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    aa = np.ascontiguousarray(a, dtype=np.float64)
    kk = np.ascontiguousarray(keep, dtype=np.int32)
    out = torch.empty(raw.shape[0], len(kk), EMG_DIM, dtype=torch.float32, device=raw.device)
    _lib.check(lib.cp_preprocess_emg(raw.data_ptr(), raw.shape[0], raw.shape[1], bb.ctypes.data_as(C.POINTER(C.c_double)),
                                     aa.ctypes.data_as(C.POINTER(C.c_double)), len(bb), int(rms_window), float(gain),
                                     kk.ctypes.data_as(C.POINTER(C.c_int32)), len(kk), out.data_ptr(), _stream(raw)),
               "cp_preprocess_emg")
    return out


def emg_stats(seg: torch.Tensor, use: torch.Tensor = None, complete: bool = False):
    """RunningStats(...).mean_std() over the segments with use != 0: (mean, std), each (12,) f32 on the GPU."""
    lib = _lib.load()
    _need_gpu(seg)
    assert seg.dtype == torch.float32 and seg.is_contiguous() and seg.shape[-1] == EMG_DIM
    S, n_out = int(np.prod(seg.shape[:-2])), seg.shape[-2]
    if use is not None:
        use = use.to(device=seg.device, dtype=torch.uint8).reshape(-1).contiguous()
        assert use.numel() == S
    scratch = torch.empty(S * EMG_DIM, dtype=torch.float64, device=seg.device)
    ms = torch.empty(2, EMG_DIM, dtype=torch.float32, device=seg.device)
    _lib.check(lib.cp_emg_stats(seg.data_ptr(), S, n_out, use.data_ptr() if use is not None else None, int(complete),
                                scratch.data_ptr(), ms.data_ptr(), _stream(seg)), "cp_emg_stats")
    return ms[0], ms[1]


def normalize_(seg: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """RunningStats.normalize in place: (seg - mean) / std."""
    lib = _lib.load()
    _need_gpu(seg)
    assert seg.dtype == torch.float32 and seg.is_contiguous() and seg.shape[-1] == EMG_DIM
    ms = torch.stack((mean.reshape(-1).expand(EMG_DIM) if mean.numel() == 1 else mean.reshape(EMG_DIM),
                      std.reshape(-1).expand(EMG_DIM) if std.numel() == 1 else std.reshape(EMG_DIM))).to(torch.float32).contiguous()
    _lib.check(lib.cp_emg_normalize(seg.data_ptr(), seg.numel() // EMG_DIM, ms.data_ptr(), _stream(seg)), "cp_emg_normalize")
    return seg


def slice_recording(emg: np.ndarray, restimulus: np.ndarray, rerepetition: np.ndarray, stimulus: int, repetition: int):
    """code/load.py:95-102: the first SEGMENT_LEN samples where both label columns match (host index arithmetic)."""
    idx = np.flatnonzero((restimulus.reshape(-1) == stimulus) & (rerepetition.reshape(-1) == repetition))[:SEGMENT_LEN]
    if idx.size < SEGMENT_LEN:
        raise ValueError(f"stimulus {stimulus} repetition {repetition}: only {idx.size} of {SEGMENT_LEN} samples")
    return emg[idx]


def build_emg_tensor(recordings, train_people=None, train_reps=None, complete: bool = False, device="cuda"):
    """`DB23.load_dataset` (code/load.py:112-151) for `recordings`: one entry per person, each a pair
    ((emg, restimulus, rerepetition) of exercise B, the same of exercise C) as `get_np` returns them.
    Stimulus ids follow the reference: 0 = rest (taken from exercise B), 1..17 exercise B, 18..40 exercise C
    with the recording's own numbering (np.searchsorted(TASK_DIST.cumsum(), stimulus) picks the exercise).
    Returns (EMG (people, 41, 6, 100, 12) normalised f32 on the GPU, mean, std)."""
    P = len(recordings)
    raw = np.empty((P, MAX_TASKS, MAX_REPS, SEGMENT_LEN, EMG_DIM), dtype=np.float32)
    cum = TASK_DIST.cumsum()
    for i, person in enumerate(recordings):
        for rep in range(MAX_REPS):
            for stim in range(MAX_TASKS):
                emg, st, rp = person[int(np.searchsorted(cum, stim))]
                raw[i, stim, rep] = slice_recording(emg, st, rp, stim, rep + 1)
    dev_raw = torch.from_numpy(raw.reshape(-1, SEGMENT_LEN, EMG_DIM)).to(device)
    seg = preprocess_segments(dev_raw)
    use = np.zeros((P, MAX_TASKS, MAX_REPS), dtype=np.uint8)
    people = range(P) if train_people is None else train_people
    reps = range(MAX_REPS) if train_reps is None else train_reps
    use[np.ix_(list(people), list(range(MAX_TASKS)), list(reps))] = 1         # load.py:140 (every task is a training task)
    mean, std = emg_stats(seg, torch.from_numpy(use.reshape(-1)), complete=complete)
    normalize_(seg, mean, std)
    return seg.reshape(P, MAX_TASKS, MAX_REPS, seg.shape[1], EMG_DIM), mean, std
