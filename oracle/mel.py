"""Oracle: 128-bin log-mel frontend (numpy, float32) -- test infrastructure.

Restates reference audio.py:31-96 (filterbank) and audio.py:211-278 (STFT +
log-mel).  Pinned bit-for-bit / to 2e-6 against tests/golden (generated from
the live reference by tools/gen_golden.py).
"""
from __future__ import annotations

import functools

import numpy as np

SAMPLE_RATE = 16_000
N_FFT = 400
HOP = 160
N_MELS = 128


@functools.lru_cache(maxsize=8)
def mel_filterbank(n_fft: int = N_FFT, n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE,
                   f_min: float = 0.0, f_max: float = 8000.0) -> np.ndarray:
    """audio.py:41-80.  HTK mel scale 2595*log10(1+f/700) (audio.py:31-38 --
    the reference's comment says Slaney, the formula is HTK), triangles on
    linspace(0, sr/2, n_fft/2+1), each row divided by (f_right - f_left)
    (audio.py:76-78).  float64 math stored to float32 row by row."""
    n_freqs = n_fft // 2 + 1
    bins = np.linspace(0.0, sample_rate / 2.0, n_freqs)
    to_mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)          # noqa: E731
    to_hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)         # noqa: E731
    edges = to_hz(np.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2))
    fb = np.zeros((n_mels, n_freqs), dtype=np.float32)
    for m in range(n_mels):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        tri = np.maximum(0.0, np.minimum((bins - lo) / (ce - lo), (hi - bins) / (hi - ce)))
        fb[m] = tri                      # float64 -> float32 store (audio.py:73)
        if hi - lo > 0.0:
            fb[m] /= (hi - lo)           # float32 array / float64 scalar (audio.py:78)
    return fb


def stft_power(x: np.ndarray, n_fft: int = N_FFT, hop: int = HOP) -> np.ndarray:
    """audio.py:211-235 then audio.py:266: reflect-pad n_fft/2, symmetric
    Hann (np.hanning), frames 1+(len-n_fft)//hop, rfft per frame, LAST frame
    dropped, |.|**2.  Returns float32 (n_fft/2+1, n_frames-1)."""
    x = np.asarray(x, dtype=np.float32)
    w = np.hanning(n_fft).astype(np.float32)
    xp = np.pad(x, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(xp) - n_fft) // hop
    idx = (np.arange(n_frames)[:, None] * hop) + np.arange(n_fft)[None, :]
    frames = xp[idx] * w[None, :]                       # float32
    spec = np.fft.rfft(frames, n=n_fft, axis=1).astype(np.complex64)   # numpy>=2 keeps c64
    spec = spec[:-1]                                    # drop last frame (audio.py:266)
    return (np.abs(spec) ** 2).T.astype(np.float32)     # (n_freqs, n_frames-1)


def log_mel_spectrogram(x: np.ndarray, n_fft: int = N_FFT, hop: int = HOP, n_mels: int = N_MELS,
                        sample_rate: int = SAMPLE_RATE, f_min: float = 0.0,
                        f_max: float = 8000.0) -> np.ndarray:
    """audio.py:238-278.  float32 (n_mels, n_frames)."""
    p = stft_power(x, n_fft, hop)
    mel = mel_filterbank(n_fft, n_mels, sample_rate, f_min, f_max) @ p
    ls = np.log10(np.maximum(mel, 1e-10))
    ls = np.maximum(ls, ls.max() - 8.0)
    return ((ls + 4.0) / 4.0).astype(np.float32)
