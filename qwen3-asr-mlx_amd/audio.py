"""Audio I/O and the HIP log-mel frontend (mirror of reference audio.py).

`log_mel_spectrogram` keeps the reference signature (audio.py:238-246) and returns a
``torch.Tensor`` on the GPU instead of an ``mx.array``.  The arithmetic runs in
``q3asr_log_mel`` (csrc/mel.hip); the constant tables (Hann window, DFT twiddles, the
reference's HTK-scale / width-normalised filterbank with its four dead rows) are built here
once per parameter set and kept on the device.
"""
from __future__ import annotations

import struct
import threading
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import _lib

SAMPLE_RATE = 16_000
N_FFT = 400
HOP_LENGTH = 160
N_MELS = 128
F_MIN = 0.0
F_MAX = 8_000.0

_mel_filterbank_cache: dict[tuple, np.ndarray] = {}
_frontends: dict[tuple, "MelFrontend"] = {}
_cache_lock = threading.Lock()


def _get_mel_filterbank(n_fft: int = N_FFT, n_mels: int = N_MELS, sample_rate: int = SAMPLE_RATE,
                        f_min: float = F_MIN, f_max: float = F_MAX) -> np.ndarray:
    """(n_mels, n_fft//2+1) float32 triangles, cached per parameter tuple (audio.py:41-96).
    Mel scale 2595*log10(1+f/700); rows scaled by 1/(f_right - f_left)."""
    key = (n_fft, n_mels, sample_rate, f_min, f_max)
    fb = _mel_filterbank_cache.get(key)
    if fb is None:
        n_bins = n_fft // 2 + 1
        hz = np.linspace(0.0, sample_rate / 2.0, n_bins)
        mel_lo = 2595.0 * np.log10(1.0 + f_min / 700.0)
        mel_hi = 2595.0 * np.log10(1.0 + f_max / 700.0)
        pts = 700.0 * (10.0 ** (np.linspace(mel_lo, mel_hi, n_mels + 2) / 2595.0) - 1.0)
        fb = np.zeros((n_mels, n_bins), dtype=np.float32)
        for m in range(n_mels):
            left, centre, right = pts[m], pts[m + 1], pts[m + 2]
            rise = (hz - left) / (centre - left)
            fall = (right - hz) / (right - centre)
            fb[m] = np.maximum(0.0, np.minimum(rise, fall))
            if right - left > 0.0:
                fb[m] /= (right - left)
        _mel_filterbank_cache[key] = fb
    return fb


# --------------------------------------------------------------------------- audio files
def _read_wav_pcm(path: str | Path) -> tuple[np.ndarray, int]:
    """RIFF fast path: PCM16 / PCM32 / float32, interleaved channels averaged (audio.py:103-170)."""
    with open(path, "rb") as f:
        if f.read(4) != b"RIFF":
            raise ValueError("Not a RIFF file")
        f.read(4)
        if f.read(4) != b"WAVE":
            raise ValueError("Not WAVE")
        fmt = None
        data_at = data_len = None
        while True:
            cid = f.read(4)
            if len(cid) < 4:
                break
            (size,) = struct.unpack("<I", f.read(4))
            if cid == b"fmt ":
                fmt = f.read(size)
            else:
                if cid == b"data":
                    data_at, data_len = f.tell(), size
                f.seek(size, 1)
        if fmt is None or data_at is None:
            raise ValueError("Missing fmt or data chunk")
        tag, channels, rate = struct.unpack_from("<HHI", fmt, 0)
        bits = struct.unpack_from("<H", fmt, 14)[0]
        if tag not in (1, 3):
            raise ValueError(f"Unsupported audio format: {tag}")
        f.seek(data_at)
        raw = f.read(data_len)
    n = data_len // (bits // 8) // channels
    if tag == 1 and bits == 16:
        x = np.frombuffer(raw, dtype=np.int16).astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(raw, dtype=np.int32).astype(np.float32) / 2147483648.0
    elif tag == 3 and bits == 32:
        x = np.frombuffer(raw, dtype=np.float32).copy()
    else:
        raise ValueError(f"Unsupported PCM depth: {bits}-bit")
    if channels > 1:
        x = x[: n * channels].reshape(n, channels).mean(axis=1)
    return x.astype(np.float32), rate


def load_audio(path: str | Path, target_sr: int = SAMPLE_RATE) -> np.ndarray:
    """Mono float32 at target_sr; WAV natively, soundfile otherwise, linear-interp resample
    with target_len = int(len * target_sr / sr) (audio.py:173-204)."""
    path = Path(path)
    samples: Optional[np.ndarray] = None
    sr: Optional[int] = None
    if path.suffix.lower() == ".wav":
        try:
            samples, sr = _read_wav_pcm(path)
        except Exception:
            samples = None
    if samples is None:
        import soundfile as sf
        samples, sr = sf.read(str(path), dtype="float32", always_2d=False)
        if samples.ndim == 2:
            samples = samples.mean(axis=1)
    if sr != target_sr:
        n_out = int(len(samples) * target_sr / sr)
        samples = np.interp(np.linspace(0.0, len(samples) - 1, n_out), np.arange(len(samples)), samples).astype(np.float32)
    return samples


# --------------------------------------------------------------------------- HIP frontend
class MelFrontend:
    """Device-resident tables + batched launch of q3asr_log_mel (the reference's default n_fft / hop_length / n_mels, the
    triple the encoder is built for) or q3asr_log_mel_generic (any other triple audio.py:238-246 accepts)."""

    def __init__(self, device, sample_rate: int = SAMPLE_RATE, f_min: float = F_MIN, f_max: float = F_MAX,
                 n_fft: int = N_FFT, hop_length: int = HOP_LENGTH, n_mels: int = N_MELS):
        import torch
        if n_fft < 2 or hop_length < 1 or n_mels < 1:
            raise ValueError(f"log-mel: n_fft={n_fft} hop_length={hop_length} n_mels={n_mels}")
        self.device = torch.device(device)
        self.n_fft, self.hop, self.n_mels = int(n_fft), int(hop_length), int(n_mels)
        self.default = (self.n_fft, self.hop, self.n_mels) == (N_FFT, HOP_LENGTH, N_MELS)
        fb = _get_mel_filterbank(self.n_fft, self.n_mels, sample_rate, f_min, f_max)
        rng = np.zeros((self.n_mels, 2), dtype=np.int32)
        for m in range(self.n_mels):
            nz = np.nonzero(fb[m])[0]
            if len(nz):
                rng[m] = (nz[0], nz[-1] + 1)
        ang = 2.0 * np.pi * np.arange(self.n_fft, dtype=np.float64) / self.n_fft
        tw = np.stack([np.cos(ang), np.sin(ang)], axis=1)
        self.fb = torch.from_numpy(fb).to(self.device)
        self.fb_range = torch.from_numpy(rng).to(self.device)
        self.window = torch.from_numpy(np.hanning(self.n_fft).astype(np.float32)).to(self.device)
        self.twiddle = torch.from_numpy(tw).to(self.device)

    def frames(self, n_samples: int) -> int:
        """1 + (len + 2 (n_fft // 2) - n_fft) // hop frames, the last one dropped (audio.py:224-226,266)."""
        padded = int(n_samples) + 2 * (self.n_fft // 2)
        return max(0, (padded - self.n_fft) // self.hop)

    def __call__(self, audio, lens, lens_host: Sequence[int]):
        """audio fp32 [B][stride] (device, zero padded), lens int32 [B] (device).
        Returns (mel fp32 [B][n_mels][Fmax], n_frames list)."""
        import torch
        B = audio.shape[0]
        n_frames = [self.frames(n) for n in lens_host]
        if min(n_frames) < 1:
            raise ValueError(f"audio shorter than one hop ({self.hop} samples) has no mel frames")
        fmax = max(n_frames)
        out = torch.empty((B, self.n_mels, fmax), dtype=torch.float32, device=self.device)
        gkey = torch.empty((B,), dtype=torch.int32, device=self.device)
        if self.default:
            _lib.call("q3asr_log_mel", _lib.ptr(audio), audio.stride(0), _lib.ptr(lens), B, fmax, _lib.ptr(self.window),
                      _lib.ptr(self.twiddle), _lib.ptr(self.fb), _lib.ptr(self.fb_range), _lib.ptr(out), _lib.ptr(gkey),
                      _lib.stream_ptr())
        else:
            _lib.call("q3asr_log_mel_generic", _lib.ptr(audio), audio.stride(0), _lib.ptr(lens), B, fmax, self.n_fft, self.hop,
                      self.n_mels, _lib.ptr(self.window), _lib.ptr(self.twiddle), _lib.ptr(self.fb), _lib.ptr(out), _lib.ptr(gkey),
                      _lib.stream_ptr())
        return out, n_frames


def get_frontend(device=None, sample_rate: int = SAMPLE_RATE, f_min: float = F_MIN, f_max: float = F_MAX,
                 n_fft: int = N_FFT, hop_length: int = HOP_LENGTH, n_mels: int = N_MELS) -> MelFrontend:
    import torch
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    key = (str(dev), sample_rate, f_min, f_max, n_fft, hop_length, n_mels)
    with _cache_lock:
        fe = _frontends.get(key)
        if fe is None:
            fe = _frontends[key] = MelFrontend(dev, sample_rate, f_min, f_max, n_fft, hop_length, n_mels)
    return fe


# Pinned staging buffers for the H2D copy of a request, pooled per (device, lane): a lane is one request stream
# (decoder.set_lane), so its buffer is reused from call to call whichever short-lived worker thread serves it; the event of
# the last copy that read a buffer guards its reuse.  release_staging() (Qwen3ASR.close) gives the pinned memory back.
_staging: dict = {}
_staging_lock = threading.Lock()
_STAGING_MIN = 1 << 20            # floats


def release_staging() -> None:
    with _staging_lock:
        _staging.clear()


# How a request's rows travel to the device.  "single": fill the whole pinned buffer, then ONE copy.  "grouped": rows in groups
# of ~8 MB, the copy of one group issued while the host fills the next (round 4).  bench.py measures both in one process
# (`end_to_end.h2d`): the round-4 driver run showed the end-to-end figure 31 ms above the resident one with the grouped form
# where round 3 (single copy) showed 3 ms; PACK_MODE is whatever that A/B keeps.
PACK_MODE = "single"
PACK_STATS: Optional[dict] = None      # set to a dict to have the next pack_audio call record its host times and copy events

# The staging fill runs on a small thread pool of its own, never on torch's intra-op pool.  Round 5 found the cause of the
# 20-35 ms `end_to_end` spikes (and of the round-4 driver's +31 ms): `tensor[...] = from_numpy(row)` parallelises a 2 MB row copy over
# torch's default thread count -- 128 on the GPU boxes, whose container is allowed 16 CPUs (cgroup cpu.max 1600000 100000) -- and 128
# spinning threads exhaust the container's CPU quota, so the kernel parks the whole process for the rest of the 100 ms period
# (cpu.stat: nr_throttled, throttled_usec).  Rows are copied with numpy (which releases the GIL) by at most 8 workers, fewer if
# the affinity mask or the cgroup quota is smaller.
_fill_pool = None
_fill_pool_lock = threading.Lock()


def _fill_workers() -> int:
    import os
    n = 8
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        n = min(n, os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _fill_rows(host_np, batch, lens, lmax, i0: int, i1: int) -> None:
    """Rows i0..i1-1 of the request into the pinned buffer (a numpy view of it), zero tails included."""
    global _fill_pool
    with _fill_pool_lock:
        if _fill_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            n = _fill_workers()
            _fill_pool = (ThreadPoolExecutor(max_workers=n, thread_name_prefix="q3asr-fill"), n)
        pool, workers = _fill_pool

    def rows(a: int, b: int) -> None:
        for i in range(a, b):
            np.copyto(host_np[i, : lens[i]], batch[i], casting="unsafe")
            if lens[i] < lmax:
                host_np[i, lens[i]:] = 0.0

    w = min(workers, i1 - i0)
    if w <= 1:
        rows(i0, i1)
        return
    step = (i1 - i0 + w - 1) // w
    for f in [pool.submit(rows, a, min(a + step, i1)) for a in range(i0, i1, step)]:
        f.result()


def pack_audio(batch: Sequence[np.ndarray], device, mode: Optional[str] = None):
    """Host list of 1-D float32 arrays -> (padded device tensor [B][Lmax], lens device, lens host).
    On a GPU device the rows are staged in a pinned buffer kept per (device, lane) (allocating 61 MB of pinned memory per
    32 x 30 s request cost 4-25 ms of a 250 ms request); the buffer is reused only after the copy that last read it has
    completed.  A CPU device gets a plain tensor of its own.  `mode`: PACK_MODE above."""
    import time
    import torch
    lens = [int(len(a)) for a in batch]
    lmax = max(lens)
    n = len(batch) * lmax
    dev_t = torch.device(device)
    if dev_t.type != "cuda":
        host = torch.zeros((len(batch), lmax), dtype=torch.float32)
        for i, a in enumerate(batch):
            host[i, : lens[i]] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        return host.to(dev_t), torch.tensor(lens, dtype=torch.int32).to(dev_t), lens
    from .decoder import current_lane
    mode = mode or PACK_MODE
    stats = PACK_STATS
    t0 = time.perf_counter()
    key = (dev_t.index if dev_t.index is not None else torch.cuda.current_device(), current_lane())
    with _staging_lock:
        st = _staging.pop(key, None)                       # taken out while in use: a second thread on the same lane gets its own
    if st is None or st[0].numel() < n:
        st = (torch.empty((max(n, _STAGING_MIN),), dtype=torch.float32, pin_memory=True), None)
    flat, ev = st
    if ev is not None:
        ev.synchronize()                                   # the previous request's H2D copy has read the buffer
    host = flat[:n].view(len(batch), lmax)
    host_np = host.numpy()                                 # the same pinned bytes
    dev = torch.empty((len(batch), lmax), dtype=torch.float32, device=dev_t)
    t1 = time.perf_counter()
    stream = torch.cuda.current_stream(dev.device)
    e0 = e1 = None
    if stats is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_fill = t_issue = 0.0
    group = len(batch) if mode == "single" else max(1, (8 << 20) // max(4 * lmax, 1))
    first = True
    for i0 in range(0, len(batch), group):
        i1 = min(i0 + group, len(batch))
        f0 = time.perf_counter()
        _fill_rows(host_np, batch, lens, lmax, i0, i1)
        f1 = time.perf_counter()
        if first and e0 is not None:
            e0.record(stream)
        first = False
        dev[i0:i1].copy_(host[i0:i1], non_blocking=True)
        t_fill += f1 - f0
        t_issue += time.perf_counter() - f1
    if e1 is not None:
        e1.record(stream)
    lens_dev = torch.tensor(lens, dtype=torch.int32).to(dev_t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(stream)
    with _staging_lock:
        _staging[key] = (flat, ev)
    if stats is not None:
        stats.update({"mode": mode, "bytes": 4 * n, "groups": (len(batch) + group - 1) // group, "pinned": bool(flat.is_pinned()),
                      "setup_ms": 1e3 * (t1 - t0), "host_fill_ms": 1e3 * t_fill, "copy_issue_ms": 1e3 * t_issue,
                      "host_total_ms": 1e3 * (time.perf_counter() - t0), "_events": (e0, e1)})
    return dev, lens_dev, lens


def pack_stats_finish(stats: dict) -> dict:
    """Resolve the copy events of a PACK_STATS record (after the stream has been synchronised): first copy issued -> last copy
    done on the device timeline (with the grouped form that span contains the host fills between the copies)."""
    e0, e1 = stats.pop("_events", (None, None))
    if e0 is not None:
        e1.synchronize()
        stats["copy_span_ms"] = float(e0.elapsed_time(e1))
        stats["copy_GBps_over_span"] = stats["bytes"] / (stats["copy_span_ms"] * 1e-3) / 1e9 if stats["copy_span_ms"] > 0 else None
    return stats


def log_mel_spectrogram(audio, n_fft: int = N_FFT, hop_length: int = HOP_LENGTH, n_mels: int = N_MELS,
                        sample_rate: int = SAMPLE_RATE, f_min: float = F_MIN, f_max: float = F_MAX):
    """Log-mel spectrogram of one utterance, shape (n_mels, n_frames), float32 on the GPU.
    Same steps as reference audio.py:238-278, for any n_fft / hop_length / n_mels it accepts (the defaults run the
    specialised kernel the encoder path uses, other triples q3asr_log_mel_generic)."""
    if not isinstance(audio, np.ndarray):
        audio = load_audio(audio, target_sr=sample_rate)
    fe = get_frontend(None, sample_rate, f_min, f_max, n_fft, hop_length, n_mels)
    dev, lens_dev, lens = pack_audio([np.asarray(audio, dtype=np.float32)], fe.device)
    mel, _ = fe(dev, lens_dev, lens)
    return mel[0]
