#!/usr/bin/env python3
"""Golden log-mel arrays for NON-DEFAULT (n_fft, hop_length, n_mels, sample_rate, f_min, f_max) from the LIVE reference
(audio.py:238-278, loaded exactly as tools/gen_golden.py loads it).  Build container only; writes
tests/golden/mel_params.npz + mel_params.json (data only: the inputs are the seeded signals of tools/gen_golden.py)."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from gen_golden import OUT, _load_ref_audio, signal

CASES = {
    "fft512_hop128_mels80": dict(kind="noise", n=16000, n_fft=512, hop_length=128, n_mels=80),
    "fft256_hop100_mels40_sr8k": dict(kind="speechlike", n=8000, n_fft=256, hop_length=100, n_mels=40, sample_rate=8000, f_min=50.0, f_max=3800.0),
    "fft400_hop200": dict(kind="noise", n=1234, n_fft=400, hop_length=200, n_mels=128),
    "fft1024_hop256_mels64": dict(kind="tone", n=16000, n_fft=1024, hop_length=256, n_mels=64),
    "fft201_hop67_mels20_odd": dict(kind="noise", n=5000, n_fft=201, hop_length=67, n_mels=20),
    "fft400_hop160_mels128_fmax4k": dict(kind="speechlike", n=16000, n_fft=400, hop_length=160, n_mels=128, f_max=4000.0),
    "fft64_hop16_mels8_short": dict(kind="noise", n=40, n_fft=64, hop_length=16, n_mels=8),      # 2 frames: pad 32 < 40 samples
}


def main() -> None:
    audio = _load_ref_audio()
    arrs, meta = {}, {}
    for name, c in CASES.items():
        c = dict(c)
        x = signal(c.pop("kind"), c.pop("n"))
        try:
            mel = np.asarray(audio.log_mel_spectrogram(x, **c))
        except Exception as e:                              # noqa: BLE001
            meta[name] = {"params": c, "raises": type(e).__name__}
            continue
        assert mel.dtype == np.float32
        arrs[name] = mel
        meta[name] = {"params": c, "shape": list(mel.shape)}
    np.savez_compressed(OUT / "mel_params.npz", **arrs)
    (OUT / "mel_params.json").write_text(json.dumps(meta, indent=1))
    for k, v in meta.items():
        print(k, v)


if __name__ == "__main__":
    main()
