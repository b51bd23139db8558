#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ from the LIVE reference code.

Runs only in the build container (needs /root/reference); the GPU box never
runs this.  What is executed from the reference:

* ``audio.py`` (pure numpy; its only ``mlx`` use is the final
  ``mx.array(log_spec)`` wrap at audio.py:278) -- loaded by file path with
  ``sys.modules['mlx.core']`` bound to a namespace whose ``array`` is
  ``numpy.asarray``.
* ``tokenizer.py`` (no mlx import) -- loaded by file path, ``build_prompt``.
* ``model._find_split_points`` (pure numpy, model.py:454-513) -- the single
  function is extracted from the module's AST and exec'd, because the module
  imports mlx-backed classes at import time.

Outputs are data only (inputs are re-derivable from the seeds recorded here).
"""
from __future__ import annotations

import ast
import hashlib
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference/src/qwen3_asr_mlx")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _load_ref_audio():
    mlx = types.ModuleType("mlx")
    core = types.ModuleType("mlx.core")
    core.array = np.asarray
    mlx.core = core
    sys.modules["mlx"] = mlx
    sys.modules["mlx.core"] = core
    spec = importlib.util.spec_from_file_location("_ref_audio", REF / "audio.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    del sys.modules["mlx"], sys.modules["mlx.core"]
    return mod


def _load_ref_tokenizer():
    spec = importlib.util.spec_from_file_location("_ref_tok", REF / "tokenizer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load_ref_split():
    tree = ast.parse((REF / "model.py").read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_find_split_points")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "model.py", "exec"), ns)
    return ns["_find_split_points"]


def signal(kind: str, n: int) -> np.ndarray:
    """Seeded test signals (also re-created by tests/ and the oracle)."""
    if kind == "silence":
        return np.zeros(n, dtype=np.float32)
    if kind == "tone":
        t = np.arange(n, dtype=np.float64) / 16000.0
        return (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    if kind == "noise":
        return (np.random.default_rng(0).standard_normal(n) * 0.1).astype(np.float32)
    if kind == "speechlike":
        # bench.py's synthetic input: noise under a 3 Hz raised cosine with gaps
        rng = np.random.default_rng(1234)
        t = np.arange(n, dtype=np.float64) / 16000.0
        env = 0.5 * (1.0 - np.cos(2 * np.pi * 3.0 * t))
        gap = ((t % 6.0) > 5.6).astype(np.float64)
        x = 0.1 * rng.standard_normal(n) * env * (1.0 - 0.999 * gap)
        return x.astype(np.float32)
    raise ValueError(kind)


def wav_bytes(tag: int, bits: int, channels: int, rate: int, payload: bytes, extra_chunk: bytes = b"", odd_pad: bool = False,
              fmt_extra: bytes = b"") -> bytes:
    """A RIFF/WAVE image: fmt chunk (optionally with extension bytes), an optional foreign chunk BEFORE the data, data."""
    import struct
    fmt = struct.pack("<HHIIHH", tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits) + fmt_extra
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra_chunk + b"data" + struct.pack("<I", len(payload)) + payload
    if odd_pad:
        body += b"\x00"
    return b"RIFF" + struct.pack("<I", len(body)) + body


def wav_cases() -> dict:
    """name -> (wav bytes, target_sr): PCM16 stereo, PCM32, float32, a foreign chunk, 8 k / 44.1 k / 22.05 k -> 16 k, a
    truncated last frame, WAVE_FORMAT_EXTENSIBLE-length fmt, 3 channels."""
    import struct
    rng = np.random.default_rng(77)
    t = np.arange(4410) / 44100.0
    x = (0.4 * np.sin(2 * np.pi * 330.0 * t) + 0.05 * rng.standard_normal(len(t))).astype(np.float32)
    i16 = lambda v: np.clip(np.round(v * 32767.0), -32768, 32767).astype("<i2")          # noqa: E731
    i32 = lambda v: np.clip(np.round(v.astype(np.float64) * 2147483647.0), -2147483648, 2147483647).astype("<i4")  # noqa: E731
    st = np.stack([i16(x), i16(x[::-1])], axis=1).reshape(-1)
    three = np.stack([i16(x), i16(0.5 * x), i16(-0.25 * x)], axis=1).reshape(-1)
    lst = b"LIST" + struct.pack("<I", 6) + b"abcdef"
    return {
        "pcm16_stereo_44k": (wav_bytes(1, 16, 2, 44100, st.tobytes()), 16000),
        "pcm16_mono_16k": (wav_bytes(1, 16, 1, 16000, i16(x[:1600]).tobytes()), 16000),
        "pcm32_mono_8k": (wav_bytes(1, 32, 1, 8000, i32(x[:800]).tobytes()), 16000),
        "f32_mono_22k": (wav_bytes(3, 32, 1, 22050, x[:2205].tobytes()), 16000),
        "f32_stereo_16k_list_chunk": (wav_bytes(3, 32, 2, 16000, np.stack([x[:900], -x[:900]], 1).reshape(-1).tobytes(), extra_chunk=lst), 16000),
        "pcm16_3ch_8k": (wav_bytes(1, 16, 3, 8000, three[: 3 * 801].tobytes()), 16000),
        "pcm16_stereo_truncated_frame": (wav_bytes(1, 16, 2, 16000, st[:2001].tobytes()), 16000),       # odd sample count: last frame dropped
        "pcm16_mono_fmt18": (wav_bytes(1, 16, 1, 16000, i16(x[:700]).tobytes(), fmt_extra=b"\x00\x00"), 16000),
        "pcm16_mono_48k_to_8k": (wav_bytes(1, 16, 1, 48000, i16(x[:4800]).tobytes()), 8000),
        "f32_mono_16k_same_rate": (wav_bytes(3, 32, 1, 16000, x[:333].tobytes(), odd_pad=True), 16000),
    }


def main() -> None:
    OUT.mkdir(parents=True, exist_ok=True)
    audio = _load_ref_audio()
    tok = _load_ref_tokenizer()
    split = _load_ref_split()

    # ---- filterbank (audio.py:41-96)
    fb = audio._get_mel_filterbank()
    assert fb.shape == (128, 201) and fb.dtype == np.float32
    np.save(OUT / "mel_filterbank.npy", fb)

    # ---- mel (audio.py:238-278)
    full = {}
    digest = {}
    for kind in ("silence", "tone", "noise", "speechlike"):
        for secs in (0.5, 1.0, 5.0, 30.0):
            n = int(secs * 16000)
            x = signal(kind, n)
            mel = np.asarray(audio.log_mel_spectrogram(x))
            assert mel.dtype == np.float32
            key = f"{kind}_{secs:g}s"
            if secs <= 1.0 or (kind in ("noise", "speechlike") and secs == 5.0):
                full[key] = mel
            flat = mel.reshape(-1)
            digest[key] = {
                "shape": list(mel.shape),
                "sha256": hashlib.sha256(flat.tobytes()).hexdigest(),
                "every97": [float(v) for v in flat[::97][:4096]],
                "max": float(flat.max()),
                "min": float(flat.min()),
                "mean": float(flat.astype(np.float64).mean()),
            }
    # odd lengths: ragged last frame / last chunk handling
    for n in (400, 401, 1234, 16000 * 3 + 77):
        x = signal("noise", n)
        mel = np.asarray(audio.log_mel_spectrogram(x))
        full[f"noise_n{n}"] = mel
    np.savez_compressed(OUT / "mel_full.npz", **full)
    (OUT / "mel_digest.json").write_text(json.dumps(digest))

    # ---- audio files through the live reference loader (audio.py:103-204): the WAV images and what it returns
    import tempfile
    arrs, meta = {}, {}
    with tempfile.TemporaryDirectory() as td:
        for name, (img, target) in wav_cases().items():
            f = Path(td) / f"{name}.wav"
            f.write_bytes(img)
            got = audio.load_audio(f, target_sr=target)
            assert got.dtype == np.float32 and got.ndim == 1
            arrs[name + "__wav"] = np.frombuffer(img, dtype=np.uint8)
            arrs[name + "__out"] = got
            meta[name] = {"target_sr": target, "n_out": int(len(got)), "sha256": hashlib.sha256(got.tobytes()).hexdigest()}
        bad = {"not_riff": b"RIFX" + b"\x00" * 40, "no_data": wav_bytes(1, 16, 1, 16000, b"")[:36],
               "adpcm": wav_bytes(2, 4, 1, 16000, b"\x00" * 64), "pcm24": wav_bytes(1, 24, 1, 16000, b"\x00" * 48)}
        for name, img in bad.items():                      # what the RIFF fast path raises (it is then swallowed by load_audio)
            f = Path(td) / f"{name}.wav"
            f.write_bytes(img)
            try:
                audio._read_wav_pcm(f)
                meta["bad_" + name] = {"raises": None}
            except Exception as e:                          # noqa: BLE001
                meta["bad_" + name] = {"raises": type(e).__name__, "message": str(e)}
            arrs["bad_" + name + "__wav"] = np.frombuffer(img, dtype=np.uint8)
    np.savez_compressed(OUT / "load_audio.npz", **arrs)
    (OUT / "load_audio.json").write_text(json.dumps(meta, indent=1))

    # ---- raw STFT power for one short signal (audio.py:211-235, :266)
    x = signal("noise", 8000)
    st = audio._stft(x, 400, 160)
    np.save(OUT / "stft_power_noise_0.5s.npy", (np.abs(st[:, :-1]) ** 2).astype(np.float32))

    # ---- prompts (tokenizer.py:56-86)
    prompts = {}
    for n in (0, 13, 65, 390):
        prompts[f"n{n}_none"] = tok.build_prompt(n)
        prompts[f"n{n}_lang"] = tok.build_prompt(n, [6364])  # " English" is one id in Qwen vocab
    prompts["constants"] = {
        "AUDIO_START_TOKEN_ID": tok.AUDIO_START_TOKEN_ID, "AUDIO_END_TOKEN_ID": tok.AUDIO_END_TOKEN_ID,
        "AUDIO_PAD_TOKEN_ID": tok.AUDIO_PAD_TOKEN_ID, "IM_START_TOKEN_ID": tok.IM_START_TOKEN_ID,
        "IM_END_TOKEN_ID": tok.IM_END_TOKEN_ID, "ENDOFTEXT_TOKEN_ID": tok.ENDOFTEXT_TOKEN_ID,
        "ASR_TEXT_TOKEN_ID": tok.ASR_TEXT_TOKEN_ID, "EOS_TOKEN_IDS": sorted(tok.EOS_TOKEN_IDS),
    }
    prompts["parse_output"] = [
        [s, tok.parse_output(s)] for s in (
            "language English<asr_text>Hello world.<|im_end|>",
            "language Chinese<asr_text> 你好 <|endoftext|>",
            "language English some text",
            "plain text only",
            "",
            "<asr_text>",
        )
    ]
    (OUT / "prompts.json").write_text(json.dumps(prompts))

    # ---- split points (model.py:454-513); the five reference test signals + long-form
    sp = {}
    sr = 16000
    sp["short_zeros"] = split(np.zeros(sr, dtype=np.float32), sr * 20, 5 * sr)
    c = sr * 10
    sp["rng0_2.5chunks"] = split(np.random.default_rng(0).random(int(c * 2.5)).astype(np.float32), c, 5 * sr)
    sp["rng1_3chunks"] = split(np.random.default_rng(1).random(c * 3).astype(np.float32), c, 5 * sr)
    s = np.ones(sr * 25, dtype=np.float32) * 0.5
    s[sr * 9: sr * 11] = 0.0
    sp["silent_window"] = split(s, c, 5 * sr)
    c5 = sr * 5
    sp["rng2_5x5s"] = split(np.random.default_rng(2).random(c5 * 5).astype(np.float32), c5, 2 * sr)
    long = (np.random.default_rng(3).standard_normal(sr * 1200) * 0.1).astype(np.float32)
    sp["rng3_20min_30s"] = split(long, 30 * sr, 5 * sr)
    sp["speechlike_120s_30s"] = split(signal("speechlike", 120 * sr), 30 * sr, 5 * sr)
    sp["tiny_lt_frame"] = split(np.ones(100, dtype=np.float32), 50, 10)
    sp = {k: [int(v) for v in vals] for k, vals in sp.items()}
    (OUT / "split_points.json").write_text(json.dumps(sp))
    print("wrote", sorted(p.name for p in OUT.iterdir()))
    for k in ("rng0_2.5chunks", "silent_window"):
        print(k, sp[k])
    print("20min first3", sp["rng3_20min_30s"][:3], len(sp["rng3_20min_30s"]))


if __name__ == "__main__":
    main()
