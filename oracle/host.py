"""Oracle: integer host logic (prompt ids, long-audio split points, chunk /
window bookkeeping) -- test infrastructure.  All results are exact."""
from __future__ import annotations

import numpy as np

# tokenizer.py:16-24
AUDIO_START_TOKEN_ID = 151669
AUDIO_END_TOKEN_ID = 151670
AUDIO_PAD_TOKEN_ID = 151676
IM_START_TOKEN_ID = 151644
IM_END_TOKEN_ID = 151645
ENDOFTEXT_TOKEN_ID = 151643
ASR_TEXT_TOKEN_ID = 151704
EOS_TOKEN_IDS = frozenset({ENDOFTEXT_TOKEN_ID, IM_END_TOKEN_ID})


def build_prompt(n_audio_tokens: int, language_name_tokens=None) -> list:
    """tokenizer.py:56-86: <|im_start|>system\\n<|im_end|>\\n<|im_start|>user\\n
    <|audio_start|> pad*N <|audio_end|><|im_end|>\\n<|im_start|>assistant\\n
    language {name}<asr_text>."""
    head = [IM_START_TOKEN_ID, 8948, 198, IM_END_TOKEN_ID, 198, IM_START_TOKEN_ID, 872, 198,
            AUDIO_START_TOKEN_ID]
    tail = [AUDIO_END_TOKEN_ID, IM_END_TOKEN_ID, 198, IM_START_TOKEN_ID, 77091, 198]
    return (head + [AUDIO_PAD_TOKEN_ID] * n_audio_tokens + tail + [11528]
            + list(language_name_tokens or []) + [ASR_TEXT_TOKEN_ID])


def find_split_points(samples: np.ndarray, chunk_samples: int, search_samples: int,
                      frame_samples: int = 480) -> list:
    """model.py:454-513: per-frame RMS (float32, tail dropped); for each
    boundary k*chunk_samples < total: frames [c-r, c+r] clipped to
    [0, n_frames-1], first argmin -> frame*frame_samples; degenerate window
    -> the boundary itself."""
    total = len(samples)
    n_frames = total // frame_samples
    if n_frames == 0:
        return []
    fr = np.asarray(samples[: n_frames * frame_samples], dtype=np.float32).reshape(n_frames, frame_samples)
    # np.mean of float32 uses pairwise float32 summation per frame, same as the
    # reference's per-slice np.mean (model.py:491)
    energy = np.sqrt(np.mean(fr ** 2, axis=1)).astype(np.float32)
    r = search_samples // frame_samples
    out = []
    b = chunk_samples
    while b < total:
        c = b // frame_samples
        lo, hi = max(0, c - r), min(n_frames - 1, c + r)
        if lo >= hi:
            out.append(b)
        else:
            out.append((int(np.argmin(energy[lo: hi + 1])) + lo) * frame_samples)
        b += chunk_samples
    return out


def conv_output_length(n: int) -> int:
    """encoder.py:198-207: three times L -> (L-1)//2 + 1."""
    for _ in range(3):
        n = (n - 1) // 2 + 1
    return n


def chunk_layout(n_frames: int, chunk: int = 100):
    """encoder.py:254-268,289-291: real length of every 100-frame chunk and
    the number of valid tokens each keeps."""
    real = [min(chunk, n_frames - o) for o in range(0, n_frames, chunk)]
    return real, [conv_output_length(r) for r in real]


def window_cu_seqlens(total_tokens: int, window_tokens: int) -> list:
    """encoder.py:297-309: fixed windows of window_tokens (=13*(800//100)=104)
    plus a remainder block."""
    cu = [0]
    for _ in range(total_tokens // window_tokens):
        cu.append(cu[-1] + window_tokens)
    if total_tokens % window_tokens:
        cu.append(cu[-1] + total_tokens % window_tokens)
    return cu
