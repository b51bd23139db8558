"""Prompt builder, special-token ids and the BPE tokenizer wrapper (mirror of reference
tokenizer.py).  CPU, microseconds, off the roofline."""
from __future__ import annotations

import json
import re
from pathlib import Path

# reference tokenizer.py:16-24
AUDIO_START_TOKEN_ID: int = 151669
AUDIO_END_TOKEN_ID: int = 151670
AUDIO_PAD_TOKEN_ID: int = 151676
IM_START_TOKEN_ID: int = 151644
IM_END_TOKEN_ID: int = 151645
ENDOFTEXT_TOKEN_ID: int = 151643
ASR_TEXT_TOKEN_ID: int = 151704
EOS_TOKEN_IDS: frozenset[int] = frozenset({ENDOFTEXT_TOKEN_ID, IM_END_TOKEN_ID})

_SYSTEM, _USER, _ASSISTANT, _NL, _LANGUAGE = 8948, 872, 77091, 198, 11528


def build_prompt(n_audio_tokens: int, language_name_tokens: list[int] | None = None) -> list[int]:
    """Chat-template ids for one utterance (reference tokenizer.py:56-86):
    system turn (empty), user turn = audio_start + N pads + audio_end, assistant turn opened
    with ``language {name}<asr_text>``."""
    ids = [IM_START_TOKEN_ID, _SYSTEM, _NL, IM_END_TOKEN_ID, _NL,
           IM_START_TOKEN_ID, _USER, _NL, AUDIO_START_TOKEN_ID]
    ids += [AUDIO_PAD_TOKEN_ID] * n_audio_tokens
    ids += [AUDIO_END_TOKEN_ID, IM_END_TOKEN_ID, _NL, IM_START_TOKEN_ID, _ASSISTANT, _NL, _LANGUAGE]
    ids += list(language_name_tokens or [])
    ids.append(ASR_TEXT_TOKEN_ID)
    return ids


def parse_output(text: str) -> str:
    """Legacy string post-processing (reference tokenizer.py:89-108)."""
    for marker in ("<|im_end|>", "<|endoftext|>"):
        text = text.replace(marker, "")
    head, sep, tail = text.partition("<asr_text>")
    if sep:
        return tail.strip()
    return re.sub(r"^language\s+\S+\s*", "", text, flags=re.IGNORECASE).strip()


class Tokenizer:
    """HF ``tokenizers`` wrapper: tokenizer.json, else vocab.json + merges.txt as byte-level
    BPE (reference tokenizer.py:115-174)."""

    def __init__(self, model_path: str | Path) -> None:
        from tokenizers import Tokenizer as HFTokenizer

        root = Path(model_path)
        tj, vj, mt = root / "tokenizer.json", root / "vocab.json", root / "merges.txt"
        if tj.exists():
            self._tok = HFTokenizer.from_file(str(tj))
        elif vj.exists() and mt.exists():
            from tokenizers import decoders, models, pre_tokenizers
            vocab = json.loads(vj.read_text(encoding="utf-8"))
            lines = mt.read_text(encoding="utf-8").splitlines()
            if lines and lines[0].startswith("#"):
                lines = lines[1:]
            merges = [tuple(ln.split()) for ln in lines if ln.strip()]
            self._tok = HFTokenizer(models.BPE(vocab=vocab, merges=merges))
            self._tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
            self._tok.decoder = decoders.ByteLevel()
        else:
            raise FileNotFoundError(f"No tokenizer.json or vocab.json+merges.txt found in {root}")

    def encode(self, text: str) -> list[int]:
        return self._tok.encode(text).ids

    def decode(self, token_ids: list[int], skip_special_tokens: bool = True) -> str:
        return self._tok.decode(token_ids, skip_special_tokens=skip_special_tokens)

    def build_prompt(self, n_audio_tokens: int, language: str = "English") -> list[int]:
        return build_prompt(n_audio_tokens, self.encode(f" {language}"))

    def parse_output(self, text: str) -> str:
        return parse_output(text)
