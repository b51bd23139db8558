"""Special-token ids, the chat-template prompt and the BPE wrapper (host side, microseconds; API of the reference's
tokenizer.py)."""
from __future__ import annotations

import json
import re
from pathlib import Path

_IDS = {  # reference tokenizer.py:16-24
    "AUDIO_START_TOKEN_ID": 151669, "AUDIO_END_TOKEN_ID": 151670, "AUDIO_PAD_TOKEN_ID": 151676, "IM_START_TOKEN_ID": 151644,
    "IM_END_TOKEN_ID": 151645, "ENDOFTEXT_TOKEN_ID": 151643, "ASR_TEXT_TOKEN_ID": 151704,
}
globals().update(_IDS)
AUDIO_START_TOKEN_ID: int
AUDIO_END_TOKEN_ID: int
AUDIO_PAD_TOKEN_ID: int
IM_START_TOKEN_ID: int
IM_END_TOKEN_ID: int
ENDOFTEXT_TOKEN_ID: int
ASR_TEXT_TOKEN_ID: int
EOS_TOKEN_IDS: frozenset[int] = frozenset((_IDS["ENDOFTEXT_TOKEN_ID"], _IDS["IM_END_TOKEN_ID"]))

# "<|im_start|>system\n<|im_end|>\n<|im_start|>user\n<|audio_start|>" ... "<|audio_end|><|im_end|>\n<|im_start|>assistant\nlanguage"
_HEAD = (151644, 8948, 198, 151645, 198, 151644, 872, 198, 151669)
_TAIL = (151670, 151645, 198, 151644, 77091, 198, 11528)


def build_prompt(n_audio_tokens: int, language_name_tokens: list[int] | None = None) -> list[int]:
    """Ids of one utterance's prompt (reference tokenizer.py:56-86): empty system turn, user turn holding
    `n_audio_tokens` audio pads, assistant turn opened with ``language {name}<asr_text>``."""
    return [*_HEAD, *([_IDS["AUDIO_PAD_TOKEN_ID"]] * n_audio_tokens), *_TAIL, *(language_name_tokens or ()), _IDS["ASR_TEXT_TOKEN_ID"]]


def parse_output(text: str) -> str:
    """String post-processing of a decoded transcript (reference tokenizer.py:89-108)."""
    text = text.replace("<|im_end|>", "").replace("<|endoftext|>", "")
    _, found, transcript = text.partition("<asr_text>")
    if found:
        return transcript.strip()
    return re.sub(r"^language\s+\S+\s*", "", text, flags=re.IGNORECASE).strip()


def _load_backend(root: Path):
    """tokenizer.json if present, otherwise vocab.json + merges.txt as byte-level BPE (reference tokenizer.py:124-152)."""
    import tokenizers as hf
    single = root / "tokenizer.json"
    if single.exists():
        return hf.Tokenizer.from_file(str(single))
    vocab_file, merges_file = root / "vocab.json", root / "merges.txt"
    if not (vocab_file.exists() and merges_file.exists()):
        raise FileNotFoundError(f"No tokenizer.json or vocab.json+merges.txt found in {root}")
    rows = merges_file.read_text(encoding="utf-8").splitlines()
    if rows and rows[0].startswith("#"):                      # "#version: 0.2" header, only as the very first line
        rows = rows[1:]
    rows = [r for r in rows if r.strip()]
    tok = hf.Tokenizer(hf.models.BPE(vocab=json.loads(vocab_file.read_text(encoding="utf-8")), merges=[tuple(r.split()) for r in rows]))
    tok.pre_tokenizer = hf.pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = hf.decoders.ByteLevel()
    return tok


class Tokenizer:
    """Thin wrapper over HF ``tokenizers`` (reference tokenizer.py:115-174)."""

    def __init__(self, model_path: str | Path) -> None:
        self._tok = _load_backend(Path(model_path))

    def encode(self, text: str) -> list[int]:
        return self._tok.encode(text).ids

    def decode(self, token_ids: list[int], skip_special_tokens: bool = True) -> str:
        return self._tok.decode(token_ids, skip_special_tokens=skip_special_tokens)

    def build_prompt(self, n_audio_tokens: int, language: str = "English") -> list[int]:
        return build_prompt(n_audio_tokens, self.encode(f" {language}"))

    def parse_output(self, text: str) -> str:
        return parse_output(text)
