#!/usr/bin/env python3
"""Parity harness for a REAL checkpoint -- needs files this pipeline does not
have: a local copy of mlx-community/Qwen3-ASR-1.7B-bf16 (config.json, model.safetensors, tokenizer files) and a
JSON captured on a Mac from the MLX reference:

    [{"audio": "a.wav", "language": "en", "tokens": [ids...], "text": "...",
      "first_logits_top5": [[id, value], ...], "max_tokens": 64}, ...]        (max_tokens optional: the budget the
                                                                               capture ran with; default = transcribe()'s)

    python tools/check_real_checkpoint.py /path/to/model_dir reference_outputs.json

For every entry it runs this package's `transcribe` path (greedy, repetition_penalty 1.2, the API defaults)
and reports token agreement with the MLX tokens (exact match, first divergence) and, where `first_logits_top5` is given
(the five largest logits of the FIRST generated position as [token id, value] pairs, from the MLX run), compares this
package's first-position logits at those ids: same top-1, same top-5 set, largest |difference| against `--logit-atol`
(default 0.125 = 4 bf16 ulps at |logit| ~ 4; the logits are bf16 on both sides, reference decoder.py:252).  This is the only route to *demonstrating* "token-identical under greedy"
against MLX itself; inside this repository parity is established transitively through oracle/.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def first_logits(model, samples, lang: str):
    """bf16 logits of the first generated position (mel -> encoder -> prompt merge -> prefill) as float32 on the host."""
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd.generate import _merge
    from qwen3_asr_mlx_amd.tokenizer import build_prompt
    dec = model._decoder
    audio, lens_dev, lens = q.audio.pack_audio([samples], dec.device)
    mel, n_frames = q.audio.get_frontend(dec.device)(audio, lens_dev, lens)
    feats, counts = model._encoder.forward_batch(mel, n_frames)
    prompt = build_prompt(counts[0], model._tokenizer.encode(f" {lang}"))
    cache = dec.batch_cache(1, (len(prompt) + 63) // 64 * 64)
    return dec.prefill(_merge(feats, [prompt], counts, dec), [len(prompt)], cache)[0].float().cpu()


def compare_top5(logits, top5, atol: float) -> dict:
    """`top5`: [[id, value], ...] from the MLX run, best first."""
    import torch
    ids = [int(t[0]) for t in top5]
    want = torch.tensor([float(t[1]) for t in top5])
    got = logits[torch.tensor(ids)]
    mine = torch.topk(logits, len(ids)).indices.tolist()
    return {"same_top1": mine[0] == ids[0], "same_top5_set": set(mine) == set(ids),
            "max_abs_diff": float((got - want).abs().max()), "within_atol": bool((got - want).abs().max() <= atol)}


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    atol = 0.125
    if "--logit-atol" in argv:
        i = argv.index("--logit-atol")
        atol = float(argv[i + 1])
        del argv[i: i + 2]
    if len(argv) != 2:
        print(__doc__)
        return 2
    import numpy as np
    import qwen3_asr_mlx_amd as q
    model_dir, ref_path = Path(argv[0]), Path(argv[1])
    ref = json.loads(ref_path.read_text())
    model = q.Qwen3ASR.from_pretrained(model_dir)
    model.warm_up()
    exact = logit_items = logit_ok = 0
    for item in ref:
        audio_path = (ref_path.parent / item["audio"]).resolve()
        samples = q.load_audio(audio_path)
        lang = model._resolve_language(item.get("language"))
        budget = int(item["max_tokens"]) if item.get("max_tokens") else max(256, int(len(samples) / 16000 * 50))
        toks = model.generate_tokens([samples], [budget], lang, repetition_penalty=1.2)[0]
        want = item["tokens"]
        n_same = next((i for i, (a, b) in enumerate(zip(toks, want)) if a != b), min(len(toks), len(want)))
        ok = toks == want
        exact += ok
        text = model._decode_output(toks)
        print(f"{item['audio']}: {'IDENTICAL' if ok else f'diverges at token {n_same} of {len(want)}'}"
              f"{'' if ok or 'text' not in item else ' | ref: ' + item['text'][:60] + ' | got: ' + text[:60]}")
        if item.get("first_logits_top5"):
            c = compare_top5(first_logits(model, samples, lang), item["first_logits_top5"], atol)
            logit_items += 1
            logit_ok += c["same_top1"] and c["within_atol"]
            print(f"    first-position logits: top-1 {'same' if c['same_top1'] else 'DIFFERENT'}, top-5 set "
                  f"{'same' if c['same_top5_set'] else 'different'}, max |diff| {c['max_abs_diff']:.4f} (atol {atol})")
    print(f"{exact}/{len(ref)} utterances token-identical" + (f"; {logit_ok}/{logit_items} first-position logit checks passed" if logit_items else ""))
    return 0 if exact == len(ref) and logit_ok == logit_items else 1


if __name__ == "__main__":
    raise SystemExit(main())
