"""Oracle: prompt embedding merge, sampling filters and the generation loop
(torch-CPU) -- test infrastructure.  Restates reference generate.py:20-315."""
from __future__ import annotations

import torch

from .decoder import DecoderOracle, bf
from .host import AUDIO_PAD_TOKEN_ID, EOS_TOKEN_IDS


def prepare_inputs(encoder_output: torch.Tensor, input_ids, embed_weight: torch.Tensor,
                   audio_pad_id: int = AUDIO_PAD_TOKEN_ID) -> torch.Tensor:
    """generate.py:20-81: rows whose id == audio_pad_id are replaced, in order,
    by the encoder rows (rounded to the embedding dtype, bf16); count mismatch
    -> ValueError.  encoder_output (N, H); returns (T, H)."""
    ids = torch.as_tensor(list(input_ids), dtype=torch.long)
    emb = embed_weight[ids].clone()
    pos = (ids == audio_pad_id).nonzero().flatten()
    if len(pos) == 0:
        return emb
    if len(pos) != encoder_output.shape[0]:
        raise ValueError(f"Number of audio-pad tokens ({len(pos)}) does not match "
                         f"encoder output length ({encoder_output.shape[0]}).")
    emb[pos] = bf(encoder_output.to(torch.float32))
    return emb


def apply_repetition_penalty(logits: torch.Tensor, recent_tokens, penalty: float, bf16: bool = True):
    """generate.py:88-131: unique recent ids in range get l/p if l>0 else l*p.
    With bf16 logits the Python scalar is weakly typed, so it is rounded to
    bf16 (1.2 -> 1.203125) and the quotient/product is rounded to bf16."""
    if penalty == 1.0 or not recent_tokens:
        return logits
    out = logits.clone()
    V = logits.shape[0]
    idx = torch.tensor(sorted({t for t in recent_tokens if 0 <= t < V}), dtype=torch.long)
    if len(idx) == 0:
        return out
    p = bf(torch.tensor(penalty)) if bf16 else torch.tensor(penalty)
    sel = out[idx]
    pen = torch.where(sel > 0, sel / p, sel * p)
    out[idx] = bf(pen) if bf16 else pen
    return out


def top_k_filter(logits: torch.Tensor, top_k: int) -> torch.Tensor:
    """generate.py:134-141: keep logits >= k-th largest (ties kept)."""
    if top_k <= 0:
        return logits
    thr = torch.sort(logits, descending=True).values[top_k - 1]
    return torch.where(logits >= thr, logits, torch.full_like(logits, float("-inf")))


def top_p_filter(logits: torch.Tensor, top_p: float) -> torch.Tensor:
    """generate.py:144-163: softmax, sort descending, drop where cumsum - p >= top_p."""
    if top_p >= 1.0:
        return logits
    probs = torch.softmax(logits, dim=-1)
    sp, si = torch.sort(probs, descending=True, stable=True)
    remove = (torch.cumsum(sp, 0) - sp) >= top_p
    out = logits.clone()
    out[si[remove]] = float("-inf")
    return out


def sample(logits: torch.Tensor, temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.0,
           recent_tokens=None, generator=None, bf16: bool = True) -> int:
    """generate.py:166-215.  Order: penalty -> argmax if T==0 -> /T -> top-k -> top-p -> categorical."""
    logits = logits.reshape(-1).to(torch.float32)
    if repetition_penalty != 1.0 and recent_tokens:
        logits = apply_repetition_penalty(logits, recent_tokens, repetition_penalty, bf16)
    if temperature == 0.0:
        return int(torch.argmax(logits))            # first maximal index
    t = bf(torch.tensor(temperature)) if bf16 else torch.tensor(temperature)
    logits = logits / t
    if bf16:
        logits = bf(logits)
    if top_k > 0:
        logits = top_k_filter(logits, top_k)
    if top_p < 1.0:
        logits = top_p_filter(logits, top_p)
    probs = torch.softmax(logits, dim=-1)
    return int(torch.multinomial(probs, 1, generator=generator))


def generate(dec: DecoderOracle, encoder_output: torch.Tensor, input_ids, max_tokens=8192,
             temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.0,
             repetition_context_size=100, generator=None, force_tokens=None, return_logits=False):
    """generate.py:222-315.  ``force_tokens`` (test aid): teacher-force these
    ids instead of the sampled ones while still recording the logits."""
    dec.reset()
    emb = prepare_inputs(encoder_output, input_ids, dec.W["embed_tokens.weight"])
    logits = dec.forward(emb)
    all_logits = [logits[-1].clone()]
    nxt = sample(logits[-1], temperature, top_p, top_k, repetition_penalty, None, generator)
    if force_tokens is not None:
        nxt = int(force_tokens[0])
    out = [nxt]
    for step in range(max_tokens - 1):
        if nxt in EOS_TOKEN_IDS:
            break
        logits = dec.forward(dec.embed([nxt]))
        all_logits.append(logits[-1].clone())
        recent = out[-repetition_context_size:] if repetition_context_size > 0 else None
        nxt = sample(logits[-1], temperature, top_p, top_k, repetition_penalty, recent, generator)
        if force_tokens is not None:
            nxt = int(force_tokens[step + 1])
        out.append(nxt)
    return (out, all_logits) if return_logits else out
