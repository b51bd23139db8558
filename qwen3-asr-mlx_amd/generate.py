"""Prompt-embedding merge, sampling and the generation loop (mirror of reference generate.py),
plus the batched loop the reference lacks.  Token selection and all bookkeeping run on the
device (csrc/sampling.hip); the host only replays the captured decode graph."""
from __future__ import annotations

from typing import Optional, Sequence

from . import ops
from .decoder import KVCache, TextDecoder
from .tokenizer import AUDIO_PAD_TOKEN_ID, ENDOFTEXT_TOKEN_ID, EOS_TOKEN_IDS, IM_END_TOKEN_ID


def _merge(enc_feats, ids_per_seq: Sequence[Sequence[int]], tok_counts: Sequence[int], decoder: TextDecoder,
           audio_pad_id: int = AUDIO_PAD_TOKEN_ID):
    """Packed prompt embeddings bf16 [sum T][H]: audio-pad rows <- encoder rows (in order, rounded
    to bf16), others <- embedding rows (reference generate.py:46-81)."""
    import torch
    ids, src = [], []
    base = 0
    for seq, n_enc in zip(ids_per_seq, tok_counts):
        n_pad = 0
        for t in seq:
            if t == audio_pad_id:
                src.append(base + n_pad)
                n_pad += 1
            else:
                src.append(-1)
        if n_pad and n_pad != n_enc:
            raise ValueError(f"Number of audio-pad tokens ({n_pad}) does not match encoder output length ({n_enc}).")
        if not n_pad:                                     # no placeholders: pure text (generate.py:55-56)
            src[len(src) - len(seq):] = [-1] * len(seq)
        ids += list(seq)
        base += n_enc
    dev = decoder.device
    ids_t = torch.tensor(ids, dtype=torch.int32, device=dev)
    src_t = torch.tensor(src, dtype=torch.int32, device=dev)
    H = decoder.config.hidden_size
    out = torch.empty((len(ids), H), dtype=torch.bfloat16, device=dev)
    enc = enc_feats if enc_feats is not None and enc_feats.numel() else None
    ops.call("q3asr_embed_merge", ops.ptr(ids_t), ops.ptr(src_t), ops.ptr(decoder.w["embed"]), ops.ptr(enc), ops.ptr(out),
             len(ids), H, ops.sp())
    return out


def prepare_inputs(encoder_output, input_ids: list[int], embed_tokens, audio_pad_id: int = AUDIO_PAD_TOKEN_ID):
    """(1, N, H) encoder output + prompt ids -> (1, T, H) embeddings in the embedding dtype
    (reference generate.py:20-81).  ``embed_tokens`` is ``decoder.embed_tokens``."""
    import torch
    from . import _lib

    class _Shim:                                          # reuse _merge with just an embedding table
        pass
    sh = _Shim()
    sh.device = embed_tokens.weight.device
    sh.w = {"embed": embed_tokens.weight}
    sh.config = type("C", (), {"hidden_size": embed_tokens.weight.shape[1]})()
    enc = torch.as_tensor(encoder_output).to(sh.device, torch.float32)
    enc = enc.reshape(-1, enc.shape[-1]).contiguous()
    _lib.load()
    return _merge(enc, [list(input_ids)], [enc.shape[0]], sh, audio_pad_id)[None]


def sample(logits, temperature: float = 0.0, top_p: float = 1.0, top_k: int = 0, repetition_penalty: float = 1.0,
           recent_tokens: Optional[list[int]] = None, seed: Optional[int] = None) -> int:
    """One token from (1,V)/(V,) logits (reference generate.py:166-215) via q3asr_sample_step.
    Logits are taken at bf16 (the decoder's logits dtype)."""
    import torch
    global _sample_calls
    dev = logits.device if isinstance(logits, torch.Tensor) and logits.is_cuda else torch.device("cuda")
    row = torch.as_tensor(logits).reshape(-1).to(dev, torch.float32)
    V = row.numel()
    Vp = (V + 31) // 32 * 32
    buf = torch.full((1, Vp), float("-inf"), dtype=torch.bfloat16, device=dev)
    buf[0, :V] = row.to(torch.bfloat16)
    recent = list(recent_tokens or [])
    n = len(recent)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
    out_tokens = i32([recent + [0]])
    n_gen, fin, cur, pos, mx = i32([n]), i32([0]), i32([0]), i32([0]), i32([n + 1])
    if seed is None:
        _sample_calls += 1
        seed = (torch.initial_seed() * 1000003 + _sample_calls) & 0xFFFFFFFFFFFFFFFF
    ops.call("q3asr_sample_step", ops.ptr(buf), Vp, 1, Vp, ops.ptr(out_tokens), n + 1, ops.ptr(n_gen), ops.ptr(fin),
             ops.ptr(cur), ops.ptr(pos), ops.ptr(mx), float(temperature), float(top_p), int(top_k),
             float(repetition_penalty), max(n, 1), int(seed), -1, -1, ops.sp())
    return int(cur.item())


_sample_calls = 0


def generate_batch(decoder: TextDecoder, enc_feats, tok_counts: Sequence[int], prompts: Sequence[Sequence[int]],
                   max_tokens, temperature: float = 0.0, top_p: float = 1.0, top_k: int = 0,
                   repetition_penalty: float = 1.0, repetition_context_size: int = 100, seed: int = 0,
                   use_graph: bool = True, return_state: bool = False):
    """Batched greedy/sampled decoding of B independent sequences (B <= 64).  Per sequence the
    result equals the reference loop (generate.py:265-315): first token without penalty, EOS kept,
    at most max_tokens ids."""
    import torch
    cfg = decoder.config
    B = len(prompts)
    if B < 1 or B > 64:
        raise ValueError("generate_batch handles 1..64 sequences per call")
    mt = [int(max_tokens)] * B if isinstance(max_tokens, int) else [int(m) for m in max_tokens]
    if min(mt) < 1:
        raise ValueError("max_tokens must be >= 1")
    seq_lens = [len(p) for p in prompts]
    embeds = _merge(enc_feats, prompts, tok_counts, decoder)
    max_new = max(mt)
    Lmax = (max(t + m for t, m in zip(seq_lens, mt)) + 63) // 64 * 64
    cache = decoder.batch_cache(B, Lmax)
    ws = decoder.workspace(B, max_new)
    dev = decoder.device
    ws.n_gen.zero_()
    ws.finished.zero_()
    ws.pos.copy_(torch.tensor([t - 1 for t in seq_lens], dtype=torch.int32), non_blocking=False)
    ws.max_new.copy_(torch.tensor(mt, dtype=torch.int32))
    sampling = {"temperature": float(temperature), "top_p": float(top_p), "top_k": int(top_k),
                "repetition_penalty": float(repetition_penalty),
                "repetition_context_size": int(repetition_context_size), "seed": int(seed)}
    decoder.prefill(embeds, seq_lens, cache, logits_out=ws.logits)
    decoder.sample_step(ws, ws.logits, sampling)                     # first token (generate.py:278-286)
    steps = max_new - 1
    done = 0
    if steps > 0:
        if use_graph:
            # capture on a side stream (legacy-stream capture is not allowed), replay on it too
            side = _side_stream(dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                decoder.decode_step(ws, cache, sampling)             # eager: loads code objects, real step 1
                done = 1
                if steps > 1:
                    decoder.capture_step(ws, cache, sampling)
                checked = done
                while done < steps:
                    # several steps per launch while that cannot overshoot the budget
                    done += decoder.replay_step(ws, multi=steps - done >= decoder.MULTI_STEPS)
                    if done - checked >= 16:
                        checked = done
                        if bool(ws.finished.all()):
                            break
            torch.cuda.current_stream().wait_stream(side)
        else:
            while done < steps:
                decoder.decode_step(ws, cache, sampling)
                done += 1
                if done % 16 == 0 and bool(ws.finished.all()):
                    break
    n_gen = ws.n_gen.tolist()
    toks = ws.out_tokens.cpu()
    out = [toks[b, : n_gen[b]].tolist() for b in range(B)]
    return (out, ws, cache) if return_state else out


def generate_continuous(decoder: TextDecoder, feats: Sequence, prompts: Sequence[Sequence[int]], max_tokens: Sequence[int],
                        slots: int = 32, temperature: float = 0.0, top_p: float = 1.0, top_k: int = 0,
                        repetition_penalty: float = 1.0, repetition_context_size: int = 100, seed: int = 0,
                        check_every: int = 16, min_admit: Optional[int] = None, stats: Optional[dict] = None) -> list[list[int]]:
    """Any number of sequences through `slots` decode rows: a row whose sequence has ended (EOS or budget) is handed to the
    next waiting sequence, whose prompt is prefilled into that row of the KV cache while the other rows keep their state
    (the reference decodes one utterance at a time, model.py:239).  `feats[i]`: encoder output
    fp32 [N_i][H] of sequence i.  Sequences are admitted in the order given.  Greedy results equal generate_batch's
    (and the reference loop's) token for token: a row's arithmetic does not depend on its neighbours, its row index or the
    cache capacity.  With temperature > 0 the draws depend on the row a sequence lands in (reproducible, but not equal
    to a standalone call).  `min_admit`: free rows to wait for before prefilling (one packed prefill of several prompts
    costs little more than a prefill of one; default rows / 8), ignored once nothing else is running."""
    import torch
    from collections import deque
    n = len(prompts)
    if n == 0:
        return []
    if not (len(feats) == len(max_tokens) == n):
        raise ValueError("generate_continuous: feats, prompts and max_tokens must have one entry per sequence")
    mt = [int(m) for m in max_tokens]
    if min(mt) < 1:
        raise ValueError("max_tokens must be >= 1")
    B = max(1, min(int(slots), n, 64))
    min_admit = max(1, B // 8) if min_admit is None else max(1, int(min_admit))
    seq_lens = [len(p) for p in prompts]
    tok_counts = [int(f.shape[0]) for f in feats]
    max_new = max(mt)
    Lmax = (max(t + m for t, m in zip(seq_lens, mt)) + 63) // 64 * 64
    cache = decoder.batch_cache(B, Lmax)
    ws = decoder.workspace(B, max_new)
    dev = decoder.device
    V = decoder.config.vocab_size
    sampling = {"temperature": float(temperature), "top_p": float(top_p), "top_k": int(top_k),
                "repetition_penalty": float(repetition_penalty),
                "repetition_context_size": int(repetition_context_size), "seed": int(seed)}
    waiting = deque(range(n))
    owner: list = [None] * B                     # sequence in each row
    out: list = [None] * n
    admit_logits = torch.empty((B, V), dtype=torch.bfloat16, device=dev)
    st = {"steps": 0, "admissions": 0, "checks": 0, "row_steps_used": 0}

    def admit(rows: list) -> None:
        """Prefill the next waiting sequences into `rows`, reset those rows' counters, draw their first token."""
        seqs = [waiting.popleft() for _ in rows]
        for r, u in zip(rows, seqs):
            owner[r] = u
        packed = torch.cat([torch.as_tensor(feats[u]).to(dev, torch.float32).reshape(tok_counts[u], -1) for u in seqs])
        embeds = _merge(packed, [prompts[u] for u in seqs], [tok_counts[u] for u in seqs], decoder)
        full = len(rows) == B and rows == list(range(B))
        logits = ws.logits if full else admit_logits
        decoder.prefill(embeds, [seq_lens[u] for u in seqs], cache, logits_out=logits, slots=None if full else rows)
        idx = torch.tensor(rows, dtype=torch.long, device=dev)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
        ws.n_gen.index_fill_(0, idx, 0)
        ws.finished.index_fill_(0, idx, 0)
        ws.pos.index_copy_(0, idx, i32([seq_lens[u] - 1 for u in seqs]))
        ws.max_new.index_copy_(0, idx, i32([mt[u] for u in seqs]))
        if full:
            decoder.sample_step(ws, ws.logits, sampling)
        else:
            for j, r in enumerate(rows):                                    # first token (generate.py:278-286), this row only
                decoder.sample_rows(ws, logits[j], r, sampling)
        st["admissions"] += len(rows)

    side = _side_stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ws.finished.fill_(1)                     # rows without a sequence stay inert
        ws.n_gen.zero_()
        admit(list(range(B)))
        captured = False
        while True:
            fin, n_gen = ws.finished.tolist(), ws.n_gen.tolist()           # synchronises `side`
            st["checks"] += 1
            ended = [r for r in range(B) if owner[r] is not None and fin[r]]
            if ended:
                toks = ws.out_tokens[torch.tensor(ended, dtype=torch.long, device=dev)].cpu()
                for j, r in enumerate(ended):
                    out[owner[r]] = toks[j, : n_gen[r]].tolist()
                    st["row_steps_used"] += n_gen[r]
                    owner[r] = None
            free = [r for r in range(B) if owner[r] is None]
            active = [r for r in range(B) if owner[r] is not None]
            if waiting and free and (len(free) >= min(min_admit, len(waiting)) or not active):
                admit(free[: len(waiting)])
                continue                                                   # a budget of 1 ends at the first token
            if not active:
                break
            # steps until the next look: no further than the first row that must end by budget
            k = max(1, min(check_every, min(mt[owner[r]] - n_gen[r] for r in active)))
            if not captured:
                decoder.decode_step(ws, cache, sampling)                   # eager: loads code objects, a real step
                decoder.capture_step(ws, cache, sampling)
                captured = True
                k -= 1
                st["steps"] += 1
            while k > 0:
                d = decoder.replay_step(ws, multi=k >= decoder.MULTI_STEPS)
                k -= d
                st["steps"] += d
    torch.cuda.current_stream().wait_stream(side)
    if stats is not None:
        stats.update(st, rows=B)
    return out


_streams: dict = {}


def _side_stream(dev):
    import torch
    from .decoder import current_lane
    key = (str(dev), current_lane())
    s = _streams.get(key)
    if s is None:
        s = _streams[key] = torch.cuda.Stream(device=dev)
    return s


def generate(decoder: TextDecoder, encoder_output, input_ids: list[int], max_tokens: int = 8192,
             temperature: float = 0.0, top_p: float = 1.0, top_k: int = 0, repetition_penalty: float = 1.0,
             repetition_context_size: int = 100) -> list[int]:
    """Reference signature (generate.py:222-232): one sequence, returns generated ids (EOS included)."""
    import torch
    enc = torch.as_tensor(encoder_output).to(decoder.device, torch.float32)
    enc = enc.reshape(-1, enc.shape[-1]).contiguous()
    # the KV cache is sized for the budget actually reachable, not the 8192 default
    return generate_batch(decoder, enc, [enc.shape[0]], [list(input_ids)], int(max_tokens), temperature, top_p, top_k,
                          repetition_penalty, repetition_context_size, seed=_next_seed())[0]


_gen_calls = 0


def _next_seed() -> int:
    import torch
    global _gen_calls
    _gen_calls += 1
    return (torch.initial_seed() * 2654435761 + _gen_calls) & 0x7FFFFFFFFFFFFFFF
