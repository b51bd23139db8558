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
    import numpy as np
    import torch
    ids_parts, src_parts = [], []
    base = 0
    for seq, n_enc in zip(ids_per_seq, tok_counts):
        a = np.asarray(seq, dtype=np.int32)
        pad = a == audio_pad_id
        n_pad = int(pad.sum())
        if n_pad and n_pad != n_enc:
            raise ValueError(f"Number of audio-pad tokens ({n_pad}) does not match encoder output length ({n_enc}).")
        srow = np.full(a.shape, -1, dtype=np.int32)          # no placeholders: pure text (generate.py:55-56)
        if n_pad:
            srow[pad] = base + np.arange(n_pad, dtype=np.int32)
        ids_parts.append(a)
        src_parts.append(srow)
        base += n_enc
    ids = np.concatenate(ids_parts) if ids_parts else np.zeros(0, np.int32)
    src = np.concatenate(src_parts) if src_parts else np.zeros(0, np.int32)
    dev = decoder.device
    both = torch.from_numpy(np.stack([ids, src])).to(dev)                 # one upload
    ids_t, src_t = both[0], both[1]
    H = decoder.config.hidden_size
    out = torch.empty((len(ids), H), dtype=torch.bfloat16, device=dev)
    enc = enc_feats if enc_feats is not None and enc_feats.numel() else None
    ops.call("q3asr_embed_merge", ops.ptr(ids_t), ops.ptr(src_t), ops.ptr(decoder.w["embed"]), int(decoder.w["embed"].shape[0]), ops.ptr(enc),
             0 if enc is None else int(enc.shape[0]), ops.ptr(out), len(ids), H, ops.sp())
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
             float(repetition_penalty), max(n, 1), int(seed), -1, -1, 0, ops.sp())
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
                    if done - checked >= FINISH_CHECK_EVERY:
                        checked = done
                        if _all_finished(ws):
                            break
            # The host waits here, BEFORE the caller's stream is told to wait for the side stream: a hardware queue that sits
            # blocked on another queue's signal slows every kernel of the running queue (~0.5 us each, 110 us per decode step:
            # tools/ab_decode_loop.py), and the token read below synchronises anyway.
            side.synchronize()
            torch.cuda.current_stream().wait_stream(side)
        else:
            while done < steps:
                decoder.decode_step(ws, cache, sampling)
                done += 1
                if done % 16 == 0 and bool(ws.finished.all()):
                    break
    state = torch.stack([ws.n_gen, ws.finished]).tolist()            # one device-to-host copy for both counters
    n_gen = state[0]
    _raise_failed_rows([b for b in range(B) if state[1][b] == ROW_FAILED], n_gen)
    toks = ws.out_tokens.cpu()
    out = [toks[b, : n_gen[b]].tolist() for b in range(B)]
    return (out, ws, cache) if return_state else out


# Decode steps between two looks at the `finished` flags (a device-to-host synchronisation each: ~45 us of idle GPU).  Round 3
# (tools/ab_decode_loop.py, profiles/r03_decode_loop_cadence.txt): with the caller's stream no longer parked on the decode stream
# while the loop runs (generate_batch below) a batch-32 request takes 244.9 / 244.4 / 243.8 / 243.6 ms at a look every 4 / 16 / 32 /
# 64 steps; 16 keeps the wait for the last row's EOS short.
FINISH_CHECK_EVERY = 16
def _all_finished(ws) -> bool:
    """One look at the rows' `finished` flags: a reduction on the current stream and a synchronising read."""
    return bool(ws.finished.all())


ROW_FAILED = 2                # include/q3asr.h Q3ASR_ROW_FAILED: the device-side error flag in `finished`


def _raise_failed_rows(rows, n_gen) -> None:
    """A row whose logits had no valid maximum / draw (NaN or all -inf: a diverged or corrupt checkpoint, a fault upstream)
    stops on the device with finished = ROW_FAILED; the reference would silently go on decoding garbage (mx.argmax of NaN).
    Raised here so that it is never mistaken for a transcript."""
    if rows:
        from ._lib import Q3AsrError
        where = ", ".join(f"row {r} at generated token {max(0, int(n_gen[r]) - 1)}" for r in rows[:8])
        raise Q3AsrError(f"non-finite logits: {where} (the decoder produced no valid maximum; check the checkpoint)")


_RESERVED = object()          # RowScheduler: row being prefilled on the admission stream


class RowScheduler:
    """`rows` decode rows over one KV cache and one captured decode step; sequences are admitted into free rows (their prompts
    packed into one prefill that writes only those rows) and harvested when they end.  The building block of
    generate_continuous (a fixed list of sequences) and of serve.ContinuousTranscriber (a live queue).  All device work
    runs on the lane's side stream; call it from one thread."""

    def __init__(self, decoder: TextDecoder, rows: int, max_prompt: int, max_new: int, sampling: dict, check_every: int = 16):
        import torch
        self.dec, self.B = decoder, int(rows)
        if not 1 <= self.B <= 64:
            raise ValueError("RowScheduler handles 1..64 rows")
        self.max_prompt, self.max_new, self.check_every = int(max_prompt), int(max_new), int(check_every)
        self.Lmax = (self.max_prompt + self.max_new + 63) // 64 * 64
        self.cache = decoder.batch_cache(self.B, self.Lmax)
        self.ws = decoder.workspace(self.B, self.max_new)
        self.dev = decoder.device
        self.sampling = dict(sampling)
        self.owner: list = [None] * self.B           # key of the sequence in each row
        self.budget: list = [0] * self.B
        self.admit_logits = torch.empty((self.B, decoder.config.vocab_size), dtype=torch.bfloat16, device=self.dev)
        self.captured = False
        self.stats = {"steps": 0, "admissions": 0, "checks": 0, "row_steps_used": 0, "rows": self.B}
        self.side = _side_stream(self.dev)
        self.adm = torch.cuda.Stream(device=self.dev)          # admission prefills beside the decode steps (admit_async)
        self._adm = None                                       # (rows, items, event, row index tensor) of the one in flight
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.ws.finished.fill_(1)                # rows without a sequence stay inert
            self.ws.n_gen.zero_()
        self._fin = [1] * self.B
        self._ngen = [0] * self.B
        self._fresh = True                           # host copies of finished / n_gen match the device

    def free_rows(self) -> list:
        return [r for r in range(self.B) if self.owner[r] is None]

    def active(self) -> int:
        return sum(o is not None and o is not _RESERVED for o in self.owner)

    def admitting(self) -> bool:
        return self._adm is not None

    def admit_async(self, items: Sequence) -> None:
        """Like admit(), but the prefill runs on a second stream beside the decode steps of the running rows (MFMA-bound work
        beside latency-bound work).  The target rows are parked first (position -1: the decode attention returns at once for
        such a row, so no step in flight touches its cache while the prefill writes it); poll() activates them once the
        prefill has finished.  One admission in flight at a time."""
        import torch
        if self._adm is not None:
            raise RuntimeError("admit_async: an admission is already in flight")
        rows = self.free_rows()[: len(items)]
        if len(rows) < len(items):
            raise ValueError(f"admit_async: {len(items)} sequences for {len(rows)} free rows")
        seq_lens = [len(it[2]) for it in items]
        counts = [int(it[1].shape[0]) for it in items]
        budgets = [int(it[3]) for it in items]
        if max(seq_lens) > self.max_prompt or max(budgets) > self.max_new or min(budgets) < 1:
            raise ValueError(f"admit_async: prompt of {max(seq_lens)} tokens / budget {max(budgets)} exceeds this scheduler's {self.max_prompt} / {self.max_new}")
        with torch.cuda.stream(self.side):
            idx = torch.tensor(rows, dtype=torch.long, device=self.dev)
            self.ws.pos.index_fill_(0, idx, -1)
        self.adm.wait_stream(self.side)
        self.adm.wait_stream(torch.cuda.current_stream())        # the features were produced on the caller's stream
        with torch.cuda.stream(self.adm):
            packed = torch.cat([torch.as_tensor(it[1]).to(self.dev, torch.float32).reshape(c, -1) for it, c in zip(items, counts)])
            embeds = _merge(packed, [it[2] for it in items], counts, self.dec)
            self.dec.prefill(embeds, seq_lens, self.cache, logits_out=self.admit_logits, slots=rows)
            ev = torch.cuda.Event()
            ev.record(self.adm)
        for r in rows:
            self.owner[r] = _RESERVED
        self._adm = (rows, list(items), ev, idx, seq_lens, budgets)
        self.stats["admissions"] += len(items)
        self.stats["async_admissions"] = self.stats.get("async_admissions", 0) + 1
        self._fresh = False

    def poll(self, wait: bool = False):
        """Activate the admission in flight if its prefill has finished (or wait for it): counters reset, first tokens drawn on
        the decode stream.  Returns the items activated, or None."""
        import torch
        if self._adm is None:
            return None
        rows, items, ev, idx, seq_lens, budgets = self._adm
        if not wait and not ev.query():
            return None
        ws = self.ws
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=self.dev)      # noqa: E731
            ws.n_gen.index_fill_(0, idx, 0)
            ws.finished.index_fill_(0, idx, 0)
            ws.pos.index_copy_(0, idx, i32([t - 1 for t in seq_lens]))
            ws.max_new.index_copy_(0, idx, i32(budgets))
            for j, r in enumerate(rows):
                self.dec.sample_rows(ws, self.admit_logits[j], r, self.sampling)
        for r, it, b in zip(rows, items, budgets):
            self.owner[r], self.budget[r] = it[0], b
        self._adm = None
        self._fresh = False
        return items

    def admit(self, items: Sequence) -> None:
        """items: (key, feats fp32 [N][H], prompt ids, max_tokens) -- at most as many as there are free rows.  Prefill, reset
        the rows' counters, draw each first token (generate.py:278-286)."""
        import torch
        rows = self.free_rows()[: len(items)]
        if len(rows) < len(items):
            raise ValueError(f"admit: {len(items)} sequences for {len(rows)} free rows")
        dec, ws = self.dec, self.ws
        seq_lens = [len(it[2]) for it in items]
        counts = [int(it[1].shape[0]) for it in items]
        budgets = [int(it[3]) for it in items]
        if max(seq_lens) > self.max_prompt or max(budgets) > self.max_new or min(budgets) < 1:
            raise ValueError(f"admit: prompt of {max(seq_lens)} tokens / budget {max(budgets)} exceeds this scheduler's {self.max_prompt} / {self.max_new}")
        self.side.wait_stream(torch.cuda.current_stream())       # the features were produced on the caller's stream
        with torch.cuda.stream(self.side):
            packed = torch.cat([torch.as_tensor(it[1]).to(self.dev, torch.float32).reshape(c, -1) for it, c in zip(items, counts)])
            embeds = _merge(packed, [it[2] for it in items], counts, dec)
            full = rows == list(range(self.B))
            logits = ws.logits if full else self.admit_logits
            dec.prefill(embeds, seq_lens, self.cache, logits_out=logits, slots=None if full else rows)
            idx = torch.tensor(rows, dtype=torch.long, device=self.dev)
            i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=self.dev)      # noqa: E731
            ws.n_gen.index_fill_(0, idx, 0)
            ws.finished.index_fill_(0, idx, 0)
            ws.pos.index_copy_(0, idx, i32([t - 1 for t in seq_lens]))
            ws.max_new.index_copy_(0, idx, i32(budgets))
            if full:
                dec.sample_step(ws, ws.logits, self.sampling)
            else:
                for j, r in enumerate(rows):
                    dec.sample_rows(ws, logits[j], r, self.sampling)
        for r, it, b in zip(rows, items, budgets):
            self.owner[r], self.budget[r] = it[0], b
        self.stats["admissions"] += len(items)
        self._fresh = False

    def _look(self) -> None:
        import torch
        if not self._fresh:
            with torch.cuda.stream(self.side):
                self._fin, self._ngen = self.ws.finished.tolist(), self.ws.n_gen.tolist()       # synchronises the side stream
            self.stats["checks"] += 1
            self._fresh = True

    def harvest(self) -> list:
        """[(key, token ids)] of the sequences that have ended (EOS or budget); their rows become free.  A sequence whose row
        raised the device-side error flag (non-finite logits) comes back as (key, Q3AsrError) instead of ids."""
        import torch
        self._look()
        ended = [r for r in range(self.B) if self.owner[r] is not None and self.owner[r] is not _RESERVED and self._fin[r]]
        if not ended:
            return []
        with torch.cuda.stream(self.side):
            toks = self.ws.out_tokens[torch.tensor(ended, dtype=torch.long, device=self.dev)].cpu()
        out = []
        for j, r in enumerate(ended):
            if self._fin[r] == ROW_FAILED:                   # device-side error flag: this sequence fails, its neighbours go on
                from ._lib import Q3AsrError
                out.append((self.owner[r], Q3AsrError(f"non-finite logits at generated token {max(0, self._ngen[r] - 1)} "
                                                      "(the decoder produced no valid maximum; check the checkpoint)")))
                self.owner[r] = None
                continue
            out.append((self.owner[r], toks[j, : self._ngen[r]].tolist()))
            self.stats["row_steps_used"] += self._ngen[r]
            self.owner[r] = None
        return out

    def run(self) -> int:
        """Decode until the next look: check_every steps, or fewer if a row must end by budget before that."""
        import torch
        self._look()
        act = [r for r in range(self.B) if self.owner[r] is not None and self.owner[r] is not _RESERVED and not self._fin[r]]
        if not act:
            return 0
        k = max(1, min(self.check_every, min(self.budget[r] - self._ngen[r] for r in act)))
        done = 0
        with torch.cuda.stream(self.side):
            if not self.captured:
                self.dec.decode_step(self.ws, self.cache, self.sampling)        # eager: loads code objects, a real step
                self.dec.capture_step(self.ws, self.cache, self.sampling)
                self.captured = True
                done = 1
            while done < k:
                done += self.dec.replay_step(self.ws, multi=k - done >= self.dec.MULTI_STEPS)
        self.stats["steps"] += done
        self._fresh = False
        return done

    def finish(self) -> None:
        import torch
        torch.cuda.current_stream().wait_stream(self.side)


def generate_continuous(decoder: TextDecoder, feats: Sequence, prompts: Sequence[Sequence[int]], max_tokens: Sequence[int],
                        slots: int = 32, temperature: float = 0.0, top_p: float = 1.0, top_k: int = 0,
                        repetition_penalty: float = 1.0, repetition_context_size: int = 100, seed: int = 0,
                        check_every: int = 16, min_admit: Optional[int] = None, stats: Optional[dict] = None,
                        async_admit: bool = False) -> list[list[int]]:
    """Any number of sequences through `slots` decode rows: a row whose sequence has ended (EOS or budget) is handed to the
    next waiting sequence, whose prompt is prefilled into that row of the KV cache while the other rows keep their state
    (the reference decodes one utterance at a time, model.py:239).  `feats[i]`: encoder output
    fp32 [N_i][H] of sequence i.  Sequences are admitted in the order given.  Greedy results equal generate_batch's
    (and the reference loop's) token for token: a row's arithmetic does not depend on its neighbours, its row index or the
    cache capacity.  With temperature > 0 the draws depend on the row a sequence lands in (reproducible, but not equal
    to a standalone call).  `min_admit`: free rows to wait for before prefilling (one packed prefill of several prompts
    costs little more than a prefill of one; default rows / 8), ignored once nothing else is running.  `async_admit`: run
    the admission prefills on a second stream beside the decode steps (RowScheduler.admit_async) -- measured neutral at
    min_admit = rows / 8 and +30 % when rows are refilled one at a time (tools/bench_continuous.py)."""
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
    sampling = {"temperature": float(temperature), "top_p": float(top_p), "top_k": int(top_k),
                "repetition_penalty": float(repetition_penalty),
                "repetition_context_size": int(repetition_context_size), "seed": int(seed)}
    sch = RowScheduler(decoder, B, max(len(p) for p in prompts), max(mt), sampling, check_every)
    waiting = deque(range(n))
    out: list = [None] * n
    while True:
        for key, toks in sch.harvest():
            if isinstance(toks, Exception):
                sch.finish()
                raise toks
            out[key] = toks
        sch.poll(wait=not sch.active())                                    # an admission in flight: activate it when it is done
        free = sch.free_rows()
        if waiting and free and not sch.admitting() and (len(free) >= min(min_admit, len(waiting)) or not sch.active()):
            take = [waiting.popleft() for _ in range(min(len(free), len(waiting)))]
            group = [(u, feats[u], prompts[u], mt[u]) for u in take]
            if async_admit and sch.active():
                sch.admit_async(group)                                     # prefill beside the running rows' decode steps
            else:
                sch.admit(group)
                continue                                                   # a budget of 1 ends at the first token
        if sch.active():
            sch.run()
        elif not sch.admitting():
            break
    sch.finish()
    if stats is not None:
        stats.update(sch.stats)
    return out


_streams: dict = {}


def _side_stream(dev, chain: int = 0):
    import torch
    from .decoder import current_lane
    key = (str(dev), current_lane(), chain)
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
