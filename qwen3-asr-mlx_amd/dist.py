"""Utterance-level data parallelism over the GPUs of one node (new: the reference is single-device).

One process per GPU, full model replica each (4.08 GB of 288 GB).  Utterances share no state, so the
only communication is the batch split and the result gather: rank 0 scatters padded float32 audio
(1.92 MB per 30 s clip) and gathers int32 token ids -- `torch.distributed` scatter/gather, i.e. RCCL
over xGMI with backend "nccl", gloo on CPU in tests.  No all-reduce exists anywhere on the path.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np


def assign(lengths: Sequence[int], world: int) -> list[list[int]]:
    """Utterance indices per rank: longest first, dealt round-robin (balances ragged batches;
    equal-length clips give B/world each)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    return [order[r::world] for r in range(world)]


def run_sharded(run_fn: Callable[[list, list], list], audios: Optional[Sequence[np.ndarray]],
                max_tokens: Optional[Sequence[int]], device="cpu", group=None) -> Optional[list]:
    """Rank 0 passes the whole request (`audios`, `max_tokens`); other ranks pass None.  Every rank
    calls `run_fn(its_audios, its_max_tokens) -> list[list[int]]`.  Rank 0 returns the token lists in
    request order, the others None.  Result is identical to `run_fn(audios, max_tokens)` on one rank
    whenever run_fn is batch-invariant (the HIP path is)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == 0:
        lens = [len(a) for a in audios]
        parts = assign(lens, world)
        n_max = max(len(p) for p in parts)
        l_max = max(lens) if lens else 0
        hdr = torch.tensor([n_max, l_max, max(max_tokens) if len(lens) else 0], dtype=torch.int64, device=device)
    dist.broadcast(hdr, 0, group=group)
    n_max, l_max, t_max = (int(v) for v in hdr.tolist())
    if n_max == 0:
        return [] if rank == 0 else None
    my_audio = torch.zeros((n_max, l_max), dtype=torch.float32, device=device)
    my_meta = torch.full((n_max, 2), -1, dtype=torch.int64, device=device)          # (samples, max_tokens)
    if rank == 0:
        a_list, m_list = [], []
        for p in parts:
            a = torch.zeros((n_max, l_max), dtype=torch.float32)
            m = torch.full((n_max, 2), -1, dtype=torch.int64)
            for j, i in enumerate(p):
                a[j, : lens[i]] = torch.from_numpy(np.ascontiguousarray(audios[i], dtype=np.float32))
                m[j, 0], m[j, 1] = lens[i], int(max_tokens[i])
            a_list.append(a.to(device))
            m_list.append(m.to(device))
        dist.scatter(my_audio, a_list, src=0, group=group)
        dist.scatter(my_meta, m_list, src=0, group=group)
    else:
        dist.scatter(my_audio, None, src=0, group=group)
        dist.scatter(my_meta, None, src=0, group=group)
    meta = my_meta.cpu()
    mine = [j for j in range(n_max) if int(meta[j, 0]) >= 0]
    host = my_audio.cpu().numpy()
    toks = run_fn([host[j, : int(meta[j, 0])] for j in mine], [int(meta[j, 1]) for j in mine]) if mine else []
    out = torch.full((n_max, t_max + 1), -1, dtype=torch.int32)
    for j, t in zip(mine, toks):
        out[j, 0] = len(t)
        out[j, 1: 1 + len(t)] = torch.tensor(t, dtype=torch.int32)
    out = out.to(device)
    gathered = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
    dist.gather(out, gathered, dst=0, group=group)
    if rank != 0:
        return None
    result: list = [None] * len(audios)
    for r, p in enumerate(parts):
        g = gathered[r].cpu()
        for j, i in enumerate(p):
            n = int(g[j, 0])
            result[i] = g[j, 1: 1 + n].tolist()
    return result


def transcribe_sharded(model, audios: Optional[Sequence], language: Optional[str] = None, max_tokens: Optional[int] = None,
                       chunk_duration: float = 1200.0, device="cpu", group=None, **sampling) -> Optional[list]:
    """`Qwen3ASR.transcribe_batch` over all ranks of `group`: rank 0 passes the request (others `audios=None`), plans the
    decode jobs -- one per utterance, one per low-energy-split segment of long-form audio (reference model.py:382-447) --
    and the jobs, not the utterances, are dealt to the ranks, so a single long recording spreads over every GPU.
    Returns the `TranscriptionResult`s on rank 0 (None elsewhere); equal to the single-process result."""
    import torch.distributed as dist
    from .model import TranscriptionResult, plan_jobs
    rank = dist.get_rank(group)
    lang = model._resolve_language(language)
    jobs = seg_of = samples = None
    if rank == 0:
        samples = [model._as_samples(a) for a in audios]
        jobs, seg_of = plan_jobs(samples, chunk_duration, max_tokens)
    toks = run_sharded(lambda a, m: model.generate_tokens(a, m, lang, **sampling), [j[1] for j in jobs] if rank == 0 else None,
                       [j[2] for j in jobs] if rank == 0 else None, device=device, group=group)
    if rank != 0:
        return None
    texts = [model._decode_output(t) for t in toks]
    out = []
    for i, s in enumerate(samples):
        if len(s) == 0:
            out.append(TranscriptionResult(text="", language="Unknown", duration=0.0))
            continue
        parts = [texts[j] for j in seg_of[i]]
        long_form = len(s) / 16_000 > chunk_duration
        out.append(TranscriptionResult(text=" ".join(t for t in parts if t) if long_form else parts[0], language=lang,
                                       duration=len(s) / 16_000))
    return out
