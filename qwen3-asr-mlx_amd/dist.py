"""Utterance-level data parallelism over the GPUs of one node (new: the reference is single-device).

One process per GPU, full model replica each (4.08 GB of 288 GB).  Utterances share no state, so the
only communication is the batch split and the result gather: rank 0 scatters padded float32 audio
(1.92 MB per 30 s clip) and gathers int32 token ids -- `torch.distributed` scatter/gather, i.e. RCCL
over xGMI with backend "nccl", gloo on CPU in tests.  No all-reduce exists anywhere on the path.
The scattered audio stays on `device`: the per-rank run function receives the padded device tensor.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np

_ERR_CHARS = 240          # bytes of a failing rank's message carried through the status gather


def assign(lengths: Sequence[int], world: int) -> list[list[int]]:
    """Utterance indices per rank: longest first, dealt round-robin (balances ragged batches;
    equal-length clips give B/world each)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    return [order[r::world] for r in range(world)]


def run_sharded(run_fn: Callable, audios, max_tokens: Optional[Sequence[int]], device="cpu", group=None) -> Optional[list]:
    """Rank 0 passes the whole request, other ranks pass None for `audios` and `max_tokens`.  `audios` is either a sequence
    of 1-D float32 arrays (host) or a pair ``(padded fp32 tensor [n][l] on `device`, lengths)`` already resident there.
    Every rank calls ``run_fn(audio [m][l_max] fp32 on `device` (zero padded), lens list[int], max_tokens list[int])
    -> list[list[int]]`` on its share (no host bounce of the audio).  Rank 0 returns the token lists in request order, the
    others None.  The result is identical to one rank running the whole request whenever run_fn is batch-invariant (the HIP
    path is).  An exception inside run_fn on any rank is carried through the gather: every rank completes the
    collectives, rank 0 raises RuntimeError naming the failing ranks, the failing ranks re-raise their own exception.
    A malformed request (rank 0's validation) travels in the header broadcast, so every rank raises ValueError together
    and none is left waiting in a collective."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    bad_request = None
    if rank == 0:
        try:
            resident = isinstance(audios, tuple) and len(audios) == 2 and hasattr(audios[0], "shape") and not isinstance(audios[0], np.ndarray)
            lens = [int(n) for n in audios[1]] if resident else [len(a) for a in audios]
            if max_tokens is None or len(max_tokens) != len(lens):
                raise ValueError("run_sharded: one max_tokens entry per utterance")
            if resident and (audios[0].ndim != 2 or audios[0].shape[0] != len(lens) or (lens and max(lens) > audios[0].shape[1])):
                raise ValueError("run_sharded: resident audio must be [n][l] with every length <= l")
            if any(int(m) < 1 for m in max_tokens):
                raise ValueError("run_sharded: max_tokens must be >= 1")
            parts = assign(lens, world)
            n_max = max(len(p) for p in parts)
            l_max = max(lens) if lens else 0
            hdr = torch.tensor([n_max, l_max, max(max_tokens) if len(lens) else 0], dtype=torch.int64, device=device)
        except Exception as e:                                # noqa: BLE001 -- the other ranks are already in the broadcast
            bad_request = e
            hdr = torch.tensor([-1, 0, 0], dtype=torch.int64, device=device)
    dist.broadcast(hdr, 0, group=group)
    n_max, l_max, t_max = (int(v) for v in hdr.tolist())
    if n_max < 0:
        if bad_request is not None:
            raise bad_request if isinstance(bad_request, ValueError) else ValueError(f"run_sharded: {bad_request}")
        raise ValueError("run_sharded: rank 0 rejected the request")
    if n_max == 0:
        return [] if rank == 0 else None
    my_audio = torch.zeros((n_max, l_max), dtype=torch.float32, device=device)
    my_meta = torch.full((n_max, 2), -1, dtype=torch.int64, device=device)          # (samples, max_tokens)
    if rank == 0:
        a_list, m_list = [], []
        for p in parts:
            m = torch.full((n_max, 2), -1, dtype=torch.int64)
            for j, i in enumerate(p):
                m[j, 0], m[j, 1] = lens[i], int(max_tokens[i])
            if resident:                                    # rows picked on the device: the request never visits the host
                src = audios[0]
                a = torch.zeros((n_max, l_max), dtype=torch.float32, device=device)
                if p:
                    a[: len(p), : min(l_max, src.shape[1])] = src.to(device)[torch.tensor(p, dtype=torch.long, device=device), :l_max]
            else:
                a = torch.zeros((n_max, l_max), dtype=torch.float32)
                for j, i in enumerate(p):
                    a[j, : lens[i]] = torch.from_numpy(np.ascontiguousarray(audios[i], dtype=np.float32))
                a = a.to(device)
            a_list.append(a)
            m_list.append(m.to(device))
        dist.scatter(my_audio, a_list, src=0, group=group)
        dist.scatter(my_meta, m_list, src=0, group=group)
    else:
        dist.scatter(my_audio, None, src=0, group=group)
        dist.scatter(my_meta, None, src=0, group=group)
    meta = my_meta.cpu()
    mine = [j for j in range(n_max) if int(meta[j, 0]) >= 0]                         # a prefix: rows are dealt in order
    status = torch.zeros((1 + _ERR_CHARS,), dtype=torch.int32)
    toks, failure = [], None
    if mine:
        try:
            toks = run_fn(my_audio[: len(mine)], [int(meta[j, 0]) for j in mine], [int(meta[j, 1]) for j in mine])
            if len(toks) != len(mine):
                raise RuntimeError(f"run_fn returned {len(toks)} results for {len(mine)} utterances")
        except Exception as e:                                # noqa: BLE001 -- must reach the collectives below
            failure, toks = e, []
            msg = f"{type(e).__name__}: {e}".encode("utf-8", "replace")[:_ERR_CHARS]
            status[0] = 1
            status[1: 1 + len(msg)] = torch.tensor(list(msg), dtype=torch.int32)
    out = torch.full((n_max, t_max + 1), -1, dtype=torch.int32)
    for j, t in zip(mine, toks):
        out[j, 0] = len(t)
        out[j, 1: 1 + len(t)] = torch.tensor(t, dtype=torch.int32)
    out, status = out.to(device), status.to(device)
    gathered = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
    stat_all = [torch.empty_like(status) for _ in range(world)] if rank == 0 else None
    dist.gather(out, gathered, dst=0, group=group)
    dist.gather(status, stat_all, dst=0, group=group)
    if rank != 0:
        if failure is not None:
            raise failure
        return None
    bad = []
    for r in range(world):
        s = stat_all[r].cpu()
        if int(s[0]):
            bad.append(f"rank {r}: " + bytes(int(c) for c in s[1:].tolist() if c).decode("utf-8", "replace"))
    if bad:
        raise RuntimeError("run_sharded: " + "; ".join(bad)) from failure
    result: list = [None] * len(lens)
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
    and the jobs, not the utterances, are dealt to the ranks, so a single long recording spreads over every GPU.  A rank's
    share may hold any number of jobs (`generate_tokens_packed` walks it in batches of MAX_BATCH).
    Returns the `TranscriptionResult`s on rank 0 (None elsewhere); equal to the single-process result."""
    import torch.distributed as dist
    from .model import TranscriptionResult, plan_jobs
    rank = dist.get_rank(group)
    lang = model._resolve_language(language)
    jobs = seg_of = samples = None
    if rank == 0:
        samples = [model._as_samples(a) for a in audios]
        jobs, seg_of = plan_jobs(samples, chunk_duration, max_tokens)
    toks = run_sharded(lambda a, n, m: model.generate_tokens_packed(a, n, m, lang, **sampling),
                       [j[1] for j in jobs] if rank == 0 else None, [j[2] for j in jobs] if rank == 0 else None,
                       device=device, group=group)
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
