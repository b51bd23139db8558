"""The data-parallel scatter/gather path on CPU: 2 and 3 gloo ranks must reproduce the
single-process result exactly (token ids), including ragged lengths and empty ranks."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _fake_run(audios, budgets):
    """Deterministic stand-in for the per-GPU hot path: ids derived from the samples only."""
    out = []
    for a, m in zip(audios, budgets):
        a = np.asarray(a)
        h = int(np.abs(a).sum() * 1e4) % 1000
        out.append([(h + 7 * i) % 151936 for i in range(min(m, 3 + len(a) % 5))])
    return out


def _fake_run_packed(audio, lens, budgets):
    """The run function `run_sharded` calls: padded tensor [m][l_max] on the group's device + lengths."""
    assert isinstance(audio, torch.Tensor) and audio.shape[0] == len(lens) == len(budgets)
    return _fake_run([audio[j, :n].numpy() for j, n in enumerate(lens)], budgets)


def _request():
    rng = np.random.default_rng(0)
    audios = [rng.standard_normal(n).astype(np.float32) for n in (4000, 160, 9001, 2500, 777)]
    return audios, [4, 9, 2, 6, 5]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qwen3_asr_mlx_amd import dist as qd
    audios, budgets = _request()
    res = qd.run_sharded(_fake_run_packed, audios if rank == 0 else None, budgets if rank == 0 else None)
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_process(world):
    audios, budgets = _request()
    want = _fake_run(audios, budgets)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want


def test_assignment_balances_and_covers():
    from qwen3_asr_mlx_amd import dist as qd
    parts = qd.assign([5, 9, 1, 7, 7, 3], 4)
    assert sorted(i for p in parts for i in p) == list(range(6))
    assert parts[0][0] == 1 and [len(p) for p in parts] == [2, 2, 1, 1]
    assert [len(p) for p in qd.assign([480000] * 256, 8)] == [32] * 8
    assert qd.assign([], 2) == [[], []]


class _StandInModel:
    """Duck-typed Qwen3ASR for the host logic of transcribe_sharded: tokens are a function of the samples only."""

    def _resolve_language(self, language):
        return "English" if language is None else language

    @staticmethod
    def _as_samples(a):
        return np.asarray(a, dtype=np.float32)

    MAX_BATCH = 64

    def generate_tokens_packed(self, audio, lens, budgets, lang, **kw):
        out = []
        for s0 in range(0, len(lens), self.MAX_BATCH):                 # as Qwen3ASR.generate_tokens_packed: batches of MAX_BATCH
            part = lens[s0: s0 + self.MAX_BATCH]
            assert len(part) <= self.MAX_BATCH
            out += _fake_run_packed(audio[s0: s0 + len(part)], part, budgets[s0: s0 + len(part)])
        return out

    def _decode_output(self, toks):
        return "-".join(str(t) for t in toks)


def _long_request():
    rng = np.random.default_rng(5)
    env = lambda n: (0.05 + np.abs(np.sin(np.arange(n) / 16000.0 * 2.1))).astype(np.float32)      # noqa: E731
    return [rng.standard_normal(n).astype(np.float32) * env(n) for n in (16000 * 70, 3000, 0, 16000 * 31)]


def _worker_longform(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qwen3_asr_mlx_amd import dist as qd
    res = qd.transcribe_sharded(_StandInModel(), _long_request() if rank == 0 else None, chunk_duration=20.0)
    if rank == 0:
        q.put([(r.text, r.language, r.duration) for r in res])
    dist.barrier()
    dist.destroy_process_group()


def test_longform_segments_are_dealt_across_ranks():
    """transcribe_sharded: segments of long recordings (20 s chunks here) go to different ranks; texts are re-joined in
    order and equal the single-process plan + run."""
    from qwen3_asr_mlx_amd.model import plan_jobs
    audios = _long_request()
    m = _StandInModel()
    jobs, seg_of = plan_jobs(audios, 20.0, None)
    assert len(seg_of[0]) >= 3 and len(seg_of[3]) == 2 and len(seg_of[1]) == 1 and 2 not in seg_of
    texts = [m._decode_output(t) for t in _fake_run([j[1] for j in jobs], [j[2] for j in jobs])]
    want = []
    for i, a in enumerate(audios):
        if len(a) == 0:
            want.append(("", "Unknown", 0.0))
        else:
            parts = [texts[j] for j in seg_of[i]]
            want.append((" ".join(parts) if len(a) / 16000 > 20.0 else parts[0], "English", len(a) / 16000))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker_longform, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want


def _worker_many(rank, world, port, q, mode):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qwen3_asr_mlx_amd import dist as qd
    rng = np.random.default_rng(11)
    n = 150                                                            # 75 jobs per rank: more than MAX_BATCH
    audios = [rng.standard_normal(200 + 3 * i).astype(np.float32) for i in range(n)]
    budgets = [3 + i % 5 for i in range(n)]
    try:
        if mode == "resident":                                         # request already packed on the group's device
            lens = [len(a) for a in audios]
            packed = torch.zeros((n, max(lens)))
            for i, a in enumerate(audios):
                packed[i, : lens[i]] = torch.from_numpy(a)
            res = qd.run_sharded(_fake_run_packed, (packed, lens) if rank == 0 else None, budgets if rank == 0 else None)
            if rank == 0:
                q.put(("ok", res == _fake_run(audios, budgets)))
        elif mode == "many":
            res = qd.transcribe_sharded(_StandInModel(), audios if rank == 0 else None, max_tokens=4)
            if rank == 0:
                m = _StandInModel()
                q.put(("ok", [r.text for r in res] == [m._decode_output(t) for t in _fake_run(audios, [4] * n)]))
        elif mode == "badreq":                                         # rank 0's validation fails BEFORE the first collective
            try:
                qd.run_sharded(_fake_run_packed, audios if rank == 0 else None, budgets[:-1] if rank == 0 else None)
                q.put((rank, "no error"))
            except Exception as e:                                     # noqa: BLE001
                q.put((rank, f"{type(e).__name__}: {e}"))
            res = qd.run_sharded(_fake_run_packed, audios if rank == 0 else None, budgets if rank == 0 else None)   # the group is still usable
            if rank == 0:
                q.put(("after", res == _fake_run(audios, budgets)))
        else:                                                          # one rank's run function raises: nobody may hang
            def run(audio, lens, b):
                if rank == 1:
                    raise ValueError("clip shorter than one hop")
                return _fake_run_packed(audio, lens, b)
            try:
                qd.run_sharded(run, audios if rank == 0 else None, budgets if rank == 0 else None)
                q.put((rank, "no error"))
            except Exception as e:                                     # noqa: BLE001
                q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(target, world, port, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    return q, procs


@pytest.mark.parametrize("mode", ["many", "resident"])
def test_more_jobs_per_rank_than_max_batch(mode):
    """a rank's share above MAX_BATCH used to raise inside generate_batch on that rank while the others
    waited in the gather."""
    q, procs = _spawn(_worker_many, 2, 29400 + (os.getpid() % 90) + (0 if mode == "many" else 100), mode)
    assert q.get(timeout=180) == ("ok", True)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_failing_rank_does_not_deadlock_the_group():
    q, procs = _spawn(_worker_many, 2, 29800 + (os.getpid() % 90), "fail")
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] == "ValueError: clip shorter than one hop"
    assert got[0].startswith("RuntimeError: run_sharded: rank 1: ValueError: clip shorter than one hop")


def test_malformed_request_raises_on_every_rank_without_a_hang():
    """rank 0's request validation used to raise before the header broadcast while the other ranks were
    already waiting in it.  The verdict now travels in the header: all ranks raise ValueError, and the group goes on."""
    q, procs = _spawn(_worker_many, 2, 29900 + (os.getpid() % 90), "badreq")
    got = dict(q.get(timeout=180) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == "ValueError: run_sharded: one max_tokens entry per utterance"
    assert got[1] == "ValueError: run_sharded: rank 0 rejected the request"
    assert got["after"] is True
