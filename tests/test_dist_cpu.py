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
        h = int(np.abs(a).sum() * 1e4) % 1000
        out.append([(h + 7 * i) % 151936 for i in range(min(m, 3 + len(a) % 5))])
    return out


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
    res = qd.run_sharded(_fake_run, audios if rank == 0 else None, budgets if rank == 0 else None)
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
