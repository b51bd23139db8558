"""A/B of the sampler's two forms (q3asr_set_option(Q3ASR_OPT_SAMPLER_RESIDENT, 1 | 0)): us per q3asr_sample_step launch on
[B][151,936] bf16 logits ~ N(0, 3^2) with a 100-token penalty window, greedy and temperature 0.7 / top-p 0.9, interleaved rounds.
Usage: python tools/ab_sampler.py [B=40] [rounds=5] [reps=40]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qwen3_asr_mlx_amd import _lib, ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    dev = torch.device("cuda:0")
    lib = _lib.load()
    V = 151936
    torch.manual_seed(0)
    logits = (torch.randn(B, V, device=dev) * 3.0).to(torch.bfloat16)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
    n0 = 100
    hist = torch.randint(0, V, (B, n0 + reps + 8), dtype=torch.int32, device=dev)
    res = {}
    for cfg_name, (t, p) in (("greedy", (0.0, 1.0)), ("t0.7_p0.9", (0.7, 0.9))):
        for form in (1, 0):
            res[(cfg_name, form)] = []
        for _ in range(rounds):
            for form in (1, 0):
                lib.q3asr_set_option(4, form)
                out_tokens = hist.clone()
                n_gen, fin, cur, pos, mx = i32([n0] * B), i32([0] * B), i32([0] * B), i32([0] * B), i32([n0 + reps + 8] * B)

                def step():
                    ops.call("q3asr_sample_step", ops.ptr(logits), V, B, V, ops.ptr(out_tokens), n0 + reps + 8, ops.ptr(n_gen), ops.ptr(fin),
                             ops.ptr(cur), ops.ptr(pos), ops.ptr(mx), t, p, 0, 1.2, 100, 7, -1, -1, 0, ops.sp())
                step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    step()
                e1.record()
                e1.synchronize()
                res[(cfg_name, form)].append(1e3 * e0.elapsed_time(e1) / reps)
        lib.q3asr_set_option(4, 1)
    for (cfg_name, form), v in res.items():
        v = sorted(v)
        print(f"B={B} {cfg_name:10s} {'resident ' if form else 'streaming'}: median {v[len(v) // 2]:7.2f} us   (rounds: {' '.join(f'{x:.1f}' for x in v)})")


if __name__ == "__main__":
    main()
