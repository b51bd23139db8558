#!/usr/bin/env python3
"""In-situ A/B of launch forms inside the captured decode step (q3asr_set_option): for every arm a graph of MULTI_STEPS steps is
captured with the arm's switches on ONE decode state, and the graphs are replayed in interleaved rounds on one stream (HIP events;
every round starts from the same prefilled state).  That the forms give identical results is what the kernel tests check.  Arms: comma list of name=opt:val[+opt:val], e.g.
    base=1:0,head=1:1      (option ids: include/q3asr.h)
Usage: python3 tools/ab_step.py [batch] [arms] [rounds]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import _lib, synth  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.generate import _merge  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402
from signals import signal  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
arms = []
for a in (sys.argv[2] if len(sys.argv) > 2 else "base=1:0,head=1:1").split(","):
    name, sw = a.split("=")
    arms.append((name, [tuple(int(v) for v in s.split(":")) for s in sw.split("+")]))
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 9
dev = torch.device("cuda:0")
cfg = ModelConfig()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
enc, dec = model._encoder, model._decoder
sampling = dict(temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, repetition_context_size=100)
audio, lens_dev, lens = q.audio.pack_audio([signal("speechlike", 16000 * 30, seed=1234 + i) for i in range(B)], dev)
mel, n_frames = q.audio.get_frontend(dev)(audio, lens_dev, lens)
feats, counts = enc.forward_batch(mel, n_frames)
prompts = [build_prompt(c, [6364]) for c in counts]
seq = [len(p) for p in prompts]
emb = _merge(feats, prompts, counts, dec)
steps = dec.MULTI_STEPS
N_REPLAY = 24                                    # replays per timed round: 24 x MULTI_STEPS tokens from the prefilled state


def fresh():
    cache = dec.batch_cache(B, (max(seq) + 512 + 63) // 64 * 64)
    ws = dec.workspace(B, 512)
    ws.n_gen.zero_(); ws.finished.zero_()
    ws.pos.copy_(torch.tensor([t - 1 for t in seq], dtype=torch.int32))
    ws.max_new.copy_(torch.tensor([512] * B, dtype=torch.int32))
    dec.prefill(emb, seq, cache, logits_out=ws.logits)
    dec.sample_step(ws, ws.logits, sampling)
    return ws, cache


side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ws, cache = fresh()                          # one state for all arms: the arms do the same work on it in turn
    dec.decode_step(ws, cache, sampling)
    graphs = {}
    for name, sw in arms:
        for o, v in sw:
            _lib.load().q3asr_set_option(o, v)
        graphs[name] = dec._capture(ws, cache, sampling, steps)
    for o in (1, 2, 3, 4, 5):
        _lib.load().q3asr_set_option(o, 1)
    ts = {name: [] for name, _ in arms}
    for rnd in range(rounds + 1):
        for name, _ in (arms if rnd % 2 == 0 else arms[::-1]):
            fresh()                                                # every round from the same state (same buffers: the graphs stay valid)
            side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(N_REPLAY):
                _lib.call("q3asr_graph_launch", graphs[name], torch.cuda.current_stream().cuda_stream)
            e1.record()
            side.synchronize()
            if rnd:
                ts[name].append(e0.elapsed_time(e1) * 1000 / (N_REPLAY * steps))
    side.synchronize()
    if any(int(f) == 2 for f in ws.finished.tolist()):
        raise SystemExit("a row failed (non-finite logits)")
    for g in graphs.values():
        _lib.call("q3asr_graph_destroy", g)
for name, _ in arms:
    v = sorted(ts[name])
    print(f"{name:10s} {v[len(v) // 2]:8.1f} us per step (min {v[0]:.1f}, max {v[-1]:.1f}; {len(v)} rounds of {N_REPLAY * steps} steps)", flush=True)
