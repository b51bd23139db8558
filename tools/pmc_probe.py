#!/usr/bin/env python3
"""Minimal program for `rocprofv3 --pmc ...` / `--kernel-trace --stats` runs (placed directly after `--`): no helper
threads, no side streams, no graph capture unless asked for.  Runs, eagerly, on the legacy-free current stream:

  decode   3 x the weight-streaming GEMM launches of one batch-32 decode step (decoder.gemm_only_step)
  step     2 x one whole eager decode step at batch 32 (adds attention, residual+norm, sampling)
  encoder  one mel + encoder pass over 32 x 30 s clips
  prefill  one prefill of those 32 prompts
  batch1   4 eager decode steps at batch 1 (the fused residual+norm GEMM path, 144 KiB of dynamic LDS)
  graph    capture the batch-32 decode step into a hipGraph and replay it 4 times
  graph4   the same with the 4-step graph
  regraph  capture, replay, DESTROY the graph, capture again, replay (what bench.py's stage leg does to a workspace that
           already holds a graph)
  finite   one eager batch-32 decode step with every launch followed by a synchronise and a finiteness check of the
           buffers it may have written (slab poisoned with NaN first, so a read of an unwritten slab element shows too);
           names the first launch that leaves a non-finite value.  Every mode ends with a check of the device-side error
           flag (finished == Q3ASR_ROW_FAILED) and of the last logits.

Usage: python3 tools/pmc_probe.py decode,encoder [--batch 32] [--seconds 30]
Writes gpurun_out/pmc_probe_meta.json: sha-256 (first 16 hex) of the library that ran, and what ran."""
import hashlib
import json
import sys
import time
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

what = set((sys.argv[1] if len(sys.argv) > 1 else "decode,encoder").split(","))
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 32
secs = float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else 30.0
dev = torch.device("cuda:0")
cfg = ModelConfig()
model = q.Qwen3ASR.from_state_dict(cfg, synth.random_state(cfg, seed=0, device=dev), synth.SyntheticTokenizer(), dev)
dec, enc = model._decoder, model._encoder
t0 = time.perf_counter()


def say(msg):
    print(f"[probe {time.perf_counter() - t0:6.1f}s] {msg}", file=sys.stderr, flush=True)


sampling = {"temperature": 0.0, "top_p": 1.0, "top_k": 0, "repetition_penalty": 1.2, "repetition_context_size": 100, "seed": 0}
n = int(secs * 16000)
feats = counts = None
if what & {"encoder", "prefill", "step", "graph", "graph4", "batch1", "regraph"}:
    audio, lens_dev, lens = q.audio.pack_audio([signal("speechlike", n, seed=1234 + i) for i in range(B)], dev)
    mel, n_frames = q.audio.get_frontend(dev)(audio, lens_dev, lens)
    feats, counts = enc.forward_batch(mel, n_frames)
    torch.cuda.synchronize()
    say("encoder pass done")


def prefilled(b):
    prompts = [build_prompt(c, [6364]) for c in counts[:b]]
    seq = [len(p) for p in prompts]
    cache = dec.batch_cache(b, (max(seq) + 64 + 63) // 64 * 64)
    ws = dec.workspace(b, 64)
    ws.n_gen.zero_(); ws.finished.zero_()
    ws.pos.copy_(torch.tensor([t - 1 for t in seq], dtype=torch.int32))
    ws.max_new.copy_(torch.tensor([64] * b, dtype=torch.int32))
    emb = _merge(feats[: sum(counts[:b])], prompts, counts[:b], dec)
    dec.prefill(emb, seq, cache, logits_out=ws.logits)
    dec.sample_step(ws, ws.logits, sampling)
    torch.cuda.synchronize()
    return ws, cache


if "decode" in what:
    ws = dec.workspace(B, 64)
    for _ in range(3):
        dec.gemm_only_step(ws)
    torch.cuda.synchronize()
    say("decode GEMMs done")
def check_state(ws, where):
    fin = ws.finished.tolist()
    bad_rows = [b for b, f in enumerate(fin) if f == 2]
    finite = bool(torch.isfinite(ws.logits.float()).all())
    say(f"{where}: rows with the device error flag {bad_rows}; logits finite {finite}; tokens in range "
        f"{bool(((ws.cur_tok >= 0) & (ws.cur_tok < cfg.text_decoder.vocab_size)).all())}")
    if bad_rows or not finite:
        raise SystemExit(f"{where}: non-finite logits (rows {bad_rows})")


def finite_step(ws, cache):
    """One eager decode step, every launch checked."""
    watched = {"slab": ws.slab, "h": ws.h, "x16": ws.x16, "att16": ws.att16, "act16": ws.act16, "logits": ws.logits}
    ws.slab.fill_(float("nan"))                                  # an element read before it is written shows up downstream
    torch.cuda.synchronize()
    real = _lib.call
    seen = {"n": 0, "first": None}

    def checked(name, *args):
        real(name, *args)
        torch.cuda.synchronize()
        seen["n"] += 1
        if seen["first"] is not None or not name.startswith("q3asr_") or name.startswith("q3asr_graph"):
            return
        for k, t in watched.items():
            if k == "slab":
                continue                                         # holds the poison wherever the last GEMM did not write
            if not bool(torch.isfinite(t.float()).all()):
                seen["first"] = (seen["n"], name, k)
                say(f"launch #{seen['n']} {name}: first non-finite value in {k}")
                return
    _lib.call = checked
    import qwen3_asr_mlx_amd.ops as ops_mod
    ops_mod.call = checked
    try:
        dec.decode_step(ws, cache, sampling)
    finally:
        _lib.call = real
        ops_mod.call = real
    say(f"finite: {seen['n']} launches checked, first non-finite: {seen['first']}")
    return seen["first"]


if what & {"prefill", "step", "graph", "graph4", "regraph", "finite"}:
    ws, cache = prefilled(B)
    say("prefill done")
    if "finite" in what:
        if finite_step(ws, cache) is not None:
            raise SystemExit("finite: a launch left a non-finite value")
    if "step" in what:
        for _ in range(2):
            dec.decode_step(ws, cache, sampling)
        torch.cuda.synchronize()
        say("eager decode steps done")
        check_state(ws, "step")
    for name, steps in (("graph", 1), ("graph4", dec.MULTI_STEPS)):
        if name in what:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dec.decode_step(ws, cache, sampling)
                g = dec._capture(ws, cache, sampling, steps)
                for _ in range(4):
                    _lib.call("q3asr_graph_launch", g, torch.cuda.current_stream().cuda_stream)
                side.synchronize()
                _lib.call("q3asr_graph_destroy", g)
            say(f"{name}: captured and replayed")
            check_state(ws, name)
    if "regraph" in what:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dec.decode_step(ws, cache, sampling)
            for rep in range(3):
                g = dec._capture(ws, cache, sampling, 1 if rep != 1 else dec.MULTI_STEPS)
                for _ in range(3):
                    _lib.call("q3asr_graph_launch", g, torch.cuda.current_stream().cuda_stream)
                _lib.call("q3asr_graph_destroy", g)               # destroyed while its launches may still be in flight
            side.synchronize()
        say("regraph: capture / replay / destroy x3 done")
        check_state(ws, "regraph")
if "batch1" in what:
    ws1, cache1 = prefilled(1)
    for _ in range(4):
        dec.decode_step(ws1, cache1, sampling)
    torch.cuda.synchronize()
    say("batch-1 eager steps (fused residual+norm GEMM) done")
    check_state(ws1, "batch1")
out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
(out / "pmc_probe_meta.json").write_text(json.dumps({
    "lib_sha16": hashlib.sha256(_lib.LIB_PATH.read_bytes()).hexdigest()[:16], "ran": sorted(what), "batch": B, "clip_seconds": secs}))
say("done")
