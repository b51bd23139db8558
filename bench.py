#!/usr/bin/env python3
"""Headline benchmark: real-time factor (audio-seconds / wall-seconds) of the Qwen3-ASR-1.7B hot
path on 30 s clips, batch 32 per GPU (BASELINE.json configs[2]; configs[3] = the same per GPU on 8).

A step = one pass of the hot path over one batch already resident in HBM: log-mel -> conv stem +
24-layer encoder -> prompt merge -> 28-layer prefill -> N_dec-1 hipGraph-replayed decode steps with
on-device greedy sampling (repetition penalty 1.2, the API default) -> token ids on the host.
Weights are synthetic (seeded N(0, 0.02^2), bf16, reference checkpoint layout); random weights never
emit EOS, so every sequence decodes exactly N_dec tokens.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts the N ranks itself)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

N > 1: rank 0 owns the request (32*N clips resident on its GPU); the step is the product's own
`dist.run_sharded` (scatter of the audio rows, per-rank hot path on the device tensor, gather of token
ids over torch.distributed = RCCL over xGMI); weights are replicated; weak scaling.
`value` has the batch resident in HBM when the timed region starts and includes detokenisation (the bench contract of this
build: inputs resident, PCIe never inside `value`).  SURVEY.md 8d defines the reference's metric one step wider -- host
arrays in, H2D inside: that figure, through the public `transcribe_batch`, is reported beside it as `end_to_end` with
`"conforms_to": "SURVEY.md 8d"`; `value_definition` says which is which.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
_T0 = time.perf_counter()


def lib_sha16() -> str:
    import hashlib
    from qwen3_asr_mlx_amd import _lib
    return hashlib.sha256(_lib.LIB_PATH.read_bytes()).hexdigest()[:16]


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def log(msg: str) -> None:
    """Progress on stderr (stdout carries only the JSON line)."""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def synth_clip(i: int, n: int) -> np.ndarray:
    from signals import signal
    return signal("speechlike", n, seed=1234 + i)


def decode_bytes_per_launch(cfg, B: int, bytes_per_weight: int = 2):
    """Algorithmic bytes of one skinny-GEMM launch = weight bytes (2*N*K); list over one decode step."""
    t = cfg.text_decoder
    H, I = t.hidden_size, t.intermediate_size
    qd = t.num_attention_heads * t.head_dim
    qkvd = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
    w = bytes_per_weight
    per_layer = [w * qkvd * H, w * H * qd, w * 2 * I * H, w * H * I]
    return per_layer * t.num_hidden_layers + [w * t.vocab_size * H]


def cpu_baseline(cfg, state, clip, n_dec_sample: int, n_dec_full: int, threads: int = 0):
    """The oracle (CPU restatement of the reference algorithm, oracle/) timed on this host's cores on
    ONE clip with a short decode, extrapolated to n_dec_full tokens.  threads = 0: all cores of this box's share."""
    import torch
    import oracle
    # threads actually used: the affinity mask, capped at the 16-core share a one-GPU box gets
    cores = threads or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads")
    Wd = {k[len("model."):]: v.float().cpu() for k, v in state.items() if k.startswith("model.")}
    We = {k[len("audio_tower."):]: v.float().cpu() for k, v in state.items() if k.startswith("audio_tower.")}
    a, t = cfg.audio_encoder, cfg.text_decoder
    t0 = time.perf_counter()
    mel = oracle.log_mel_spectrogram(clip)
    t_mel = time.perf_counter() - t0
    log(f"cpu_baseline: mel {t_mel:.2f}s")
    t0 = time.perf_counter()
    feats = oracle.encoder_forward(torch.from_numpy(mel), We, a.encoder_attention_heads, a.n_window, a.n_window_infer)
    t_enc = time.perf_counter() - t0
    log(f"cpu_baseline: encoder {t_enc:.2f}s")
    dec = oracle.DecoderOracle(Wd, t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.rms_norm_eps, t.rope_theta)
    prompt = oracle.build_prompt(feats.shape[0], [6364])
    t0 = time.perf_counter()
    emb = oracle.prepare_inputs(feats, prompt, Wd["embed_tokens.weight"])
    logits = dec.forward(emb)
    t_pre = time.perf_counter() - t0
    log(f"cpu_baseline: prefill {t_pre:.2f}s")
    tok = oracle.sample(logits[-1])
    out = [tok]
    t0 = time.perf_counter()
    for _ in range(n_dec_sample):
        logits = dec.forward(dec.embed([tok]))
        tok = oracle.sample(logits[-1], repetition_penalty=1.2, recent_tokens=out[-100:])
        out.append(tok)
    t_tok = (time.perf_counter() - t0) / n_dec_sample
    total = t_mel + t_enc + t_pre + (n_dec_full - 1) * t_tok
    secs = len(clip) / 16000.0
    return {
        "value": secs / total, "unit": "audio-s/wall-s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "sample": (f"oracle/ (numpy mel + torch-CPU fp32 encoder/decoder, batch 1) on 1 clip of {secs:g} s: mel {t_mel:.2f} s, "
                   f"encoder {t_enc:.2f} s, prefill {t_pre:.2f} s, {n_dec_sample} decode steps at {t_tok:.3f} s/token "
                   f"extrapolated to {n_dec_full} tokens"),
    }


def decode_gemm_roofline(model, cfg, B: int, ndec: int, bytes_per_weight: int, dev):
    """Roofline of the dominant kernel (decode weight streaming: skinny4_kernel / head_stream_kernel; skinny3_kernel where no
    skinny4 geometry fits).  HIP events on the launch stream around R back-to-back "GEMM-only" decode steps: exactly the 113
    weight-streaming launches of a step (4 per layer over all 28 layers' distinct weights, so no cache reuse between launches,
    + the tied head), same operands and split factors as the timed region, captured into a hipGraph and replayed (as the
    decode step itself runs: issued one by one from Python they are host-bound at ~8 us per launch)."""
    import torch
    from qwen3_asr_mlx_amd import _lib
    dec = model._decoder
    ws = dec.workspace(B, ndec)
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    lib = _lib.load()
    lib.q3asr_event_create(C.byref(e0))
    lib.q3asr_event_create(C.byref(e1))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        stream = C.c_void_p(side.cuda_stream)
        dec.gemm_only_step(ws)                                       # warm (code objects, LDS attributes)
        _lib.call("q3asr_graph_begin", stream)
        try:
            per_step = dec.gemm_only_step(ws)
        finally:
            graph = C.c_void_p()
            _lib.call("q3asr_graph_end", stream, C.byref(graph))
        _lib.call("q3asr_graph_launch", graph, stream)               # warm replay
        R = 20
        _lib.check(lib.q3asr_event_record(e0, stream), "event_record")
        for _ in range(R):
            _lib.call("q3asr_graph_launch", graph, stream)
        _lib.check(lib.q3asr_event_record(e1, stream), "event_record")
        side.synchronize()
        _lib.check(lib.q3asr_event_elapsed_ms(e0, e1, C.byref(ms)), "event_elapsed")
        _lib.call("q3asr_graph_destroy", graph)
    lib.q3asr_event_destroy(e0)
    lib.q3asr_event_destroy(e1)
    launches = per_step * R
    per_launch = decode_bytes_per_launch(cfg, B, bytes_per_weight)
    total_bytes = sum(per_launch) * R
    avg_ms = ms.value / launches
    achieved = total_bytes / (ms.value * 1e-3) / 1e9
    log(f"roofline leg ({bytes_per_weight} B/weight, batch {B}): {launches} decode GEMM launches in {ms.value:.2f} ms")
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "kernel": "skinny4_kernel + head_stream_kernel (decode weight streaming, all shapes of one step)",
            "launches_measured": launches, "avg_launch_us": 1e3 * avg_ms, "bytes_per_launch": sum(per_launch) / len(per_launch),
            "measured_on": "a hipGraph of the step's 113 weight-streaming launches alone (HIP events on the launch stream)"}


def sampler_times(model, cfg, B: int, dev, reps: int = 40):
    """us per q3asr_sample_step launch on bf16 logits [B][V] ~ N(0, 3^2): greedy + penalty 1.2 (the headline's sampler) beside
    temperature 0.7 / top-p 0.9 + penalty 1.2 (BASELINE.json configs[4]); HIP events around `reps` back-to-back launches."""
    import torch
    from qwen3_asr_mlx_amd.decoder import DecodeWorkspace
    dec = model._decoder
    out = {}
    for name, sm in (("greedy_us", {"temperature": 0.0, "top_p": 1.0}), ("t0.7_top_p0.9_us", {"temperature": 0.7, "top_p": 0.9})):
        ws = DecodeWorkspace(cfg.text_decoder, B, reps + 8, dev, fp8=getattr(dec, "weight_dtype", "bf16") != "bf16")
        ws.logits.copy_((torch.randn(ws.logits.shape, device=dev) * 3.0).to(torch.bfloat16))
        from qwen3_asr_mlx_amd.tokenizer import EOS_TOKEN_IDS
        ws.logits[:, sorted(EOS_TOKEN_IDS)] = float("-inf")          # random logits: an EOS draw would end a row in the middle of the timing loop
        ws.max_new.fill_(reps + 8)
        sampling = {"temperature": sm["temperature"], "top_p": sm["top_p"], "top_k": 0, "repetition_penalty": 1.2, "repetition_context_size": 100, "seed": 7}
        dec.sample_step(ws, ws.logits, sampling)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dec.sample_step(ws, ws.logits, sampling)
        e1.record()
        e1.synchronize()
        out[name] = 1e3 * e0.elapsed_time(e1) / reps
        assert int(ws.finished.sum()) == 0
    return out


def longform_fp8_leg(cfg, state, dev, q, synth, ndec: int = 128, minutes: int = 20):
    """BASELINE.json configs[4]: Qwen3-ASR-1.7B with fp8 decoder weights (prefill fp8 x fp8 on the CDNA4 fp8 MFMA, W8A16 decode
    steps), 20 min of synthetic audio, low-energy chunking at chunk_duration = 30 (reference model.py:382-447,454-513), temperature
    0.7 / top-p 0.9 / repetition penalty 1.2.  Host float32 array in, text out: split points, H2D, mel, encoder, prefill, decode and
    detokenisation are all inside the timed region.  Two budgets: SURVEY.md 8d's N_dec = 128 per 30 s segment (comparable with the
    headline), through Qwen3ASR._run_jobs -- what _transcribe_chunked runs after it has fixed the budgets -- and the reference's own
    rule max(256, 50 / s) = 1,500 tokens per segment through the public transcribe() (random weights never emit EOS: every segment
    decodes its whole budget, the worst case of that rule)."""
    import torch
    from qwen3_asr_mlx_amd.model import _find_split_points
    from signals import signal
    log("longform_fp8: building the fp8 model")
    m8 = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype="fp8")
    audio = np.concatenate([signal("speechlike", 16000 * 60, seed=100 + i) for i in range(minutes)])
    secs = len(audio) / 16000.0

    def run_fixed(budget: int):
        pts = _find_split_points(audio, 30 * 16000, 5 * 16000)
        jobs, prev = [], 0
        for sp in pts + [len(audio)]:
            seg = audio[prev:sp]
            prev = sp
            if len(seg):
                jobs.append((0, seg, budget))
        texts = m8._run_jobs(jobs, "English", 0.7, 0.9, 0, 1.2, 100)
        return len(jobs), " ".join(t_ for t_ in texts if t_)

    def timed(fn, reps):
        lat, r = [], None
        for i in range(reps + 1):                              # the first run is the warm-up (workspace, graph capture)
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            if i:
                lat.append(time.perf_counter() - s0)
        return r, float(np.median(lat))

    (n_seg, text), t128 = timed(lambda: run_fixed(ndec), 3)
    assert n_seg == 40 and len(text) > 0
    log(f"longform_fp8: {n_seg} segments x {ndec} tokens: {1e3 * t128:.1f} ms, RTF {secs / t128:.0f}x")
    r_ref, t_ref = timed(lambda: m8.transcribe(audio, temperature=0.7, top_p=0.9, chunk_duration=30.0), 1)
    assert r_ref.duration == secs and len(r_ref.text) > 0
    log(f"longform_fp8: reference budget (1,500 tokens/segment): {t_ref:.2f} s, RTF {secs / t_ref:.0f}x")
    B = n_seg
    roof = decode_gemm_roofline(m8, cfg, B, ndec, 1, dev)
    smp = sampler_times(m8, cfg, B, dev)
    # the step's algorithmic bytes: 1.72 GB of fp8 weights + B * L * 114,688 B of K/V (bf16) at the mean context of the run
    l_bar = 408 + ndec / 2.0
    step_bytes = 1.7205e9 + B * l_bar * 114688
    out = {"value": secs / t128, "unit": "audio-s/wall-s", "p50_latency_ms": 1e3 * t128,
           "workload": f"Qwen3-ASR-1.7B fp8 decoder weights, {minutes} min of audio as ONE request, chunk_duration 30 -> {n_seg} segments decoded as one "
                       f"batch, temperature 0.7 / top-p 0.9 / repetition penalty 1.2, {ndec} decoded tokens per segment (SURVEY.md 8d), "
                       "host array in -> text out (split points, H2D, detokenisation inside)",
           "reference_budget": {"value": secs / t_ref, "unit": "audio-s/wall-s", "latency_ms": 1e3 * t_ref,
                                "workload": "the same request through the public transcribe(): the reference's budget max(256, 50/s) = 1,500 tokens per segment, "
                                            "all decoded (no EOS with random weights)", "ms_per_token": 1e3 * t_ref / 1499.0},
           "roofline": {**roof, "step_algorithmic_bytes": step_bytes,
                        "note": "fp8 weight streaming of one W8A16 decode step (1 byte per weight), batch = the request's 40 segments"},
           "sampler": {**smp, "note": "q3asr_sample_step at batch 40 over 151,936 bf16 logits per row; the draw runs the histogram top-p + categorical path, greedy the argmax path"},
           "dtype": "fp8-e4m3 decoder weights; prefill fp8 x fp8 MFMA (per-token e4m3 activations), decode bf16 activations; encoder bf16"}
    m8.close()
    del m8
    torch.cuda.empty_cache()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps (BASELINE.json protocol: >= 20 repeats after 3 warm-ups)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--clip-seconds", type=float, default=30.0)
    ap.add_argument("--ndec", type=int, default=128, help="decoded tokens per utterance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the batches-in-flight throughput leg")
    ap.add_argument("--inflight", type=str, default="2,4", help="batches in flight in the throughput legs (host threads / lanes), comma-separated")
    ap.add_argument("--weights", choices=("bf16", "fp8", "fp8w"), default="bf16",
                    help="fp8: decoder weights quantised on load, prefill on the fp8 MFMA (BASELINE.json configs[4]; not the headline "
                         "config); fp8w: weight-only variant")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the ndec=1500 and 5 s secondary legs")
    ap.add_argument("--no-longform", action="store_true", help="skip the configs[4] leg (fp8 weights, 20 min long-form, T = 0.7 / top-p 0.9)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start N fresh rank processes (one per GPU) BEFORE this process touches the
        # GPU -- children of a process that never initialised HIP, never a re-exec -- and pass their output through.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        log("self-launch: " + " ".join(cmd))
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import _lib, synth
    from qwen3_asr_mlx_amd.config import ModelConfig

    # The GPU boxes show 256 CPUs, torch takes 128 intra-op threads, the container's cgroup allows 16: a 128-thread CPU op (weight
    # initialisation, the CPU baseline's GEMMs) exhausts the quota and the kernel parks the WHOLE process, timed regions included, for
    # the rest of the scheduler period (cpu.stat nr_throttled).  The harness keeps torch inside the quota.
    try:
        quota_, period_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota_ != "max":
            torch.set_num_threads(max(1, min(torch.get_num_threads(), int(quota_) // int(period_))))
    except (OSError, ValueError):
        pass
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # Q3ASR_BENCH_REHEARSE=1: every rank uses cuda:0 and the collectives go through gloo on host copies --
    # only to rehearse the N > 1 control flow on a one-GPU box; the real run is one rank per GPU over RCCL.
    rehearse = os.environ.get("Q3ASR_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    cfg = ModelConfig()                                     # Qwen3-ASR-1.7B defaults (reference config.py)
    log("building synthetic 1.7B weights on the device")
    state = synth.random_state(cfg, seed=0, device=dev)
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype=args.weights)
    log("model ready")
    model.use_graph = not args.no_graph
    if not (rank == 0 and world == 1 and (not args.no_cpu_baseline or not args.no_longform)):
        del state
    B, n = args.batch, int(args.clip_seconds * 16000)
    budgets = [args.ndec] * B

    # ---- request: resident in HBM before the timed region
    detok = model._decode_output
    if world == 1:
        host_clips = [synth_clip(i, n) for i in range(B)]
        audio, lens_dev, lens = q.audio.pack_audio(host_clips, dev)
        torch.cuda.synchronize()

        def step():
            toks_ = model.generate_tokens_device(audio, lens_dev, lens, budgets, "English", repetition_penalty=1.2)
            return toks_, [detok(t_) for t_ in toks_]
    else:
        from qwen3_asr_mlx_amd import dist as qd
        cdev = "cpu" if rehearse else dev                      # rehearsal: gloo moves host tensors
        request = all_budgets = None
        if rank == 0:
            packed = q.audio.pack_audio([synth_clip(i, n) for i in range(world * B)], dev)[0]
            request, all_budgets = (packed.to(cdev), [n] * (world * B)), [args.ndec] * (world * B)
        torch.cuda.synchronize()

        def run_rank(a_, lens_, budgets_):                     # this rank's share: padded tensor [B][n] on `cdev`
            return model.generate_tokens_packed(a_.to(dev), lens_, budgets_, "English", repetition_penalty=1.2)

        def step():
            toks_ = qd.run_sharded(run_rank, request, all_budgets, device=cdev)      # the product's DP path
            return toks_, ([detok(t_) for t_ in toks_] if rank == 0 else None)

    for i in range(args.warmup):
        w0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i}: {time.perf_counter() - w0:.3f}s")

    # N > 1: what took part, as the process group itself reports it (one entry per rank, gathered through the group), and a
    # single-GPU reference measured in this very run: every rank times the per-GPU batch alone on its own device (no
    # collective in the path), so that the aggregate can be read as a fraction of N x single without a second run.
    ranks_info = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        a1_, l1d_, l1_ = q.audio.pack_audio([synth_clip(i, n) for i in range(B)], dev)
        solo = []
        for i in range(3):
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            model.generate_tokens_device(a1_, l1d_, l1_, budgets, "English", repetition_penalty=1.2)
            torch.cuda.synchronize()
            if i:
                solo.append(time.perf_counter() - s0)
        del a1_, l1d_
        mine = {"rank": rank, "local_rank": local, "device": str(dev), "device_name": props.name,
                "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
                "pid": os.getpid(), "solo_ms_per_step": 1e3 * float(np.median(solo))}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = {"backend": dist.get_backend(), "world_size_from_group": dist.get_world_size(), "rehearsal_on_one_gpu": rehearse,
                      "distinct_devices": len({(g_["device_uuid"], g_["pci_bus_id"], g_["device"]) for g_ in gathered}), "ranks": gathered}

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(x: float) -> float:
        if world == 1:
            return x
        t_ = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    fence()
    per_step = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        toks, texts = step()
        torch.cuda.synchronize()
        per_step.append(time.perf_counter() - s0)
        log(f"timed step: {per_step[-1]:.3f}s")
    fence()
    my_elapsed = time.perf_counter() - t0
    elapsed = all_max(my_elapsed)
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, 1e3 * my_elapsed / args.steps)
        for g_, ms_ in zip(ranks_info["ranks"], per_rank):
            g_["ms_per_step"] = ms_
    if rank == 0:
        assert len(toks) == world * B and all(len(t) == args.ndec for t in toks) and len(texts) == len(toks)

    if rank != 0:
        dist.destroy_process_group()
        return

    audio_s = world * B * args.clip_seconds * args.steps
    line = {
        "metric": "real-time factor (audio-sec/wall-sec), Qwen3-ASR-1.7B bf16 30s clips",
        "value": audio_s / elapsed, "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp8": "fp8-e4m3 decoder weights; prefill fp8 x fp8 MFMA, decode bf16 activations",
                  "fp8w": "bf16 activations, fp8-e4m3 decoder weights"}[args.weights], "data": "synthetic",
        "config": {"workload": f"Qwen3-ASR-1.7B bf16, batch={B}/GPU, {args.clip_seconds:g} s clips, greedy (repetition_penalty 1.2), "
                               f"{args.ndec} decoded tokens/clip, hipGraph decode, detokenised" +
                               ("" if world == 1 else f", {world}-GPU DP (dist.run_sharded: scatter/gather over RCCL)"),
                   "batch_per_gpu": B, "clip_seconds": args.clip_seconds, "decoded_tokens": args.ndec,
                   "weights": "synthetic seeded N(0,0.02^2) bf16", "graph": not args.no_graph},
        "p50_latency_ms": 1e3 * float(np.median(per_step)),
        "value_definition": "request resident in HBM when the timed region starts, token ids detokenised inside it (bench contract); "
                            "the SURVEY.md 8d figure with the H2D copy inside is `end_to_end`",
    }
    if world > 1:
        solo_ms = float(np.median([g_["solo_ms_per_step"] for g_ in ranks_info["ranks"]]))
        ranks_info["single_gpu_ms_per_step"] = solo_ms
        ranks_info["single_gpu_value"] = B * args.clip_seconds / (solo_ms * 1e-3)
        line["ranks"] = ranks_info
        line["frac_of_N_x_single"] = line["value"] / (world * ranks_info["single_gpu_value"])

    if world == 1:
        # ---- the same request with repetition_penalty 1.0 (SURVEY.md 8d asks for both; the headline uses the API default 1.2)
        np_lat = []
        for i in range(3):
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            model.generate_tokens_device(audio, lens_dev, lens, budgets, "English", repetition_penalty=1.0)
            torch.cuda.synchronize()
            if i:
                np_lat.append(time.perf_counter() - s0)
        line["no_repetition_penalty"] = {"value": B * args.clip_seconds / float(np.median(np_lat)), "unit": "audio-s/wall-s",
                                         "p50_latency_ms": 1e3 * float(np.median(np_lat))}
    if world == 1:
        # ---- SURVEY.md 8d's end-to-end definition through the public API: host float32 arrays in (pinned staging + H2D over
        # PCIe inside the timed region), mel -> encoder -> prefill -> decode, texts out.  Never `value` (that is HBM-resident).
        # Both staging forms of audio.pack_audio are run interleaved in this one process (A/B), the shipped default last;
        # `h2d` says where `end_to_end - value` goes: host fill of the pinned buffer, the copies (HIP events around them), the rest.
        from qwen3_asr_mlx_amd import audio as qaudio
        default_mode = qaudio.PACK_MODE
        e2e = {"single": [], "grouped": []}
        h2d = {}
        res_ = None
        # One block per form, the shipped default last; two untimed runs open a block.  (Round 5 alternated the forms run by run: on
        # some boxes the first runs after a change of form carry a 5-35 ms host-fill spike -- tools/e2e_probe.py prints every run --
        # and with strict alternation every run is such a first run.)  `runs_ms` keeps every timed run, so a spike shows as one.
        for mode in sorted(e2e, key=lambda m_: m_ == default_mode):
            qaudio.PACK_MODE = mode
            for i in range(7):
                st_ = qaudio.PACK_STATS = {}
                torch.cuda.synchronize()
                s0 = time.perf_counter()
                res_ = model.transcribe_batch(host_clips, max_tokens=args.ndec)
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - s0
                qaudio.PACK_STATS = None
                if i >= 2:
                    e2e[mode].append(dt_)
                    h2d.setdefault(mode, []).append(qaudio.pack_stats_finish(st_))
        qaudio.PACK_MODE = default_mode
        assert len(res_) == B and all(isinstance(r_.text, str) for r_ in res_)
        med = {m_: float(np.median(v_)) for m_, v_ in e2e.items()}

        def h2d_summary(mode):
            recs = h2d[mode]
            f_ = lambda k: float(np.median([r_[k] for r_ in recs]))          # noqa: E731
            return {"mode": mode, "bytes": recs[0]["bytes"], "groups": recs[0]["groups"], "pinned": recs[0]["pinned"],
                    "setup_ms": f_("setup_ms"), "host_fill_ms": f_("host_fill_ms"), "copy_issue_ms": f_("copy_issue_ms"),
                    "copy_span_ms": f_("copy_span_ms"), "copy_GBps_over_span": recs[0]["bytes"] / (f_("copy_span_ms") * 1e-3) / 1e9,
                    "host_total_ms": f_("host_total_ms"), "p50_latency_ms": 1e3 * med[mode],
                    "runs_ms": [round(1e3 * v_, 1) for v_ in e2e[mode]]}
        line["end_to_end"] = {"value": B * args.clip_seconds / med[default_mode], "unit": "audio-s/wall-s",
                              "p50_latency_ms": 1e3 * med[default_mode], "conforms_to": "SURVEY.md 8d (H2D of the audio inside the timed region)",
                              "note": "Qwen3ASR.transcribe_batch(list of host float32 arrays): H2D of the batch and detokenisation inside the timed region",
                              "minus_resident_ms": 1e3 * med[default_mode] - line["p50_latency_ms"],
                              "h2d": h2d_summary(default_mode),
                              "h2d_other_form": h2d_summary("grouped" if default_mode == "single" else "single"),
                              "h2d_note": "audio.pack_audio: host_fill = copying the request's rows into the pinned staging buffer, copy_span = first copy "
                                          "issued -> last copy done on the device timeline (the grouped form's span contains the fills between its copies); "
                                          "both forms in this process, one block of 2 untimed + 5 timed runs each, medians"}
        log(f"end-to-end (H2D + detokenise): {line['end_to_end']['p50_latency_ms']:.1f} ms vs {line['p50_latency_ms']:.1f} ms resident "
            f"[{default_mode}: fill {line['end_to_end']['h2d']['host_fill_ms']:.1f} ms, copy span {line['end_to_end']['h2d']['copy_span_ms']:.1f} ms; "
            f"other form {line['end_to_end']['h2d_other_form']['p50_latency_ms']:.1f} ms]")
    if world == 1 and not args.no_extra_configs and args.weights == "bf16":
        # ---- secondary legs SURVEY.md 8d names: the reference's own token budget max(256, 50/s) = 1500 per 30 s clip,
        # and the 5 s configuration (32 tokens per clip)
        def leg(clips_n, ndec_, reps):
            a_, ld_, l_ = q.audio.pack_audio([synth_clip(i, clips_n) for i in range(B)], dev)
            lat_ = []
            for i in range(reps + 1):
                torch.cuda.synchronize()
                s0 = time.perf_counter()
                t_ = model.generate_tokens_device(a_, ld_, l_, [ndec_] * B, "English", repetition_penalty=1.2)
                _ = [detok(x_) for x_ in t_]
                torch.cuda.synchronize()
                if i:
                    lat_.append(time.perf_counter() - s0)
            assert all(len(x_) == ndec_ for x_ in t_)
            m_ = float(np.median(lat_))
            return {"value": B * clips_n / 16000.0 / m_, "unit": "audio-s/wall-s", "p50_latency_ms": 1e3 * m_,
                    "workload": f"batch={B}, {clips_n / 16000:g} s clips, {ndec_} decoded tokens/clip"}
        line["ndec1500"] = leg(n, 1500, 2)
        log(f"ndec=1500: RTF {line['ndec1500']['value']:.0f}x, {line['ndec1500']['p50_latency_ms']:.0f} ms per batch")
        line["clip5s"] = leg(80000, 32, 5)
        log(f"5 s clips, 32 tokens: RTF {line['clip5s']['value']:.0f}x, {line['clip5s']['p50_latency_ms']:.1f} ms per batch")
    if world == 1 and B > 1:
        # ---- BASELINE.json configs[1]: batch 1, one 30 s clip, same decode budget -> latency
        a1, l1d, l1 = q.audio.pack_audio([synth_clip(0, n)], dev)
        lat = []
        for i in range(7):
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            model.generate_tokens_device(a1, l1d, l1, [args.ndec], "English", repetition_penalty=1.2)
            torch.cuda.synchronize()
            if i >= 2:
                lat.append(time.perf_counter() - s0)
        line["batch1"] = {"p50_latency_ms": 1e3 * float(np.median(lat)), "rtf": args.clip_seconds / float(np.median(lat)),
                          "workload": f"batch=1, {args.clip_seconds:g} s clip, {args.ndec} decoded tokens"}
        log(f"batch-1 latency p50 {line['batch1']['p50_latency_ms']:.1f} ms")
    if world == 1 and B > 1 and not args.no_pipeline:
        # ---- two batches in flight: two host threads, each bound to its own lane (decode workspace, KV cache, graph,
        # streams) over shared weights.  Not the headline `value` (that is one batch at a time): throughput of a server
        # that keeps the GPU fed, with the per-batch latency it costs.
        import threading
        from qwen3_asr_mlx_amd.decoder import set_lane
        for n_lanes in [int(v) for v in str(args.inflight).split(",") if v.strip()]:
            n_each, lat2, errs = 8, [], []                 # per lane: enough batches that the ramp at both ends stays small
            gate = threading.Barrier(n_lanes + 1)

            def worker(lane):
                try:
                    set_lane(lane)
                    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                        model.generate_tokens_device(audio, lens_dev, lens, budgets, "English", repetition_penalty=1.2)   # lane warm-up
                        torch.cuda.current_stream().synchronize()
                        gate.wait()
                        for _ in range(n_each):
                            s0 = time.perf_counter()
                            t_ = model.generate_tokens_device(audio, lens_dev, lens, budgets, "English", repetition_penalty=1.2)
                            torch.cuda.current_stream().synchronize()
                            lat2.append(time.perf_counter() - s0)
                            assert all(len(x) == args.ndec for x in t_)
                        gate.wait()
                except Exception as e:                                       # noqa: BLE001
                    errs.append(repr(e))
                    gate.abort()

            ths = [threading.Thread(target=worker, args=(lane,)) for lane in range(1, n_lanes + 1)]
            for th in ths:
                th.start()
            try:
                gate.wait()
                p0 = time.perf_counter()
                gate.wait()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - p0
            except threading.BrokenBarrierError:
                dt2 = None
            for th in ths:
                th.join()
            if dt2 is not None and not errs:
                key = "two_in_flight" if n_lanes == 2 else f"in_flight_{n_lanes}"
                line[key] = {"value": n_lanes * n_each * B * args.clip_seconds / dt2, "unit": "audio-s/wall-s",
                             "batches": n_lanes * n_each, "p50_latency_ms": 1e3 * float(np.median(lat2)),
                             "note": f"same batch-{B} requests, {n_lanes} in flight on their own lanes/streams; `value` above is one at a time"}
                log(f"{n_lanes} batches in flight: RTF {line[key]['value']:.0f}x, per-batch latency p50 {line[key]['p50_latency_ms']:.0f} ms")
            else:
                log(f"{n_lanes}-in-flight leg failed: {errs}")
    if world == 1:
        # ---- per-stage time of one step against each stage's own roofline (SURVEY.md section 8d work figures)
        from qwen3_asr_mlx_amd.generate import _merge, _side_stream
        from qwen3_asr_mlx_amd.tokenizer import build_prompt
        dec_, enc_ = model._decoder, model._encoder
        fe = q.audio.get_frontend(dev)

        def timed(fn, reps=1):
            # reps > 1: one untimed warm-up call (the first run after the batches-in-flight legs can pay for a fresh multi-GB block
            # of the caching allocator: one line showed encoder_ms = 120 that way), then the MEDIAN of `reps` timed runs
            r = None
            if reps > 1:
                r = fn()
            lat = []
            for _ in range(reps):
                torch.cuda.synchronize()
                s0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - s0)
            return r, float(np.median(lat))

        (mel, n_frames), t_mel = timed(lambda: fe(audio, lens_dev, lens), 3)
        (feats, tok_counts), t_enc = timed(lambda: enc_.forward_batch(mel, n_frames), 3)
        prompts = [build_prompt(nt, [6364]) for nt in tok_counts]
        seq_lens = [len(p_) for p_ in prompts]
        Lmax = (max(seq_lens) + args.ndec + 63) // 64 * 64
        cache = dec_.batch_cache(B, Lmax)
        ws = dec_.workspace(B, args.ndec)
        sampling = {"temperature": 0.0, "top_p": 1.0, "top_k": 0, "repetition_penalty": 1.2, "repetition_context_size": 100, "seed": 0}

        def prefill():
            ws.n_gen.zero_(); ws.finished.zero_()
            ws.pos.copy_(torch.tensor([t_ - 1 for t_ in seq_lens], dtype=torch.int32))
            ws.max_new.copy_(torch.tensor(budgets, dtype=torch.int32))
            emb = _merge(feats, prompts, tok_counts, dec_)
            dec_.prefill(emb, seq_lens, cache, logits_out=ws.logits)
            dec_.sample_step(ws, ws.logits, sampling)
        _, t_pre = timed(prefill, 3)

        def decode():                                     # the product's loop (generate.generate_batch): one eager step, then replays of
            import importlib                              # the captured 4-step graph with a look at the finished flags every 16 steps
            gen_ = importlib.import_module("qwen3_asr_mlx_amd.generate")
            side = _side_stream(dev)
            side.wait_stream(torch.cuda.current_stream())
            steps, done = args.ndec - 1, 1
            with torch.cuda.stream(side):
                dec_.decode_step(ws, cache, sampling)
                dec_.capture_step(ws, cache, sampling)
                checked = done
                while done < steps:
                    done += dec_.replay_step(ws, multi=steps - done >= dec_.MULTI_STEPS)
                    if done - checked >= gen_.FINISH_CHECK_EVERY:
                        checked = done
                        gen_._all_finished(ws)
                side.synchronize()                        # before the cross-stream wait: a parked stream slows the running one
            torch.cuda.current_stream().wait_stream(side)
        _, t_dec = timed(decode)
        secs = args.clip_seconds / 30.0
        T_avg = sum(seq_lens) / B
        l_bar = T_avg + args.ndec / 2.0
        w_b = 2 if args.weights == "bf16" else 1
        dec_bytes = (args.ndec - 1) * (w_b * 1.7205e9 + B * l_bar * 114688)
        line["stages"] = {
            "mel_ms": 1e3 * t_mel,
            "encoder_ms": 1e3 * t_enc, "encoder_tflops": B * 374.1e9 * secs / t_enc / 1e12,
            "encoder_frac_of_2.5PF": B * 374.1e9 * secs / t_enc / 2.5e15,
            "prefill_ms": 1e3 * t_pre, "prefill_tflops": B * 1170e9 * (T_avg / 408.0) / t_pre / 1e12,
            "prefill_frac_of_2.5PF": B * 1170e9 * (T_avg / 408.0) / t_pre / 2.5e15,
            "decode_ms": 1e3 * t_dec, "decode_ms_per_token": 1e3 * t_dec / (args.ndec - 1),
            "decode_GBps": dec_bytes / t_dec / 1e9, "decode_frac_of_8TBps": dec_bytes / t_dec / 8e12,
            "statistic": "mel / encoder / prefill: median of 3 after one untimed warm-up call; decode: one run of the product loop",
            "note": "one step re-run stage by stage with a device sync between stages (decode: the product loop, 4-step graph replays); work figures: SURVEY.md 8d "
                    "(encoder 374.1 GFLOP and prefill 1,170 GFLOP per 30 s clip, decode 3.441 GB weights + B*L*114,688 B KV per token)",
        }
        log("stages: " + ", ".join(f"{k}={v:.1f}" for k, v in line["stages"].items() if k.endswith("_ms")))
        # ---- roofline of the dominant kernel (decode weight streaming), measured live: decode_gemm_roofline above
        roof = decode_gemm_roofline(model, cfg, B, args.ndec, 2 if args.weights == "bf16" else 1, dev)
        per_launch = decode_bytes_per_launch(cfg, B, 2 if args.weights == "bf16" else 1)
        # PMC traffic: only a figure collected on THIS build of the library (tools/pmc_probe.py stamps the summary with the
        # sha-256 of libq3asr_hip.so); a summary from any other build is refused and `traffic` is null.
        traffic, traffic_from = None, "no PMC summary for this build of libq3asr_hip.so"
        lib_sha = lib_sha16()
        for pmc in sorted((ROOT / "profiles").glob("r*_pmc_summary.json"), reverse=True):
            d_ = json.loads(pmc.read_text())
            per = d_.get("decode_gemm_hbm_bytes_per_launch") or d_.get("skinny3_hbm_bytes_per_launch")
            if d_.get("lib_sha16") == lib_sha and per:
                traffic, traffic_from = per, f"profiles/{pmc.name} (lib sha16 {lib_sha})"
                break
        # the same launches inside the real decode step (attention, residual+norm and sampling launches between them), from
        # the rocprofv3 kernel trace of THIS build (tools/in_situ.py stamps it with the library hash): slightly slower than the
        # GEMM-only graph, and the number to quote for the step as it runs
        in_situ, in_situ_from = None, "no kernel-trace summary for this build of libq3asr_hip.so"
        for f_ in sorted((ROOT / "profiles").glob("r*_decode_in_situ.json"), reverse=True):
            d_ = json.loads(f_.read_text())
            if d_.get("lib_sha16") == lib_sha and d_.get("avg_launch_us"):
                in_situ, in_situ_from = d_, f"profiles/{f_.name} (lib sha16 {lib_sha})"
                break
        line["roofline"] = {**roof, "traffic": traffic, "traffic_from": traffic_from,
                            "in_situ": None if in_situ is None else {
                                "avg_launch_us": in_situ["avg_launch_us"], "achieved": sum(per_launch) / len(per_launch) / (in_situ["avg_launch_us"] * 1e-6) / 1e9,
                                "frac": sum(per_launch) / len(per_launch) / (in_situ["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "launches": in_situ.get("launches"), "whole_step_us": in_situ.get("whole_step_us")},
                            "in_situ_from": in_situ_from}
        if args.weights == "bf16" and not args.no_longform:
            line["longform_fp8"] = longform_fp8_leg(cfg, state, dev, q, synth)
        if not args.no_cpu_baseline and args.weights == "bf16":
            model.close()
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline(cfg, state, synth_clip(0, n), 8, args.ndec)
            # BASELINE.json configs[0] (5 s clip, 32 tokens) on all cores and on one (k = all and k = 1)
            line["cpu_baseline"]["clip5s_all_cores"] = cpu_baseline(cfg, state, synth_clip(0, 80000), 4, 32)
            line["cpu_baseline"]["clip5s_k1"] = cpu_baseline(cfg, state, synth_clip(0, 80000), 2, 32, threads=1)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
