"""Parity tests with stated tolerances at the Qwen3-ASR-1.7B shapes:

* last-position prefill logits of a 5 s and a 30 s clip against the oracle, atol stated;
* margin-aware greedy tokens under teacher forcing over >= 512 tokens: every step whose oracle top-1 margin exceeds twice
  the logit tolerance must match EXACTLY, the rest must lie inside the tolerance; exact-match rates are recorded;
* the same check, equally strict, against the reference-semantics oracle (gpu_rounding=False: fp32 encoder activations,
  unrounded softmax probabilities);
* BASELINE.json configs[2] as stated: 32 x 30 s clips in one batch, hipGraph decode, rows equal to single runs bit for
  bit and margin-aware against the oracle;
* weight_dtype="fp8" (configs[4]) against an oracle flavour that quantises the same operands to e4m3 per token;
* BASELINE.json configs[4] at full size: 20 min of audio, chunk_duration 30, fp8 weights, temperature 0.7 / top-p 0.9.

Rates are written to gpurun_out/parity_rates.json (copied to profiles/ by hand after a run)."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from signals import signal

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

# Logit tolerance in bf16 ulps OF THE TOP LOGIT (logits are bf16, decoder.py:252: one ulp is the resolution of the
# reference's own output).  2 ulps for the 2-layer config; 6 ulps after 28 layers of bf16 re-rounding at the 1.7B shapes.
# Rounds 1-4 used 4 ulps there (measured maximum 3-4).  Round 5 moved the SwiGLU of BOTH the oracle and the GPU to the
# reference's three rounding points (sigmoid -> bf16, gate * sigmoid -> bf16, * up -> bf16; decoder.py:96-97); a chain of three
# roundings passes a one-ulp difference of its operand on as up to three (stage tests: SwiGLU max 2 -> 3 ulps, one layer end to
# end 93.9 % -> 92.8 % within one ulp), and after 28 layers the largest gap of a non-decisive step measured 5 ulps
# (gpurun_out/r05a_parity.log).  The arithmetic of every stage is pinned amplification-free by the 28-layer stage test below.
# The 2-layer config ran at 2 ulps in rounds 1-4; round 5 saw the SAME build give a largest non-decisive gap of 2 ulps in one pytest
# process and 3 in another (gpurun_out/r05n_*): the GPU side is bitwise reproducible, the CPU oracle is not -- torch's CPU GEMMs sum
# in an order that depends on the thread count the process happens to run with.  The oracles now run on a fixed thread count
# (ORACLE_THREADS) and the small bar is 3 ulps.
ULPS_SMALL, ULPS_FULL = 3.0, 6.0
ORACLE_THREADS = 8
# weight_dtype="fp8": per-token e4m3 quantisation is discontinuous -- a one-bf16-ulp difference in an operand becomes a whole
# e4m3 step (2^-4 relative) when it crosses a rounding boundary, and a one-ulp difference in a row's largest element moves the
# row's whole grid.  The e4m3 oracle flavour itself moves 3-5 x as far as the bf16 flavour under the same sub-ulp perturbation
# of its projections (tests/test_oracle_cpu.py::test_fp8_flavour_amplifies_perturbations pins that factor), so two correct
# implementations of the W8A8 prefill agree to a multiple of the bf16 bar, not to the bf16 bar; the multiple grows with depth
# (measured on the CPU oracle alone: 3.2-4.8 x at 2 layers, 6-7 x at the 1.7B shapes' 28 layers, where two faithful runs keep
# only ~40 % of their quantisation noise in common).  What pins the arithmetic itself is bit-exact and depth-free: the
# quantiser and the fp8 GEMM in tests/test_kernels_gpu.py, and one whole layer against the oracle below.
FP8_AMPLIFICATION = {"small": 4.0, "full": 8.0}
# absolute bound on the last-position logits at the 1.7B shapes (|logit| <= ~5 with N(0, 0.02^2) weights)
FULL_LOGIT_ATOL = 0.125


def _ulp_bf16(x: float) -> float:
    return 2.0 ** (math.floor(math.log2(max(abs(x), 1e-30))) - 7)


def _record(key: str, value) -> None:
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    f = out / "parity_rates.json"
    data = json.loads(f.read_text()) if f.exists() else {}
    data[key] = value
    f.write_text(json.dumps(data, indent=1, sort_keys=True))
    print(f"[parity] {key}: {value}")


def _oracles(cfg, state):
    """(encoder fn, decoder) for both oracle flavours on CPU copies of the weights."""
    import oracle
    torch.set_num_threads(ORACLE_THREADS)                   # a fixed summation order for the CPU side (see ULPS_SMALL)
    Wd = {k[len("model."):]: v.float().cpu() for k, v in state.items() if k.startswith("model.")}
    We = {k[len("audio_tower."):]: v.float().cpu() for k, v in state.items() if k.startswith("audio_tower.")}
    t, a = cfg.text_decoder, cfg.audio_encoder
    out = {}
    for name, g in (("gpu_rounding", True), ("reference", False)):
        dec = oracle.DecoderOracle(Wd, t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.rms_norm_eps, t.rope_theta,
                                   gpu_rounding=g)
        enc = (lambda mel, g=g: oracle.encoder_forward(torch.from_numpy(mel), We, a.encoder_attention_heads, a.n_window,
                                                       a.n_window_infer, gpu_rounding=g))
        out[name] = (enc, dec)
    return out


def _teacher_forced_logits(dec, feats, toks, fp8_prompt=False):
    """Oracle logits of every generated position with the GPU's tokens forced: ONE full-context pass (cached == full
    context is a tested property of the oracle, tests/test_oracle_cpu.py).  fp8_prompt: the prompt rows in the e4m3 x e4m3
    arithmetic of a W8A8 prefill, the generated rows with bf16 activations (oracle/decoder.py, `fp8_rows`)."""
    import oracle
    prompt = oracle.build_prompt(feats.shape[0], [6364])
    dec.reset()
    emb = oracle.prepare_inputs(feats, prompt, dec.W["embed_tokens.weight"])
    T = emb.shape[0]
    full = torch.cat([emb, dec.embed(list(toks[:-1]))]) if len(toks) > 1 else emb
    return dec.forward(full, keep_from=T - 1, fp8_rows=T if fp8_prompt else 0)


def _margin_check(logits, toks, penalty, ulps, ctx=100, strict=True):
    """Per step: tol = `ulps` bf16 ulps of the oracle's top (penalised) logit.  margin > 2 tol -> the GPU token must BE the
    oracle's argmax; otherwise it must lie within tol of the best.  Returns counts."""
    import oracle
    st = {"n": len(toks), "exact": 0, "decisive": 0, "decisive_exact": 0, "max_gap_in_tol_units": 0.0}
    for i, tok in enumerate(toks):
        lg = logits[i]
        recent = list(toks[max(0, i - ctx): i])
        if penalty != 1.0 and recent:
            lg = oracle.apply_repetition_penalty(lg, recent, penalty)
        top2 = torch.topk(lg, 2)
        best, margin = int(top2.indices[0]), float(top2.values[0] - top2.values[1])
        tol = ulps * _ulp_bf16(float(top2.values[0]))
        gap = float(top2.values[0] - lg[tok])
        st["exact"] += int(tok == best)
        st["max_gap_in_tol_units"] = max(st["max_gap_in_tol_units"], gap / tol)
        if margin > 2 * tol:
            st["decisive"] += 1
            st["decisive_exact"] += int(tok == best)
            if strict:
                assert tok == best, f"step {i}: oracle margin {margin:.3f} > 2 x tol {tol:.3f} but GPU chose {tok}, oracle {best}"
        elif strict:
            assert gap <= tol, f"step {i}: GPU token {tok} is {gap:.3f} below the oracle's best (tol {tol:.3f})"
    return st


def _run_margin_case(model, cfg, state, clips, n_tok, penalty, ulps, tag):
    import oracle
    oracles = _oracles(cfg, state)
    toks_all = model.generate_tokens(clips, [n_tok] * len(clips), "English", repetition_penalty=penalty)
    tot = {k: {"n": 0, "exact": 0, "decisive": 0, "decisive_exact": 0, "max_gap_in_tol_units": 0.0} for k in oracles}
    for x, toks in zip(clips, toks_all):
        assert len(toks) == n_tok
        mel = oracle.log_mel_spectrogram(x)
        for name, (enc, dec) in oracles.items():
            st = _margin_check(_teacher_forced_logits(dec, enc(mel), toks), toks, penalty, ulps, strict=False)
            for k, v in st.items():
                tot[name][k] = max(tot[name][k], v) if k.startswith("max") else tot[name][k] + v
    for name, t in tot.items():
        t["exact_rate"] = round(t["exact"] / t["n"], 4)
        t["decisive_rate"] = round(t["decisive"] / t["n"], 4)
        t["decisive_exact_rate"] = round(t["decisive_exact"] / max(1, t["decisive"]), 4)
        t["max_gap_in_tol_units"] = round(t["max_gap_in_tol_units"], 3)
        t["tol_ulps"] = ulps
    _record(tag, tot)                       # both flavours on record first, then the strict bar on both
    for name, t in tot.items():
        assert t["decisive_exact"] == t["decisive"], (tag, name, "a decisive step (oracle margin > 2 tol) chose another token", t)
        assert t["max_gap_in_tol_units"] <= 1.0, (tag, name, "a non-decisive step lies outside the tolerance", t)
    return tot


# ----------------------------------------------------------------------------- small config (2 layers), 1,024 tokens
@pytest.mark.parametrize("embed_std", [0.02, 0.08])
def test_small_margin_aware_tokens_1024(dev, embed_std):
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import synth
    cfg = synth.small_config()
    state = synth.random_state(cfg, seed=5, bias_std=0.02, norm_jitter=0.1, embed_std=embed_std)
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
    clips = [signal("speechlike", n, seed=300 + i) for i, n in enumerate((80000, 48077, 160000, 240000, 16000, 112000, 333333, 480000))]
    tot = _run_margin_case(model, cfg, state, clips, 128, 1.2, ULPS_SMALL, f"small_embed_std_{embed_std}")
    g = tot["gpu_rounding"]
    assert g["n"] == 1024 and g["decisive_exact"] == g["decisive"] and g["decisive"] >= 300
    assert g["exact_rate"] >= 0.9
    r = tot["reference"]                                              # vs fp32-activation semantics: the same bar
    assert r["decisive_exact"] == r["decisive"] and r["decisive"] >= 300 and r["exact_rate"] >= 0.9


# ----------------------------------------------------------------------------- 1.7B shapes
@pytest.fixture(scope="module")
def full_wide(dev):
    """Qwen3-ASR-1.7B shapes with a 4x wider tied embedding (logit std ~3.6 instead of ~0.9)."""
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import synth
    from qwen3_asr_mlx_amd.config import ModelConfig
    cfg = ModelConfig()
    state = synth.random_state(cfg, seed=0, device=dev, embed_std=0.08)
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
    return cfg, state, model


@pytest.fixture(scope="module")
def full_std(dev):
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import synth
    from qwen3_asr_mlx_amd.config import ModelConfig
    cfg = ModelConfig()
    state = synth.random_state(cfg, seed=0, device=dev)
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
    return cfg, state, model


def _gpu_prefill_logits(model, x):
    """mel -> encoder -> prompt merge -> prefill: bf16 logits of the last prompt position, through the product kernels."""
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd.generate import _merge
    from qwen3_asr_mlx_amd.tokenizer import build_prompt
    dev = model._decoder.device
    audio, lens_dev, lens = q.audio.pack_audio([x], dev)
    mel, n_frames = q.audio.get_frontend(dev)(audio, lens_dev, lens)
    feats, counts = model._encoder.forward_batch(mel, n_frames)
    prompt = build_prompt(counts[0], [6364])
    cache = model._decoder.batch_cache(1, (len(prompt) + 63) // 64 * 64)
    emb = _merge(feats, [prompt], counts, model._decoder)
    return model._decoder.prefill(emb, [len(prompt)], cache)[0].float().cpu(), feats.cpu()


@pytest.mark.parametrize("seconds", [5, 30])
def test_full_size_prefill_logits(dev, full_std, seconds):
    """1.7B shapes: encoder features and last-position prefill logits (24 + 28 layers deep) against the oracle."""
    import oracle
    cfg, state, model = full_std
    torch.set_num_threads(ORACLE_THREADS)
    x = signal("speechlike", seconds * 16000, seed=70 + seconds)
    got, feats = _gpu_prefill_logits(model, x)
    res = {}
    for name, (enc, dec) in _oracles(cfg, state).items():
        f_ref = enc(oracle.log_mel_spectrogram(x))
        assert f_ref.shape == feats.shape
        dec.reset()
        prompt = oracle.build_prompt(f_ref.shape[0], [6364])
        ref = dec.forward(oracle.prepare_inputs(f_ref, prompt, dec.W["embed_tokens.weight"]))[-1]
        d = (got - ref).abs()
        res[name] = {"feat_max": round(float((feats - f_ref).abs().max()), 4), "feat_mean": round(float((feats - f_ref).abs().mean()), 5),
                     "logit_max": round(float(d.max()), 4), "logit_mean": round(float(d.mean()), 5),
                     "logit_absmax": round(float(ref.abs().max()), 3), "same_argmax": bool(int(got.argmax()) == int(ref.argmax())),
                     "top1_margin": round(float(torch.topk(ref, 2).values.diff().abs()), 4)}
    _record(f"full_prefill_logits_{seconds}s", res)
    g = res["gpu_rounding"]
    assert g["logit_max"] <= FULL_LOGIT_ATOL and g["logit_mean"] <= 0.02, g
    assert g["feat_max"] < 0.08 and g["feat_mean"] < 0.01, g
    r = res["reference"]                                              # the reference-semantics flavour at the same bound
    assert r["logit_max"] <= FULL_LOGIT_ATOL and r["logit_mean"] <= 0.02 and r["feat_max"] < 0.08, r
    assert all(v["same_argmax"] or v["top1_margin"] <= 2 * FULL_LOGIT_ATOL for v in res.values()), res


def test_full_size_margin_aware_tokens_512(dev, full_wide):
    """1.7B shapes, 4 clips (5 / 12 / 30 / 30 s) x 128 greedy tokens with the API-default penalty."""
    cfg, state, model = full_wide
    torch.set_num_threads(ORACLE_THREADS)
    clips = [signal("speechlike", n, seed=500 + i) for i, n in enumerate((80000, 192000, 480000, 480000))]
    tot = _run_margin_case(model, cfg, state, clips, 128, 1.2, ULPS_FULL, "full_wide_embed_512")
    for g in (tot["gpu_rounding"], tot["reference"]):
        assert g["n"] == 512 and g["decisive_exact"] == g["decisive"] and g["decisive"] >= 128
        assert g["exact_rate"] >= 0.8


# ----------------------------------------------------------------------------- BASELINE.json configs[2] as stated
def test_config3_full_size_batch32_30s_graph_decode_equals_singles_and_oracle(dev, full_wide):
    """Qwen3-ASR-1.7B shapes, batch = 32, 30 s clips, greedy with the API-default penalty, hipGraph decode (the bench
    workload; reference generate.py:222-315 run once per clip).  Rows 0, 17 and 31 of the batch must equal the same clips
    decoded alone bit for bit (two m-tiles per decode GEMM against one, 8-phase against 128^2 GEMM tiles, packed against
    single prefill), and rows 17 and 31 are checked margin-aware against both oracle flavours."""
    import oracle
    cfg, state, model = full_wide
    torch.set_num_threads(ORACLE_THREADS)
    assert model.use_graph
    clips = [signal("speechlike", 480000, seed=1234 + i) for i in range(32)]
    toks = model.generate_tokens(clips, [128] * 32, "English", repetition_penalty=1.2)
    assert len(toks) == 32 and all(len(t) == 128 for t in toks)
    assert len({tuple(t) for t in toks}) == 32                              # 32 different clips, 32 different transcripts
    for r in (0, 17, 31):
        assert model.generate_tokens([clips[r]], [128], "English", repetition_penalty=1.2)[0] == toks[r], f"row {r} differs from its single run"
    model.use_graph = False
    try:                                                                    # eager decode of the whole batch: the same tokens
        assert model.generate_tokens(clips, [128] * 32, "English", repetition_penalty=1.2) == toks
    finally:
        model.use_graph = True
    oracles = _oracles(cfg, state)
    tot = {k: {"n": 0, "exact": 0, "decisive": 0, "decisive_exact": 0, "max_gap_in_tol_units": 0.0} for k in oracles}
    for r in (17, 31):
        mel = oracle.log_mel_spectrogram(clips[r])
        for name, (enc, dec) in oracles.items():
            st = _margin_check(_teacher_forced_logits(dec, enc(mel), toks[r]), toks[r], 1.2, ULPS_FULL, strict=False)
            for k, v in st.items():
                tot[name][k] = max(tot[name][k], v) if k.startswith("max") else tot[name][k] + v
    _record("config3_batch32_30s", {**tot, "rows_equal_to_singles": [0, 17, 31], "graph_equals_eager": True, "tol_ulps": ULPS_FULL})
    for t in tot.values():                  # strict on both flavours: every decisive step exact, every other step inside the tolerance
        assert t["n"] == 256 and t["decisive_exact"] == t["decisive"] and t["decisive"] >= 64 and t["max_gap_in_tol_units"] <= 1.0, tot


# ----------------------------------------------------------------------------- long unchunked contexts (the default chunk_duration is 1200 s)
def test_long_unchunked_context_matches_oracle(dev):
    """200 s of audio in ONE sequence (2,600 audio tokens; prefill attention over 41 query tiles, decode attention over three
    1,024-token cache chunks merged in chunk order): margin-aware greedy tokens against the oracle, small config."""
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import synth
    cfg = synth.small_config()
    state = synth.random_state(cfg, seed=9, bias_std=0.02, norm_jitter=0.1)
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
    clips = [signal("speechlike", 200 * 16000, seed=910)]
    tot = _run_margin_case(model, cfg, state, clips, 48, 1.2, ULPS_SMALL, "small_long_context_200s")
    g = tot["gpu_rounding"]
    assert g["n"] == 48 and g["decisive_exact"] == g["decisive"] and g["decisive"] >= 24


def test_full_size_20min_unchunked_runs_in_one_sequence(dev, full_std):
    """Exactly 20 min with the default chunk_duration is NOT chunked (strict >, reference model.py:313): one sequence of 15,600
    audio tokens + prompt at the 1.7B shapes (KV cache 1.8 GB at that length).  No CPU oracle at this size: the property
    checked is that the path runs, is deterministic, and that graph replay == eager."""
    cfg, state, model = full_std
    audio = np.concatenate([signal("speechlike", 16000 * 60, seed=100 + i) for i in range(20)])
    r1 = model.transcribe(audio, max_tokens=12)
    assert r1.duration == 1200.0 and isinstance(r1.text, str)
    toks = model.generate_tokens([audio], [12], "English")[0]
    assert len(toks) == 12 and model.generate_tokens([audio], [12], "English")[0] == toks
    model.use_graph = False
    try:
        assert model.generate_tokens([audio], [12], "English")[0] == toks
    finally:
        model.use_graph = True
    model._decoder._caches.clear()                                          # give the 1.8 GB cache back


# ----------------------------------------------------------------------------- reference-semantics encoder mode (fp32 activations)
@pytest.mark.parametrize("which", ["small", "full"])
def test_encoder_parity_mode_matches_reference_semantics_oracle(dev, full_std, which):
    """AudioEncoder.precision = "fp32": float32 activations end to end, as the reference (audio.py:278 + encoder.py:272-323).
    Against the oracle WITHOUT the GPU's rounding points the features must agree to fp32 accuracy -- two orders of magnitude
    tighter than the fast path's bound."""
    import oracle
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import synth
    if which == "small":
        cfg = synth.small_config()
        state = synth.random_state(cfg, seed=8, bias_std=0.02, norm_jitter=0.1)
        model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
        clips, atol = [signal("speechlike", n, seed=60 + i) for i, n in enumerate((16000, 48077, 480000))], 2e-5
    else:
        cfg, state, model = full_std
        clips, atol = [signal("speechlike", 80000, seed=75), signal("speechlike", 480000, seed=76)], 2e-4
    enc_ref = _oracles(cfg, state)["reference"][0]
    res = {}
    try:
        for x in clips:
            mel = q.log_mel_spectrogram(x)
            ref = enc_ref(oracle.log_mel_spectrogram(x))
            model._encoder.precision = "fp32"
            got = model._encoder(mel)[0].cpu()
            model._encoder.precision = "bf16"
            fast = model._encoder(mel)[0].cpu()
            assert got.shape == ref.shape
            res[f"{len(x) / 16000:g}s"] = {"parity_max": float((got - ref).abs().max()), "parity_mean": float((got - ref).abs().mean()),
                                         "fast_max": float((fast - ref).abs().max()), "fast_mean": float((fast - ref).abs().mean()),
                                         "ref_absmax": float(ref.abs().max())}
    finally:
        model._encoder.precision = "bf16"
    _record(f"encoder_parity_mode_{which}", res)
    for k, r in res.items():
        assert r["parity_max"] <= atol, (k, r)
        assert r["parity_max"] * 20 < r["fast_max"], (k, r)            # the fp32 mode really is a different regime


def test_fast_vs_parity_mode_token_agreement(dev, full_wide):
    """1.7B shapes, 4 clips x 128 greedy tokens: encoder in the fast (bf16 operands) and in the reference-semantics (fp32)
    mode.  Recorded: token agreement between the two modes, and each mode's margin-aware agreement with the
    reference-semantics oracle (fp32 encoder activations, unrounded softmax probabilities)."""
    cfg, state, model = full_wide
    torch.set_num_threads(ORACLE_THREADS)
    clips = [signal("speechlike", n, seed=500 + i) for i, n in enumerate((80000, 192000, 480000, 480000))]
    fast = model.generate_tokens(clips, [128] * 4, "English", repetition_penalty=1.2)
    try:
        model._encoder.precision = "fp32"
        tot = _run_margin_case(model, cfg, state, clips, 128, 1.2, ULPS_FULL, "full_wide_embed_512_parity_encoder")
        par = model.generate_tokens(clips, [128] * 4, "English", repetition_penalty=1.2)
    finally:
        model._encoder.precision = "bf16"
    same = sum(a == b for ta, tb in zip(fast, par) for a, b in zip(ta, tb))
    first_div = [next((i for i, (a, b) in enumerate(zip(ta, tb)) if a != b), len(ta)) for ta, tb in zip(fast, par)]
    _record("fast_vs_parity_encoder_tokens", {"n": 512, "same_position_tokens": same, "rate": round(same / 512, 4),
                                              "first_divergence_per_clip": first_div})
    assert all(t["decisive_exact"] == t["decisive"] for t in tot.values())


# ----------------------------------------------------------------------------- fp8 MFMA prefill (configs[4]) against the same-arithmetic oracle
def _fp8_effective_state(state):
    """The checkpoint with every quantised tensor replaced by its dequantised value q * s (oracle quantiser, not the product's)."""
    import oracle.decoder as od
    out = dict(state)
    for k, v in state.items():
        if k.startswith("model.") and (k.endswith("_proj.weight") or k == "model.embed_tokens.weight"):
            q, s = od.quantize_weight_rows_e4m3(v.float().cpu())
            out[k] = (q * s[:, None]).to(torch.bfloat16)
    return out


@pytest.mark.parametrize("which", ["small", "full"])
def test_fp8_mfma_prefill_matches_the_e4m3_oracle(dev, which):
    """weight_dtype="fp8": prefill projections fp8 x fp8 on the fp8 MFMA with per-token activation scales (decode steps keep
    bf16 activations).  Checked STRICTLY against the oracle flavour that performs the same quantisations (e4m3 weights per
    output row, e4m3 activations per token at the RMSNorm / attention / SwiGLU outputs of the prompt rows): last-position
    logits and margin-aware greedy tokens, every decisive step exact, every other step inside the tolerance.  The weight-only
    variant "fp8w" is held to the bf16 model's bar against the dequantised-weight oracle; "fp8" to FP8_AMPLIFICATION times that
    bar (the discontinuity of e4m3 rounding, see the constant), with the counts at the bf16 bar recorded beside it; and each
    weight-only mode must sit closer to its own oracle flavour than to the other one (at 28 layers two W8A8 runs share too
    little of their quantisation noise for that to hold the other way round: recorded)."""
    import oracle
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import ops, synth
    from qwen3_asr_mlx_amd.config import ModelConfig
    torch.set_num_threads(ORACLE_THREADS)
    if which == "small":
        cfg = synth.small_config()
        state = synth.random_state(cfg, seed=2, device=dev, bias_std=0.02, norm_jitter=0.1)
        clips = [signal("speechlike", n, seed=700 + i) for i, n in enumerate((80000, 160000, 240000, 480000))]
        ulps = ULPS_SMALL
    else:
        cfg = ModelConfig()
        state = synth.random_state(cfg, seed=0, device=dev, embed_std=0.08)
        clips = [signal("speechlike", n, seed=700 + i) for i, n in enumerate((80000, 480000))]
        ulps = ULPS_FULL
    eff = {k: v.cpu() for k, v in _fp8_effective_state(state).items()}
    for k in ("model.embed_tokens.weight", "model.layers.0.mlp.down_proj.weight"):     # the product quantiser makes the same weights
        assert torch.equal(ops.quantize_fp8_rows(state[k])[2].cpu(), eff[k])
    enc, dec = _oracles(cfg, eff)["gpu_rounding"]
    res = {}
    for mode in ("fp8w", "fp8"):
        f8 = mode == "fp8"
        tol_ulps = ulps * (FP8_AMPLIFICATION[which] if f8 else 1.0)
        model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype=mode)
        got, _ = _gpu_prefill_logits(model, clips[-1])
        f_ref = enc(oracle.log_mel_spectrogram(clips[-1]))
        emb = oracle.prepare_inputs(f_ref, oracle.build_prompt(f_ref.shape[0], [6364]), dec.W["embed_tokens.weight"])
        dec.reset()
        ref = dec.forward(emb, fp8_rows=emb.shape[0] if f8 else 0)[-1]
        dec.reset()
        ref_other = dec.forward(emb, fp8_rows=0 if f8 else emb.shape[0])[-1]    # the OTHER flavour: the GPU mode must sit closer to its own
        toks_all = model.generate_tokens(clips, [128] * len(clips), "English", repetition_penalty=1.2)
        tot = {"n": 0, "exact": 0, "decisive": 0, "decisive_exact": 0, "max_gap_in_tol_units": 0.0}
        at_bf16_bar = {"decisive": 0, "decisive_exact": 0, "max_gap_in_tol_units": 0.0}
        for x, toks in zip(clips, toks_all):
            lg = _teacher_forced_logits(dec, enc(oracle.log_mel_spectrogram(x)), toks, fp8_prompt=f8)
            st = _margin_check(lg, toks, 1.2, tol_ulps, strict=False)
            for k, v in st.items():
                tot[k] = max(tot[k], v) if k.startswith("max") else tot[k] + v
            if f8:                                                             # recorded: the same steps held to the bf16 model's bar
                sb = _margin_check(lg, toks, 1.2, ulps, strict=False)
                for k in at_bf16_bar:
                    at_bf16_bar[k] = max(at_bf16_bar[k], sb[k]) if k.startswith("max") else at_bf16_bar[k] + sb[k]
        top = float(ref.abs().max())
        res[mode] = {"logit_max_diff": round(float((got - ref).abs().max()), 4), "logit_mean_diff": round(float((got - ref).abs().mean()), 5),
                     "logit_absmax": round(top, 3), "logit_tol": tol_ulps * _ulp_bf16(top),
                     "same_argmax": bool(int(got.argmax()) == int(ref.argmax())),
                     "top1_margin": round(float(torch.topk(ref, 2).values.diff().abs()), 4),
                     "vs_other_flavour_max": round(float((got - ref_other).abs().max()), 4),
                     "vs_other_flavour_mean": round(float((got - ref_other).abs().mean()), 5),
                     "tokens": tot["n"], "exact_rate": round(tot["exact"] / tot["n"], 4), "decisive": tot["decisive"],
                     "decisive_exact": tot["decisive_exact"], "max_gap_in_tol_units": round(tot["max_gap_in_tol_units"], 2), "tol_ulps": tol_ulps}
        if f8:
            res[mode]["at_the_bf16_bar"] = {**at_bf16_bar, "max_gap_in_tol_units": round(at_bf16_bar["max_gap_in_tol_units"], 2), "tol_ulps": ulps}
        del model
    _record(f"fp8_mfma_prefill_{which}", res)
    f8r, wr = res["fp8"], res["fp8w"]
    # weight-only: the bf16 model's own bar, margin-aware and strict
    assert wr["logit_max_diff"] <= wr["logit_tol"], wr
    assert wr["same_argmax"] or wr["top1_margin"] <= 2 * wr["logit_tol"], wr
    assert wr["decisive_exact"] == wr["decisive"] and wr["max_gap_in_tol_units"] <= 1.0, wr
    # W8A8: the amplified tolerance is a sanity bound only -- at the 1.7B shapes it is 4 logit units wide and NO step is
    # decisive under it (round-4 review, weak point 2), so nothing is asserted on decisive steps at that bar.  What pins the
    # fp8 arithmetic at full depth is test_all_28_layers_prefill_and_decode_stage_by_stage_on_a_real_prompt (every stage of
    # every layer from identical inputs, amplification-free); what is asserted here end to end is the distance to its own
    # oracle flavour relative to the weight-only mode and, below, the decisive steps AT THE BF16 MODEL'S BAR.
    assert f8r["logit_max_diff"] <= f8r["logit_tol"], f8r
    assert wr["decisive"] >= (300 if which == "small" else 32), res
    assert wr["logit_mean_diff"] < wr["vs_other_flavour_mean"], res            # weight-only sits with the bf16-activation oracle ...
    assert f8r["logit_mean_diff"] <= 8.0 * wr["logit_mean_diff"], res          # ... W8A8 within the amplification of ITS oracle (CPU: 3-7 x)
    b = f8r["at_the_bf16_bar"]                                                  # and, held to the bf16 model's bar, the decisive steps still agree
    assert b["decisive"] >= (300 if which == "small" else 64) and b["decisive_exact"] >= b["decisive"] - (0 if which == "small" else 2), res


def _stage_cmp(got, ref):
    """Element-wise agreement of one kernel stage with the oracle run on the SAME inputs.  One ulp = the bf16 spacing at
    max(|ref element|, |got element|, rms of the row): an element much smaller than its row's scale is not held to its own
    tiny ulp (its error is set by the terms that cancelled into it), and two NEIGHBOURING bf16 values on either side of a
    power of two are one ulp apart, not two (round 5: with the spacing taken at the reference alone, a one-step difference just
    below a binade boundary read as 1.5-2 "ulps" at the 1.7B shapes' real activations)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    rms = ref.pow(2).mean(-1, keepdim=True).sqrt()
    ulp = torch.exp2(torch.floor(torch.log2(torch.maximum(torch.maximum(ref.abs(), got.abs()), rms).clamp_min(1e-30))) - 7)
    return {"equal": round(float((d == 0).float().mean()), 5), "within_1_ulp": round(float((d <= ulp).float().mean()), 5),
            "max_ulps": round(float((d / ulp).max()), 2)}


def _e4m3(t):
    return t.view(torch.float8_e4m3fn).float()


def _prefill_layer_stages(tr, Wl, dims, f8, pos0=0):
    """One decoder layer of a packed single-sequence prefill, kernel stage by kernel stage: every stage's GPU output against
    the oracle's arithmetic applied to the inputs the GPU ITSELF fed that stage (`tr` = one entry of
    decoder._layers_prefill(trace=...)).  Wl: the layer's float32 CPU weights keyed without the 'layers.N.' prefix.
    Returns {stage: agreement record}."""
    import oracle.decoder as od
    nh, nkv, hd, eps, theta, I = dims
    tr = {k: v.cpu() for k, v in tr.items()}
    T = tr["h_in"].shape[0]
    W8 = {k: od.quantize_weight_rows_e4m3(v) for k, v in Wl.items() if k.endswith("_proj.weight")} if f8 else {}

    def lin(x, names, x8=None, sa=None):
        """bf16-valued outputs of the projections `names` (concatenated): bf16 operands, or e4m3 bytes + per-token scales."""
        outs = []
        for n in names:
            if f8:
                qw, sw = W8[n]
                outs.append(od.bf((x8 @ qw.T) * (sw[None, :] * sa[:, None])))
            else:
                outs.append(od.bf(x @ Wl[n].T))
        return torch.cat(outs, dim=1)

    rec = {}

    def quant_stage(name, src, got8, gots):
        q8, sc = od.quantize_act_rows_e4m3(src)
        rec[name] = {"bytes_equal": round(float((_e4m3(got8) == q8).float().mean()), 5), "scales_equal": round(float((gots == sc).float().mean()), 5)}
        return _e4m3(got8), gots.float()

    # --- stage 1: input RMSNorm (+ per-token quantisation), then the q/k/v projections
    x1 = od.rms_norm(tr["h_in"].float(), Wl["input_layernorm.weight"], eps)
    qkv_names = ["self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight"]
    if f8:
        x8, sa = quant_stage("rmsnorm1_quant", x1, tr["x8_1"], tr["sx_1"])
        rec["qkv"] = _stage_cmp(tr["qkv"], lin(None, qkv_names, x8, sa))
    else:
        rec["rmsnorm1"] = _stage_cmp(tr["x_1"], x1)
        rec["qkv"] = _stage_cmp(tr["qkv"], lin(tr["x_1"].float(), qkv_names))
    # --- stage 2: QK-norm + RoPE + causal GQA attention, from the GPU's own qkv
    qkv = tr["qkv"].float()
    pos = torch.arange(pos0, pos0 + T)
    qh = od.rope(od.rms_norm(qkv[:, : nh * hd].reshape(T, nh, hd), Wl["self_attn.q_norm.weight"], eps).transpose(0, 1), pos, theta)
    kh = od.rope(od.rms_norm(qkv[:, nh * hd: (nh + nkv) * hd].reshape(T, nkv, hd), Wl["self_attn.k_norm.weight"], eps).transpose(0, 1), pos, theta)
    vh = qkv[:, (nh + nkv) * hd:].reshape(T, nkv, hd).transpose(0, 1)
    rec["q_norm_rope"] = _stage_cmp(tr["q"], qh.transpose(0, 1).reshape(T, nh * hd))
    sc_ = (qh @ kh.repeat_interleave(nh // nkv, dim=0).transpose(1, 2)) * hd ** -0.5
    sc_ = sc_.masked_fill(~(pos[None, :] <= pos[:, None])[None], float("-inf"))
    pr = torch.exp(sc_ - sc_.max(-1, keepdim=True).values)
    att = od.bf((od.bf(pr) @ vh.repeat_interleave(nh // nkv, dim=0)) / pr.sum(-1, keepdim=True)).transpose(0, 1).reshape(T, nh * hd)
    rec["attention"] = _stage_cmp(tr["att"], att)
    # --- stage 3: o projection + residual, from the GPU's own attention output
    if f8:
        a8, sa = quant_stage("attention_quant", tr["att"].float(), tr["a8"], tr["sa"])
        o = lin(None, ["self_attn.o_proj.weight"], a8, sa)
    else:
        o = lin(tr["att"].float(), ["self_attn.o_proj.weight"])
    rec["o_proj_residual"] = _stage_cmp(tr["h_mid"], od.bf(tr["h_in"].float() + o))
    # --- stage 4: post-attention RMSNorm, gate/up + SwiGLU, from the GPU's own residual stream
    x2 = od.rms_norm(tr["h_mid"].float(), Wl["post_attention_layernorm.weight"], eps)
    if f8:
        x8, sa = quant_stage("rmsnorm2_quant", x2, tr["x8_2"], tr["sx_2"])
        gu = lin(None, ["mlp.gate_proj.weight", "mlp.up_proj.weight"], x8, sa)
    else:
        rec["rmsnorm2"] = _stage_cmp(tr["x_2"], x2)
        gu = lin(tr["x_2"].float(), ["mlp.gate_proj.weight", "mlp.up_proj.weight"])
    gt, up = gu[:, :I], gu[:, I:]
    rec["gate_up_swiglu"] = _stage_cmp(tr["act"], od.swiglu(gt, up))
    # --- stage 5: down projection + residual, from the GPU's own activation
    if f8:
        c8, sc8 = quant_stage("swiglu_quant", tr["act"].float(), tr["c8"], tr["sc"])
        dn = lin(None, ["mlp.down_proj.weight"], c8, sc8)
    else:
        dn = lin(tr["act"].float(), ["mlp.down_proj.weight"])
    rec["down_proj_residual"] = _stage_cmp(tr["h_out"], od.bf(tr["h_mid"].float() + dn))
    return rec


def _assert_prefill_stage_bars(rec, f8, where=""):
    """Bars of one layer's stages (docstring of the one-layer test)."""
    gemm_equal = 0.98 if f8 else 0.999          # the fp8 MFMA's accumulation differs from an fp32 dot product in more last bits
    for name, r in rec.items():
        if name.endswith("_quant"):
            assert r["bytes_equal"] >= 0.999 and r["scales_equal"] >= 0.99, (where, name, r)
        elif name in ("rmsnorm1", "rmsnorm2", "q_norm_rope"):
            assert r["equal"] >= 0.999 and r["within_1_ulp"] >= 0.9999 and r["max_ulps"] <= 2.0, (where, name, r)
        elif name in ("qkv", "o_proj_residual", "down_proj_residual"):
            assert r["equal"] >= gemm_equal and r["within_1_ulp"] >= 0.999 and r["max_ulps"] <= 2.0, (where, name, r)
        elif name == "gate_up_swiglu":
            assert r["equal"] >= 0.97 and r["within_1_ulp"] >= 0.999 and r["max_ulps"] <= 6.0, (where, name, r)
        elif name == "attention":
            assert r["equal"] >= 0.75 and r["within_1_ulp"] >= 0.999 and r["max_ulps"] <= 2.0, (where, name, r)
        else:
            assert name == "end_to_end", name


def _worst(recs):
    """Per stage, the worst figure over a list of per-layer records."""
    out = {}
    for rec in recs:
        for name, r in rec.items():
            o = out.setdefault(name, dict(r))
            for k, v in r.items():
                o[k] = max(o[k], v) if k == "max_ulps" else min(o[k], v)
    return out


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_one_full_size_layer_stage_by_stage_from_identical_inputs(dev, mode):
    """ONE decoder layer at the 1.7B dimensions (hidden 2,048, 16/8 heads, MLP 6,144) over 408 packed prompt rows, checked
    kernel stage by kernel stage: every stage's output on the GPU against the oracle's arithmetic applied to the inputs the
    GPU ITSELF fed that stage (decoder._layers_prefill(trace=...)).  No noise is carried from stage to stage, so nothing is
    amplified: a slip anywhere in the chain -- which operand is quantised and with which row's scale, e4m3 rounding, where
    the residual enters, the SwiGLU pairing -- shows at its stage, in bf16 and in fp8 mode alike.  (End to end, the same
    layer differs from the oracle in 46 % of its elements in bf16 mode and 92 % in fp8 mode, by up to 5 and 40 ulps: e4m3
    rounding amplifies the one-ulp differences the attention hands it, which is why the full-depth fp8 test above has a
    wider bar -- measured here, recorded as `end_to_end`.)  Synthetic input rows of very different scale (one row x 30).
    Bars: quantised bytes and per-token scales identical (>= 99.9 %; measured 100 %); norms and RoPE >= 99.9 % identical,
    >= 99.99 % within one ulp, never further than 2 (over 28 layers of real activations a handful of 11 M RoPE outputs sit 2 ulps
    off: sincosf against torch's); GEMM stages (+ residual) >= 99.9 % identical in bf16 mode and >= 98 % in fp8 mode, >= 99.9 %
    within one ulp, never further than 2; SwiGLU (three bf16 roundings, which pass a 1-2 ulp difference of the gate on as up to
    5) >= 97 % identical, >= 99.9 % within one ulp, never further than 6; attention (online softmax against a two-pass softmax)
    >= 75 % identical, >= 99.9 % within one ulp, never further than 2."""
    import oracle
    import oracle.decoder as od
    from qwen3_asr_mlx_amd import synth
    from qwen3_asr_mlx_amd.config import ModelConfig, TextDecoderConfig
    from qwen3_asr_mlx_amd.decoder import KVCache, TextDecoder
    torch.set_num_threads(ORACLE_THREADS)
    full = ModelConfig()
    tcfg = TextDecoderConfig(**{**full.text_decoder.__dict__, "num_hidden_layers": 1, "vocab_size": 4096})
    cfg = ModelConfig(audio_encoder=synth.small_config().audio_encoder, text_decoder=tcfg)
    state = synth.random_state(cfg, seed=21, device=dev, norm_jitter=0.1)
    W = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
    f8 = mode == "fp8"
    dec = TextDecoder(tcfg)
    dec.load_state(W, dev, "fp8" if f8 else "bf16")
    T, H, I = 408, tcfg.hidden_size, tcfg.intermediate_size
    nh, nkv, hd, eps = tcfg.num_attention_heads, tcfg.num_key_value_heads, tcfg.head_dim, tcfg.rms_norm_eps
    g = torch.Generator().manual_seed(5)
    h0 = (torch.randn(T, H, generator=g) * 0.5).to(torch.bfloat16)
    h0[7] *= 30.0                                                              # rows of very different scale (per-token scales)
    h = h0.to(dev).clone()
    cache = KVCache()
    cache.ensure(tcfg, 1, 448, dev)
    trace: list = []
    dec._layers_prefill(h, cache, [T], trace=trace)
    Wf = {k: v.float().cpu() for k, v in W.items()}
    Wl = {k[len("layers.0."):]: v for k, v in Wf.items() if k.startswith("layers.0.")}
    rec = _prefill_layer_stages(trace[0], Wl, (nh, nkv, hd, eps, tcfg.rope_theta, I), f8)
    assert torch.equal(trace[0]["h_out"], h)
    # --- and the whole layer end to end against the oracle (recorded: this is what amplification does to it)
    W8 = {k: od.quantize_weight_rows_e4m3(v) for k, v in Wf.items() if k.endswith("_proj.weight")} if f8 else {}
    Wd = {k: (W8[k][0] * W8[k][1][:, None] if k in W8 else v) for k, v in Wf.items()}
    orc = oracle.DecoderOracle(Wd, nh, nkv, hd, eps, tcfg.rope_theta, gpu_rounding=True)
    rec["end_to_end"] = _stage_cmp(h.cpu(), orc.forward(h0.float(), fp8_rows=T if f8 else 0, return_hidden=True))
    _record(f"one_layer_stages_{mode}", rec)
    _assert_prefill_stage_bars(rec, f8)


@pytest.fixture(scope="module")
def full_state(dev):
    """The 1.7B shapes' synthetic checkpoint on the device (shared by the two 28-layer stage tests)."""
    from qwen3_asr_mlx_amd import synth
    from qwen3_asr_mlx_amd.config import ModelConfig
    cfg = ModelConfig()
    return cfg, synth.random_state(cfg, seed=0, device=dev, norm_jitter=0.1)


def _real_prompt_embeddings(model, x, n_copies=1):
    """Packed bf16 prompt embeddings of a real request (mel -> encoder -> build_prompt -> audio-pad merge), product kernels."""
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd.generate import _merge
    from qwen3_asr_mlx_amd.tokenizer import build_prompt
    dev = model._decoder.device
    audio, lens_dev, lens = q.audio.pack_audio([x] * n_copies, dev)
    mel, n_frames = q.audio.get_frontend(dev)(audio, lens_dev, lens)
    feats, counts = model._encoder.forward_batch(mel, n_frames)
    prompts = [build_prompt(c, [6364]) for c in counts]
    return _merge(feats, prompts, counts, model._decoder), [len(p) for p in prompts]


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_all_28_layers_prefill_and_decode_stage_by_stage_on_a_real_prompt(dev, full_state, mode):
    """The round-4 review's item 1(a): the one-layer stage test at FULL DEPTH on a REAL prompt.  The real 1.7B-shape
    TextDecoder (28 layers) prefills the 408 prompt rows of a 30 s clip (mel -> encoder -> prompt merge through the product
    kernels); `_layers_prefill(trace=...)` hands out every intermediate of every layer, and each of the 28 x 9 (bf16) /
    28 x 11 (fp8: + quantised bytes and per-token scales) stages is checked against the oracle's arithmetic applied to the
    inputs the GPU itself fed that stage -- amplification-free, so the ONE-layer bars apply unchanged at every depth: a
    layer-indexed slip in tile / scale addressing of layers 1..27 would show at its layer.  Then ONE decode step of a batch
    of 3 (the 7-launch form the bench runs) on the caches the prefill filled, every layer's launches the same way: in fp8
    mode that is the W8A16 step (fp8 weight bytes widened in registers, bf16 activations).  The worst figure per stage over
    the 28 layers is recorded; the layer index of any failure is in the assertion message."""
    import oracle.decoder as od
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import ops, synth
    torch.set_num_threads(ORACLE_THREADS)
    cfg, state = full_state
    f8 = mode == "fp8"
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype="fp8" if f8 else "bf16")
    dec, t = model._decoder, cfg.text_decoder
    nh, nkv, hd, eps, I, H = t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.rms_norm_eps, t.intermediate_size, t.hidden_size
    dims = (nh, nkv, hd, eps, t.rope_theta, I)
    x = signal("speechlike", 480000, seed=4321)
    emb, lens = _real_prompt_embeddings(model, x)
    T = lens[0]
    assert T == 408 and emb.shape == (T, H)

    def layer_weights(li, dequantised):
        """float32 CPU weights of layer li, keyed without the prefix; `dequantised`: the projections as q * s (what a W8A16
        step multiplies by -- exactly representable in bf16)."""
        P = f"model.layers.{li}."
        Wl = {k[len(P):]: v.float().cpu() for k, v in state.items() if k.startswith(P)}
        if dequantised:
            for k in list(Wl):
                if k.endswith("_proj.weight"):
                    qw, sw = od.quantize_weight_rows_e4m3(Wl[k])
                    Wl[k] = qw * sw[:, None]
        return Wl

    # ---------------- prefill: 28 layers, one real sequence
    cache = dec.batch_cache(3, 448)
    h = emb.clone()
    trace: list = []
    dec._layers_prefill(h, cache, [T], slots=[0], trace=trace)
    assert len(trace) == t.num_hidden_layers == 28
    recs = []
    for li in range(28):
        recs.append(_prefill_layer_stages(trace[li], layer_weights(li, False), dims, f8))
        if li + 1 < 28:
            assert torch.equal(trace[li]["h_out"], trace[li + 1]["h_in"])       # the stream one layer hands the next is what was checked
    assert torch.equal(trace[-1]["h_out"], h)
    _record(f"all_layers_prefill_stages_{mode}", {"layers": 28, "rows": T, "worst_over_layers": _worst(recs)})
    for li, rec in enumerate(recs):
        _assert_prefill_stage_bars(rec, f8, where=f"prefill layer {li}")
    del trace

    # ---------------- one decode step of a batch of 3 on real caches: rows 1 and 2 hold shorter prefixes of the same prompt
    pre = [T, 300, 129]
    h3 = torch.cat([emb[:n] for n in pre]).clone()
    dec._layers_prefill(h3, cache, pre)
    ws = dec.workspace(3, 4)
    g = torch.Generator().manual_seed(77)
    toks = torch.randint(0, t.vocab_size, (3,), generator=g).to(torch.int32)
    ws.cur_tok.copy_(toks.to(dev))
    ws.pos.copy_(torch.tensor(pre, dtype=torch.int32, device=dev))
    k_before = [cache.k[li].clone() for li in (0, 27)]
    ops.call("q3asr_embed_norm_x16", ops.ptr(ws.cur_tok), ops.ptr(dec.w["embed"]), t.vocab_size, ops.ptr(ws.h), ops.ptr(dec.w["layers"][0]["ln1"]),
             ops.ptr(ws.x16), 3, H, eps, ops.sp())
    dtrace: list = []
    dec._layers_decode(ws, cache, ws.pos, trace=dtrace)
    assert len(dtrace) == 28
    drecs = []
    for li in range(28):
        tr = {k: (v.cpu() if hasattr(v, "cpu") else v) for k, v in dtrace[li].items()}
        Wl = layer_weights(li, f8)
        rec = {}

        def slab_sum(sl):                                        # the consumers add slabs in order 0, 1, 2, ... in float32
            acc = sl[0].float().clone()
            for s_ in range(1, sl.shape[0]):
                acc = acc + sl[s_].float()
            return od.bf(acc)

        qkv_names = ["self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight"]
        qkv = slab_sum(tr["qkv_slabs"])
        rec["qkv"] = _stage_cmp(qkv, torch.cat([od.bf(tr["x_in"].float() @ Wl[n].T) for n in qkv_names], dim=1))
        # attention from the GPU's own qkv and the GPU's own cache (keys / values 0 .. pos-1 as the prefill wrote them)
        att_ref = torch.empty(3, nh * hd)
        k_app = []
        kc, vc = cache.k[li].float().cpu(), cache.v[li].float().cpu()
        for b in range(3):
            pos = torch.tensor([pre[b]])
            qb = od.rope(od.rms_norm(qkv[b, : nh * hd].reshape(1, nh, hd), Wl["self_attn.q_norm.weight"], eps).transpose(0, 1), pos, t.rope_theta)
            kb = od.rope(od.rms_norm(qkv[b, nh * hd: (nh + nkv) * hd].reshape(1, nkv, hd), Wl["self_attn.k_norm.weight"], eps).transpose(0, 1), pos, t.rope_theta)
            vb = qkv[b, (nh + nkv) * hd:].reshape(1, nkv, hd).transpose(0, 1)
            # the appended cache row: V is the projection's bf16 value as it stands, K went through QK-norm + RoPE (sincosf on
            # the GPU against torch's: compared as a stage); the attention below then reads the GPU's OWN K/V rows 0 .. pos
            assert torch.equal(vc[b, :, pre[b]], vb[:, 0]), f"decode layer {li}: appended V row differs"
            k_app.append((kc[b, :, pre[b]].reshape(-1), kb[:, 0].reshape(-1)))
            K = kc[b, :, : pre[b] + 1].repeat_interleave(nh // nkv, dim=0)
            V = vc[b, :, : pre[b] + 1].repeat_interleave(nh // nkv, dim=0)
            sc_ = (qb @ K.transpose(1, 2)) * hd ** -0.5
            pr = torch.exp(sc_ - sc_.max(-1, keepdim=True).values)
            att_ref[b] = od.bf((od.bf(pr) @ V) / pr.sum(-1, keepdim=True)).transpose(0, 1).reshape(nh * hd)
        rec["k_norm_rope_append"] = _stage_cmp(torch.stack([g_ for g_, _ in k_app]), torch.stack([r_ for _, r_ in k_app]))
        rec["attention"] = _stage_cmp(tr["att"], att_ref)
        o = slab_sum(tr["o_slabs"])
        rec["o_proj"] = _stage_cmp(o, od.bf(tr["att"].float() @ Wl["self_attn.o_proj.weight"].T))
        h_mid = od.bf(tr["h_in"].float() + o)
        rec["residual_rmsnorm2"] = _stage_cmp(torch.cat([tr["h_mid"].float(), tr["x_2"].float()], 1),
                                              torch.cat([h_mid, od.rms_norm(h_mid, Wl["post_attention_layernorm.weight"], eps)], 1))
        x2 = tr["x_2"].float()
        rec["gate_up_swiglu"] = _stage_cmp(tr["act"], od.swiglu(od.bf(x2 @ Wl["mlp.gate_proj.weight"].T), od.bf(x2 @ Wl["mlp.up_proj.weight"].T)))
        dn = slab_sum(tr["down_slabs"])
        rec["down_proj"] = _stage_cmp(dn, od.bf(tr["act"].float() @ Wl["mlp.down_proj.weight"].T))
        h_out = od.bf(tr["h_mid"].float() + dn)
        nw = state[f"model.layers.{li + 1}.input_layernorm.weight"] if li + 1 < 28 else state["model.norm.weight"]
        rec["residual_rmsnorm_next"] = _stage_cmp(torch.cat([tr["h_out"].float(), tr["x_next"].float()], 1),
                                                  torch.cat([h_out, od.rms_norm(h_out, nw.float().cpu(), eps)], 1))
        if li + 1 < 28:
            assert torch.equal(dtrace[li]["x_next"], dtrace[li + 1]["x_in"]) and torch.equal(dtrace[li]["h_out"], dtrace[li + 1]["h_in"])
        drecs.append(rec)
    for li, kb in zip((0, 27), k_before):                        # the step wrote ONE cache row per sequence and nothing else
        diff = (cache.k[li] != kb).any(dim=-1).any(dim=1)        # [B][Lmax]
        assert [int(i) for i in diff.nonzero()[:, 1]] == pre and int(diff.sum()) == 3
    _record(f"all_layers_decode_step_stages_{mode}", {"layers": 28, "batch": 3, "positions": pre, "weights": "W8A16" if f8 else "bf16",
                                                      "worst_over_layers": _worst(drecs)})
    # bars of a decode-step stage (3 rows per stage: rates are in steps of 1 / 6,144 .. 1 / 18,432 elements): residual + RMSNorm
    # >= 99.9 % identical, never further than 1 ulp; the split-K GEMMs (slabs summed in fp32, one rounding) >= 99.5 % identical,
    # never further than 2; SwiGLU >= 97 % identical, never further than 4; attention >= 75 % identical, never further than 2
    for li, rec in enumerate(drecs):
        for name, r in rec.items():
            where = (f"decode layer {li}", name, r)
            if name.startswith("residual") or name == "k_norm_rope_append":
                assert r["equal"] >= 0.995 and r["max_ulps"] <= 2.0, where
            elif name in ("qkv", "o_proj", "down_proj"):
                assert r["equal"] >= 0.995 and r["within_1_ulp"] >= 0.999 and r["max_ulps"] <= 2.0, where
            elif name == "gate_up_swiglu":
                assert r["equal"] >= 0.97 and r["within_1_ulp"] >= 0.999 and r["max_ulps"] <= 6.0, where
            else:
                assert r["equal"] >= 0.75 and r["within_1_ulp"] >= 0.999 and r["max_ulps"] <= 2.0, where
    del model
    dec._caches.clear()


# ----------------------------------------------------------------------------- BASELINE.json configs[4] at full size
def test_config5_full_size_longform_fp8_sampling(dev):
    """Qwen3-ASR-1.7B shapes, fp8 weights, 20 min of audio with chunk_duration=30 (low-energy split, reference
    model.py:382-447), temperature 0.7 / top-p 0.9: 40 segments, each with the reference budget max(256, 50/s), decoded
    as one batch; same seed -> same transcript."""
    import qwen3_asr_mlx_amd as q
    import importlib
    from qwen3_asr_mlx_amd import synth
    gen = importlib.import_module("qwen3_asr_mlx_amd.generate")     # the package attribute `generate` is the function
    from qwen3_asr_mlx_amd.config import ModelConfig
    from qwen3_asr_mlx_amd.model import _find_split_points
    cfg = ModelConfig()
    state = synth.random_state(cfg, seed=0, device=dev)
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev, weight_dtype="fp8")
    del state
    assert model._decoder.weight_dtype == "fp8"
    audio = np.concatenate([signal("speechlike", 16000 * 60, seed=100 + i) for i in range(20)])
    assert len(audio) == 1200 * 16000
    pts = _find_split_points(audio, 30 * 16000, 5 * 16000)
    assert len(pts) == 39 and all(abs(p - 30 * 16000 * (i + 1)) <= 5 * 16000 for i, p in enumerate(pts))
    # exactly 20 min is NOT chunked with the default chunk_duration (strict >, model.py:313): config 5 passes 30
    calls = []
    real = model.generate_tokens

    def spy(batch, max_tokens, lang, *a, **kw):
        out = real(batch, max_tokens, lang, *a, **kw)
        calls.append(([len(b) for b in batch], list(max_tokens), [len(t) for t in out], kw.get("seed")))
        return out
    model.generate_tokens = spy
    texts = []
    for _ in range(2):
        torch.manual_seed(1234)
        gen._gen_calls = 0                                              # the seed of a sampled run = f(torch seed, call count)
        r = model.transcribe(audio, temperature=0.7, top_p=0.9, chunk_duration=30.0)
        texts.append(r.text)
        assert r.duration == 1200.0 and r.language == "English"
    assert texts[0] == texts[1] and len(texts[0]) > 0
    assert len(calls) == 2 and calls[0][3] == calls[1][3]
    lens, budgets, n_out, _ = calls[0]
    assert len(lens) == 40 and sum(lens) == len(audio)
    assert sorted(lens) == sorted(b - a for a, b in zip([0] + pts, pts + [len(audio)]))
    assert all(b == max(256, int(n / 16000 * 50)) for n, b in zip(lens, budgets))
    assert all(1 <= n <= b for n, b in zip(n_out, budgets))
    assert sum(n == b for n, b in zip(n_out, budgets)) >= 36            # random weights: an EOS draw is rare (2 ids of 151,936)
    torch.manual_seed(99)
    gen._gen_calls = 0
    assert model.transcribe(audio, temperature=0.7, top_p=0.9, chunk_duration=30.0).text != texts[0]
    _record("config5_full_size", {"segments": 40, "tokens": int(sum(n_out)), "reproducible": True})
