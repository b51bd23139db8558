"""CPU suite (-m "not gpu"): the oracle against the reference-generated golden vectors and the
reference tests' known answers; the host logic of the product package; the C-ABI's symbol table."""
import ctypes
import hashlib
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle
from signals import signal

GOLD = Path(__file__).parent / "golden"
ROOT = Path(__file__).parent.parent


# ----------------------------------------------------------------------------- mel (pinned to the live reference)
def test_filterbank_bit_exact_and_quirks():
    fb = np.load(GOLD / "mel_filterbank.npy")
    got = oracle.mel_filterbank()
    assert got.dtype == np.float32 and got.shape == (128, 201)
    assert np.array_equal(got, fb)
    assert [i for i in range(128) if not fb[i].any()] == [0, 3, 6, 13]      # four dead rows
    assert (fb >= 0).all() and abs(float(fb.max()) - 0.02908) < 1e-4
    assert oracle.mel_filterbank() is oracle.mel_filterbank()               # cached (reference test_audio.py:103-116)


def test_mel_bit_exact_against_reference_fixtures():
    full = np.load(GOLD / "mel_full.npz")
    for key in full.files:
        if "_n" in key:
            x = signal("noise", int(key.split("_n")[1]))
        else:
            kind, s = key.split("_")
            x = signal(kind, int(float(s[:-1]) * 16000))
        got = oracle.log_mel_spectrogram(x)
        assert got.dtype == np.float32
        assert np.array_equal(got, full[key]), key
    dg = json.loads((GOLD / "mel_digest.json").read_text())
    for kind in ("tone", "noise"):
        got = oracle.log_mel_spectrogram(signal(kind, 480000))
        assert hashlib.sha256(got.reshape(-1).tobytes()).hexdigest() == dg[f"{kind}_30s"]["sha256"]
    assert np.array_equal(oracle.stft_power(signal("noise", 8000)), np.load(GOLD / "stft_power_noise_0.5s.npy"))


MEL_PARAM_SIGNALS = {"fft512_hop128_mels80": ("noise", 16000), "fft256_hop100_mels40_sr8k": ("speechlike", 8000),
                     "fft400_hop200": ("noise", 1234), "fft1024_hop256_mels64": ("tone", 16000),
                     "fft201_hop67_mels20_odd": ("noise", 5000), "fft400_hop160_mels128_fmax4k": ("speechlike", 16000),
                     "fft64_hop16_mels8_short": ("noise", 40)}          # the inputs tools/gen_golden_mel_params.py used


def mel_param_cases():
    meta = json.loads((GOLD / "mel_params.json").read_text())
    arrs = np.load(GOLD / "mel_params.npz")
    for name, m in meta.items():
        kind, n = MEL_PARAM_SIGNALS[name]
        yield name, signal(kind, n), m["params"], arrs[name]


def test_mel_non_default_parameters_bit_exact_against_reference_fixtures():
    """audio.py:238-246 accepts any n_fft / hop_length / n_mels / sample_rate / f_min / f_max; fixtures from the live reference
    (tools/gen_golden_mel_params.py), odd n_fft and a 2-frame signal included."""
    n = 0
    for name, x, params, want in mel_param_cases():
        p = dict(params)
        got = oracle.log_mel_spectrogram(x, p.pop("n_fft"), p.pop("hop_length"), p.pop("n_mels"), **p)
        assert got.shape == want.shape and np.array_equal(got, want), name
        n += 1
    assert n == 7


def test_mel_reference_test_properties():
    """reference tests/test_audio.py:49-89."""
    assert oracle.log_mel_spectrogram(signal("noise", 16000)).shape == (128, 100)
    assert np.allclose(oracle.log_mel_spectrogram(np.zeros(16000, np.float32)), -1.5, atol=1e-6)
    m = oracle.log_mel_spectrogram(signal("noise", 16000))
    assert m.max() - m.min() <= 2.0 + 1e-6                                   # 8 decades / 4


# ----------------------------------------------------------------------------- host integer logic
def test_prompts_exact():
    pr = json.loads((GOLD / "prompts.json").read_text())
    for n in (0, 13, 65, 390):
        assert oracle.build_prompt(n) == pr[f"n{n}_none"]
        assert oracle.build_prompt(n, [6364]) == pr[f"n{n}_lang"]
    p = oracle.build_prompt(3, [6364])
    assert p[:9] == [151644, 8948, 198, 151645, 198, 151644, 872, 198, 151669]      # reference test_tokenizer.py:48-73
    assert p[9:12] == [151676] * 3 and p[12:] == [151670, 151645, 198, 151644, 77091, 198, 11528, 6364, 151704]
    assert sorted(oracle.EOS_TOKEN_IDS) == pr["constants"]["EOS_TOKEN_IDS"]


def test_split_points_exact():
    sp = json.loads((GOLD / "split_points.json").read_text())
    sr = 16000
    c = sr * 10
    assert oracle.find_split_points(np.zeros(sr, np.float32), sr * 20, 5 * sr) == sp["short_zeros"] == []
    assert oracle.find_split_points(np.random.default_rng(0).random(int(c * 2.5)).astype(np.float32), c, 5 * sr) == sp["rng0_2.5chunks"]
    assert oracle.find_split_points(np.random.default_rng(1).random(c * 3).astype(np.float32), c, 5 * sr) == sp["rng1_3chunks"]
    s = np.ones(sr * 25, np.float32) * 0.5
    s[sr * 9: sr * 11] = 0.0
    got = oracle.find_split_points(s, c, 5 * sr)
    assert got == sp["silent_window"] and sr * 9 <= got[0] <= sr * 11
    assert oracle.find_split_points(np.random.default_rng(2).random(sr * 25).astype(np.float32), sr * 5, 2 * sr) == sp["rng2_5x5s"]
    assert oracle.find_split_points(signal("speechlike", 120 * sr), 30 * sr, 5 * sr) == sp["speechlike_120s_30s"]
    assert oracle.find_split_points(np.ones(100, np.float32), 50, 10) == sp["tiny_lt_frame"] == []


def test_chunk_and_window_bookkeeping():
    """reference test_encoder.py:65-89,161-164 and SURVEY a6."""
    assert [oracle.conv_output_length(n) for n in (100, 50, 1, 2, 3)] == [13, 7, 1, 1, 1]
    for frames, toks in ((100, 13), (300, 39), (250, 33), (3000, 390)):
        assert sum(oracle.chunk_layout(frames)[1]) == toks
    assert oracle.window_cu_seqlens(390, 104) == [0, 104, 208, 312, 390]
    assert oracle.window_cu_seqlens(65, 104) == [0, 65]
    assert oracle.window_cu_seqlens(208, 104) == [0, 104, 208]


# ----------------------------------------------------------------------------- sampling known answers (reference test_generate.py:132-194)
def test_oracle_sampling_known_answers():
    assert oracle.sample(torch.tensor([[0.1, 0.5, 0.2, 0.9, 0.3]])) == 3
    f = oracle.top_k_filter(torch.tensor([0.1, 0.9, 0.5, 0.3, 0.7]), 2)
    assert torch.isfinite(f).tolist() == [False, True, False, False, True]
    f = oracle.top_p_filter(torch.tensor([100.0, 0.0, 0.0, 0.0]), 0.99)
    assert torch.isfinite(f).tolist() == [True, False, False, False]
    pen = oracle.apply_repetition_penalty(torch.ones(10), [0, 1, 2], 2.0, bf16=False)
    assert torch.allclose(pen[:3], torch.full((3,), 0.5)) and torch.allclose(pen[3:], torch.ones(7))
    x = torch.tensor([1.0, 2.0, 3.0, 0.5])
    assert torch.equal(oracle.apply_repetition_penalty(x, [0, 1], 1.0), x)
    g = torch.Generator().manual_seed(0)
    assert all(oracle.sample(torch.tensor([[0.1, 0.5, 0.2, 0.9, 0.3]]), temperature=1.0, top_k=1, generator=g) == 3 for _ in range(10))
    assert len({oracle.sample(torch.ones(1, 512), temperature=2.0, generator=g) for _ in range(20)}) > 1


# ----------------------------------------------------------------------------- decoder properties (reference test_decoder.py)
def _tiny_decoder(seed=0, layers=2, H=64, nh=4, nkv=2, hd=16, I=128, V=512):
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16).float()
    W = {"embed_tokens.weight": bf(torch.randn(V, H, generator=g) * 0.5), "norm.weight": torch.ones(H)}
    for i in range(layers):
        p = f"layers.{i}."
        W[p + "input_layernorm.weight"] = torch.ones(H)
        W[p + "post_attention_layernorm.weight"] = torch.ones(H)
        W[p + "self_attn.q_norm.weight"] = torch.ones(hd)
        W[p + "self_attn.k_norm.weight"] = torch.ones(hd)
        for n, o, k in (("self_attn.q_proj", nh * hd, H), ("self_attn.k_proj", nkv * hd, H), ("self_attn.v_proj", nkv * hd, H),
                        ("self_attn.o_proj", H, nh * hd), ("mlp.gate_proj", I, H), ("mlp.up_proj", I, H), ("mlp.down_proj", H, I)):
            W[p + n + ".weight"] = bf(torch.randn(o, k, generator=g) * (k ** -0.5))
    return oracle.DecoderOracle(W, nh, nkv, hd, 1e-6, 1e6)


def test_rmsnorm_formula_and_rope_properties():
    import oracle.decoder as od
    x = torch.randn(3, 5, 64)
    ref = x / torch.sqrt((x ** 2).mean(-1, keepdim=True) + 1e-6)
    assert torch.allclose(od.rms_norm(x, torch.ones(64), 1e-6), od.bf(ref), atol=1e-4)       # test_decoder.py:77-89
    q = od.bf(torch.randn(2, 4, 6, 16))
    r = od.rope(q, torch.arange(6), 1e6)
    assert r.shape == q.shape
    assert torch.allclose(r.norm(dim=-1), q.norm(dim=-1), rtol=2e-2)                           # norm preserving (:107-126)
    one = od.rope(q[:, :, 1:2], torch.tensor([1]), 1e6)
    assert torch.equal(one, r[:, :, 1:2])                                                      # offset equivalence (:128-149)
    assert torch.equal(od.rope(q[:, :, :1], torch.tensor([0]), 1e6), q[:, :, :1])             # position 0 is identity
    qn = od.rms_norm(torch.randn(7, 4, 16) * 5, torch.ones(16), 1e-6)
    assert torch.allclose((qn ** 2).mean(-1).sqrt(), torch.ones(7, 4), atol=2e-2)             # QK-norm unit RMS (:157-173)


def test_cached_decode_equals_full_context():
    dec = _tiny_decoder()
    ids = [5, 17, 300, 42, 7, 99]
    full = dec.forward(dec.embed(ids), last_only=False)
    assert full.shape == (6, 512)                                                              # logits (T, V) (:208-226)
    dec.reset()
    dec.forward(dec.embed(ids[:-1]))
    assert dec.k[0].shape == (2, 5, 16) and dec.offset == 5                                    # cache growth (:254-283)
    step = dec.forward(dec.embed(ids[-1:]))
    assert dec.k[0].shape == (2, 6, 16)
    assert torch.allclose(step[-1], full[-1], atol=0.15, rtol=0.05)                           # bf16 re-rounding only (:285-312)
    # tied head: logits = norm(h) @ embed^T (:228-247) -- with identity-free check on shapes and finiteness
    assert torch.isfinite(full).all()


def test_fp8_activation_flavour_of_the_decoder_oracle():
    """configs[4] flavour (oracle/decoder.py, `fp8_rows`): e4m3 x e4m3 projections on the prompt rows.  Checked here:
    the weight quantiser is idempotent on dequantised weights (the GPU tests hand the oracle q * s), the activation
    quantiser reproduces the e4m3 grid and its scale rule, rows beyond `fp8_rows` keep the bf16 arithmetic, a W8A8
    prefill followed by W8A16 cached steps equals ONE teacher-forced pass with fp8_rows = prompt length, and the
    flavour really is a different arithmetic."""
    import oracle.decoder as od
    torch.manual_seed(3)
    W = (torch.randn(96, 64) * 0.02).to(torch.bfloat16).float()
    W[5] = 0.0                                                                 # an all-zero row must not divide by zero
    q, s = od.quantize_weight_rows_e4m3(W)
    assert q.abs().max() <= 448 and torch.equal(torch.log2(s), torch.log2(s).round())           # power-of-two scales
    assert torch.equal(q.to(torch.float8_e4m3fn).float(), q)                                     # values on the e4m3 grid
    w_eff = q * s[:, None]
    assert torch.equal(w_eff.to(torch.bfloat16).float(), w_eff)                                  # exactly representable in bf16
    q2, s2 = od.quantize_weight_rows_e4m3(w_eff)
    assert torch.equal(q2 * s2[:, None], w_eff)                                                  # idempotent in value
    x = torch.randn(5, 64)
    x[2] = 0.0
    x8, sa = od.quantize_act_rows_e4m3(x)
    assert sa[2] == 1.0 and torch.equal(x8[2], torch.zeros(64))
    live = [0, 1, 3, 4]
    assert torch.allclose(sa[live] * 448, x[live].abs().amax(-1), rtol=1e-6) and torch.all(x8.abs().amax(-1)[live] == 448)
    assert (x8 * sa[:, None] - x).abs().max() <= x.abs().max() / 16 + 1e-6                       # 3 mantissa bits: 2^-4 relative

    dec = _tiny_decoder()
    dec.W = {k: (od.quantize_weight_rows_e4m3(v)[0] * od.quantize_weight_rows_e4m3(v)[1][:, None]
                 if k.endswith("_proj.weight") or k == "embed_tokens.weight" else v) for k, v in dec.W.items()}
    ids = [5, 17, 300, 42, 7, 99, 3, 250]
    plain = dec.forward(dec.embed(ids), last_only=False)
    dec.reset()
    f8 = dec.forward(dec.embed(ids), last_only=False, fp8_rows=len(ids))
    assert torch.isfinite(f8).all() and not torch.equal(f8, plain)
    assert (f8 - plain).abs().max() < 0.25 * plain.abs().max()                # a perturbation, not another function
    dec.reset()
    mixed = dec.forward(dec.embed(ids), last_only=False, fp8_rows=5)          # prompt = 5 rows, 3 teacher-forced tokens
    dec.reset()
    pre = dec.forward(dec.embed(ids[:5]), last_only=False, fp8_rows=5)
    assert torch.equal(pre, mixed[:5])                                         # per-token quantisation: causal rows cannot see later ones
    for i in range(5, 8):
        step = dec.forward(dec.embed(ids[i:i + 1]))
        assert torch.allclose(step[-1], mixed[i], atol=1e-5)                  # cached W8A16 step == row i of the one-pass form
    dec.reset()
    assert torch.equal(dec.forward(dec.embed(ids), last_only=False, fp8_rows=0), plain)


def test_fp8_flavour_amplifies_perturbations():
    """Why weight_dtype="fp8" is compared with its oracle at FP8_AMPLIFICATION x the bf16 bar (tests/test_parity_gpu.py):
    perturb every projection of the decoder oracle by the same relative noise (what another fp32 summation order or exp
    implementation does) and compare how far the last-position logits move in the two flavours.  e4m3 rounding is
    discontinuous, so the fp8 flavour moves several times further; measured 3.2-4.8 x on the small configuration's 30 s
    prompt, 2.5-2.9 x on random embeddings (here), for noise from 1e-5 to 3e-3."""
    import oracle.decoder as od
    g = torch.Generator().manual_seed(0)
    dec = _tiny_decoder(seed=1, layers=4, H=256, nh=4, nkv=2, hd=64, I=512, V=2048)
    dec.W = {k: (od.quantize_weight_rows_e4m3(v)[0] * od.quantize_weight_rows_e4m3(v)[1][:, None]
                 if k.endswith("_proj.weight") or k == "embed_tokens.weight" else v) for k, v in dec.W.items()}
    dec.gpu_rounding = True
    emb = (torch.randn(64, 256, generator=g)).to(torch.bfloat16).float()
    moved = {}
    for name, n8 in (("bf16", 0), ("fp8", 64)):
        dec.reset()
        dec.perturb = 0.0
        base = dec.forward(emb, fp8_rows=n8)[-1]
        d = []
        for rep in range(4):
            dec.reset()
            dec.perturb = 1e-4
            torch.manual_seed(50 + rep)
            d.append(float((dec.forward(emb, fp8_rows=n8)[-1] - base).abs().mean()))
        moved[name] = sum(d) / len(d)
    dec.perturb = 0.0
    assert moved["bf16"] > 0 and 1.8 < moved["fp8"] / moved["bf16"] < 8.0, moved


def test_prepare_inputs_semantics():
    """reference test_generate.py:63-124."""
    E = torch.arange(20 * 4, dtype=torch.float32).reshape(20, 4)
    pad = 7
    out = oracle.prepare_inputs(torch.zeros(0, 4), [1, 2, 3, 4], E, audio_pad_id=pad)
    assert torch.equal(out, E[[1, 2, 3, 4]])
    out = oracle.prepare_inputs(torch.full((2, 4), 9.0), [1, pad, pad, 2], E, audio_pad_id=pad)
    assert torch.equal(out[1:3], torch.full((2, 4), 9.0)) and torch.equal(out[0], E[1]) and torch.equal(out[3], E[2])
    with pytest.raises(ValueError, match="does not match"):
        oracle.prepare_inputs(torch.zeros(3, 4), [1, pad, pad], E, audio_pad_id=pad)


def test_generate_loop_bounds_and_eos():
    """reference test_generate.py:201-242: <= max_tokens ids, list[int], EOS kept then stop."""
    dec = _tiny_decoder(V=512)
    out = oracle.generate(dec, torch.zeros(0, 64), [1, 2, 3], max_tokens=5)
    assert isinstance(out, list) and len(out) == 5 and all(isinstance(t, int) for t in out)
    assert oracle.generate(dec, torch.zeros(0, 64), [1, 2, 3], max_tokens=1) == out[:1]
    out2 = oracle.generate(dec, torch.zeros(0, 64), [1, 2, 3], max_tokens=5)
    assert out2 == out                                                                         # greedy determinism


def test_encoder_oracle_shapes_and_block_locality():
    """Conv stem token counts and the block-diagonal window: changing audio in one 104-token window
    must not change another window's output beyond the shared conv chunks (encoder.py:210-229)."""
    g = torch.Generator().manual_seed(1)
    d, C1, ffn, out_dim, heads = 64, 8, 128, 32, 1
    W = {"conv2d1.weight": torch.randn(C1, 3, 3, 1, generator=g) * 0.3, "conv2d1.bias": torch.zeros(C1),
         "conv2d2.weight": torch.randn(C1, 3, 3, C1, generator=g) * 0.1, "conv2d2.bias": torch.zeros(C1),
         "conv2d3.weight": torch.randn(C1, 3, 3, C1, generator=g) * 0.1, "conv2d3.bias": torch.zeros(C1),
         "conv_out.weight": torch.randn(d, C1 * 16, generator=g) * 0.1,
         "ln_post.weight": torch.ones(d), "ln_post.bias": torch.zeros(d),
         "proj1.weight": torch.randn(d, d, generator=g) * 0.1, "proj1.bias": torch.zeros(d),
         "proj2.weight": torch.randn(out_dim, d, generator=g) * 0.1, "proj2.bias": torch.zeros(out_dim)}
    p = "layers.0."
    for n, o, k in (("self_attn.q_proj", d, d), ("self_attn.k_proj", d, d), ("self_attn.v_proj", d, d),
                    ("self_attn.out_proj", d, d), ("fc1", ffn, d), ("fc2", d, ffn)):
        W[p + n + ".weight"] = torch.randn(o, k, generator=g) * 0.1
        W[p + n + ".bias"] = torch.zeros(o)
    for n in ("self_attn_layer_norm", "final_layer_norm"):
        W[p + n + ".weight"], W[p + n + ".bias"] = torch.ones(d), torch.zeros(d)
    mel = torch.randn(128, 1000, generator=g)
    out = oracle.encoder_forward(mel, W, n_heads=heads)
    assert out.shape == (130, out_dim)
    mel2 = mel.clone()
    mel2[:, 900:] += 1.0                                    # only the last chunk (window 2: tokens 104..129)
    out2 = oracle.encoder_forward(mel2, W, n_heads=heads)
    assert torch.equal(out[:104], out2[:104]) and not torch.allclose(out[117:], out2[117:])
    assert oracle.encoder_forward(mel[:, :250], W, n_heads=heads).shape[0] == 33
    pe = oracle.sinusoid_table(1500, 64)
    assert pe.shape == (1500, 64) and torch.equal(pe[:13], oracle.sinusoid_table(13, 64))     # test_encoder.py:171-198
    assert torch.equal(pe[0], torch.cat([torch.zeros(32), torch.ones(32)]))


# ---------------------------------------------------------------------------------------------------
# Second, independent statement of the same architecture: the PyTorch implementation that ships with `transformers`
# (modeling_qwen3_asr.py for the audio tower, Qwen3ForCausalLM for the text side), instantiated from a local tiny config
# with the SAME random weights -- nothing is fetched.  It pins the oracle's structure (conv flatten order, per-chunk
# positional embedding, pad-token strip, attention windows; QK-norm before RoPE, half-split RoPE, GQA, tied head,
# cached == full context); the MLX-specific bf16 rounding points stay as read from the reference.
def _small_weights():
    from qwen3_asr_mlx_amd import synth
    cfg = synth.small_config()
    st = synth.random_state(cfg, seed=0, bias_std=0.02, norm_jitter=0.1)
    We = {k[len("audio_tower."):]: v.float() for k, v in st.items() if k.startswith("audio_tower.")}
    Wd = {k[len("model."):]: v.float() for k, v in st.items() if k.startswith("model.")}
    return cfg, We, Wd


def test_encoder_oracle_matches_transformers_implementation():
    m = pytest.importorskip("transformers.models.qwen3_asr.modeling_qwen3_asr")
    c = pytest.importorskip("transformers.models.qwen3_asr.configuration_qwen3_asr")
    import oracle
    from signals import signal
    cfg, We, _ = _small_weights()
    a = cfg.audio_encoder
    ec = c.Qwen3ASREncoderConfig(num_mel_bins=a.num_mel_bins, encoder_layers=a.encoder_layers, encoder_attention_heads=a.encoder_attention_heads,
                                 encoder_ffn_dim=a.encoder_ffn_dim, d_model=a.d_model, n_window=a.n_window, output_dim=a.output_dim,
                                 n_window_infer=a.n_window_infer, downsample_hidden_size=a.downsample_hidden_size)
    ec._attn_implementation = "eager"
    enc = m.Qwen3ASREncoder(ec).eval().float()
    sd = {k: (v.permute(0, 3, 1, 2).contiguous() if k.startswith("conv2d") and k.endswith("weight") else v)     # (O,kH,kW,I) -> (O,I,kH,kW)
          for k, v in We.items() if not k.startswith("proj")}
    enc.load_state_dict(sd, strict=True)
    for n in (16000 * 3 + 777, 16000 * 11):                    # one window / several 104-token windows, ragged last chunk
        mel = oracle.log_mel_spectrogram(signal("speechlike", n, seed=3))
        F = mel.shape[1]
        Fp = (F + 99) // 100 * 100
        feat = torch.zeros(1, 128, Fp)
        feat[0, :, :F] = torch.from_numpy(mel)
        mask = torch.zeros(1, Fp, dtype=torch.long)
        mask[0, :F] = 1
        with torch.no_grad():
            h = enc(feat, mask).last_hidden_state
            theirs = torch.nn.functional.gelu(h @ We["proj1.weight"].T + We["proj1.bias"]) @ We["proj2.weight"].T + We["proj2.bias"]
        mine = oracle.encoder_forward(torch.from_numpy(mel), We, a.encoder_attention_heads, a.n_window, a.n_window_infer, gpu_rounding=False)
        assert mine.shape == theirs.shape
        torch.testing.assert_close(mine, theirs, atol=2e-6, rtol=1e-5)


def test_decoder_oracle_matches_transformers_qwen3(monkeypatch):
    tr = pytest.importorskip("transformers")
    import oracle
    import oracle.decoder as od
    cfg, _, Wd = _small_weights()
    t = cfg.text_decoder
    hc = tr.Qwen3Config(vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                        num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                        num_key_value_heads=t.num_key_value_heads, head_dim=t.head_dim, hidden_act="silu", max_position_embeddings=4096,
                        rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta, tie_word_embeddings=True, attention_bias=False,
                        use_sliding_window=False, attn_implementation="eager")
    hf = tr.Qwen3ForCausalLM(hc).eval().float()
    sd = {"model." + k: v for k, v in Wd.items()}
    sd["lm_head.weight"] = Wd["embed_tokens.weight"]
    hf.load_state_dict(sd, strict=True)
    emb = torch.randn(1, 37, t.hidden_size, generator=torch.Generator().manual_seed(0)) * 0.5
    with torch.no_grad():
        ref = hf(inputs_embeds=emb).logits[0]
    monkeypatch.setattr(od, "bf", lambda x: x)                 # fp32 at every op: structure, not rounding
    dec = oracle.DecoderOracle(Wd, t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.rms_norm_eps, t.rope_theta)
    full = dec.forward(emb[0].clone(), last_only=False)
    torch.testing.assert_close(full, ref, atol=2e-5, rtol=1e-5)
    dec.reset()                                                # prefill 30 tokens, then 7 cached single steps
    inc = torch.cat([dec.forward(emb[0, :30].clone(), last_only=False)] + [dec.forward(emb[0, i:i + 1].clone(), last_only=False) for i in range(30, 37)])
    torch.testing.assert_close(inc, ref, atol=2e-5, rtol=1e-5)


def test_whole_model_oracle_matches_transformers_conditional_generation(monkeypatch):
    """mel -> encoder -> prompt -> audio-pad merge -> decoder logits, oracle vs transformers' Qwen3ASRForConditionalGeneration
    on the same random weights: pins prepare_inputs / build_prompt placement and the composition of the stages."""
    m = pytest.importorskip("transformers.models.qwen3_asr.modeling_qwen3_asr")
    c = pytest.importorskip("transformers.models.qwen3_asr.configuration_qwen3_asr")
    import oracle
    import oracle.decoder as od
    import importlib
    og = importlib.import_module("oracle.generate")
    from qwen3_asr_mlx_amd.tokenizer import AUDIO_PAD_TOKEN_ID, build_prompt
    from signals import signal
    cfg, We, Wd = _small_weights()
    a, t = cfg.audio_encoder, cfg.text_decoder
    hc = c.Qwen3ASRConfig(
        audio_config=dict(num_mel_bins=a.num_mel_bins, encoder_layers=a.encoder_layers, encoder_attention_heads=a.encoder_attention_heads,
                          encoder_ffn_dim=a.encoder_ffn_dim, d_model=a.d_model, n_window=a.n_window, output_dim=a.output_dim,
                          n_window_infer=a.n_window_infer, downsample_hidden_size=a.downsample_hidden_size),
        text_config=dict(model_type="qwen3", vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                         num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, head_dim=t.head_dim, hidden_act="silu", max_position_embeddings=4096,
                         rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta, tie_word_embeddings=True, attention_bias=False,
                         use_sliding_window=False),
        audio_token_id=AUDIO_PAD_TOKEN_ID)
    hc._attn_implementation = "eager"
    hf = m.Qwen3ASRForConditionalGeneration(hc).eval().float()
    sd = {}
    for k, v in We.items():
        if k.startswith("proj"):
            sd["model.multi_modal_projector.linear_" + k[4:]] = v
        else:
            sd["model.audio_tower." + k] = v.permute(0, 3, 1, 2).contiguous() if k.startswith("conv2d") and k.endswith("weight") else v
    sd.update({"model.language_model." + k: v for k, v in Wd.items()})
    sd["lm_head.weight"] = Wd["embed_tokens.weight"]
    hf.load_state_dict(sd, strict=True)

    monkeypatch.setattr(od, "bf", lambda x: x)                 # fp32 at every op: structure, not rounding
    monkeypatch.setattr(og, "bf", lambda x: x)
    mel = oracle.log_mel_spectrogram(signal("speechlike", 16000 * 5 + 321, seed=5))
    F = mel.shape[1]
    Fp = (F + 99) // 100 * 100
    feat = torch.zeros(1, 128, Fp)
    feat[0, :, :F] = torch.from_numpy(mel)
    mask = torch.zeros(1, Fp, dtype=torch.long)
    mask[0, :F] = 1
    enc = oracle.encoder_forward(torch.from_numpy(mel), We, a.encoder_attention_heads, a.n_window, a.n_window_infer, gpu_rounding=False)
    ids = build_prompt(enc.shape[0], [6364])                   # " English"
    with torch.no_grad():
        ref = hf(input_ids=torch.tensor([ids]), input_features=feat, input_features_mask=mask).logits[0]
    emb = oracle.prepare_inputs(enc, ids, Wd["embed_tokens.weight"])
    dec = oracle.DecoderOracle(Wd, t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.rms_norm_eps, t.rope_theta)
    mine = dec.forward(emb, last_only=False)
    assert mine.shape == ref.shape
    torch.testing.assert_close(mine, ref, atol=3e-5, rtol=1e-5)
    assert int(mine[-1].argmax()) == int(ref[-1].argmax())
    with pytest.raises(ValueError):                            # both refuse a pad count that differs from the encoder length
        oracle.prepare_inputs(enc[:-1], ids, Wd["embed_tokens.weight"])
    with pytest.raises(ValueError), torch.no_grad():
        hf(input_ids=torch.tensor([build_prompt(enc.shape[0] + 1, [6364])]), input_features=feat, input_features_mask=mask)
