"""Audio encoder on HIP kernels (mirror of reference encoder.py; same class / loader names).

Batched over utterances: every utterance's 100-frame chunks go through the conv stem as one
batch of images, the valid tokens of all utterances are packed into one [N_total, d_model]
matrix, and attention runs over independent 104-token windows (never across utterances).
Mixed precision: MFMA operands bf16, accumulation / residual stream / LayerNorm in fp32.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Optional, Sequence

from . import ops
from .config import AudioEncoderConfig


class SinusoidalPositionEmbedding:
    """[sin | cos] table, inv_ts = exp(-i*ln(1e4)/(half-1)) (reference encoder.py:21-44)."""

    def __init__(self, max_positions: int, d_model: int) -> None:
        import torch
        half = d_model // 2
        inc = math.log(10000.0) / (half - 1)
        inv = torch.exp(-torch.arange(half, dtype=torch.float32) * inc)
        ang = torch.arange(max_positions, dtype=torch.float32)[:, None] * inv[None, :]
        self._positional_embedding = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)

    def __call__(self, seqlen: int):
        return self._positional_embedding[:seqlen, :]


class AudioEncoder:
    def __init__(self, config: AudioEncoderConfig) -> None:
        self.config = config
        self.chunk_size = config.n_window * 2
        if self.chunk_size != 100 or config.num_mel_bins != 128:
            raise NotImplementedError("HIP conv stem is specialised for 100-frame chunks of 128 mel bins")
        if config.d_model % config.encoder_attention_heads or config.d_model // config.encoder_attention_heads != 64:
            raise NotImplementedError("HIP encoder attention is specialised for head_dim 64")
        self.positional_embedding = SinusoidalPositionEmbedding(config.max_source_positions, config.d_model)
        self.tokens_per_chunk = self._conv_output_length(self.chunk_size)
        self.window_tokens = self.tokens_per_chunk * (config.n_window_infer // self.chunk_size)
        self.w = None
        self.w3 = None               # parity mode: [W | W | W] copies of every GEMM / conv weight, built on first use
        self.device = None
        self._plans: dict = {}
        # "bf16": MFMA operands rounded to bf16 (fast path).  "fp32": the reference's semantics -- float32 activations end to
        # end (audio.py:278, encoder.py:272-323 never cast), each GEMM fp32-accurate through the three-way bf16 split of
        # csrc/parity.hip at three times the MFMA work.  For parity measurements, not throughput.
        self.precision = "bf16"

    # ------------------------------------------------------------------ helpers (reference encoder.py:197-229)
    @staticmethod
    def _conv_output_length(input_length: int) -> int:
        n = input_length
        for _ in range(3):
            n = (n - 1) // 2 + 1
        return n

    @staticmethod
    def _window_cu_seqlens(total_tokens: int, window_tokens: int) -> list[int]:
        cu = [0]
        for _ in range(total_tokens // window_tokens):
            cu.append(cu[-1] + window_tokens)
        if total_tokens % window_tokens:
            cu.append(cu[-1] + total_tokens % window_tokens)
        return cu

    @staticmethod
    def _block_attention_mask(seq_len: int, cu_seqlens: list[int]):
        """Dense additive mask of the reference (encoder.py:210-229); the HIP path never builds it --
        kept for API parity and tests."""
        import torch
        if len(cu_seqlens) <= 2:
            return None
        mask = torch.full((seq_len, seq_len), -1e9)
        for s, e in zip(cu_seqlens[:-1], cu_seqlens[1:]):
            mask[s:e, s:e] = 0.0
        return mask[None, None]

    # ------------------------------------------------------------------ weights
    def load_state(self, W: dict, device) -> None:
        """W: checkpoint tensors with the ``audio_tower.`` prefix stripped, MLX layouts
        (Linear (out,in); Conv2d (O,kH,kW,I)).  Strict key set, as MLX load_weights."""
        import torch
        cfg = self.config
        dev = torch.device(device)
        need = {"conv2d1.weight", "conv2d1.bias", "conv2d2.weight", "conv2d2.bias", "conv2d3.weight", "conv2d3.bias",
                "conv_out.weight", "ln_post.weight", "ln_post.bias", "proj1.weight", "proj1.bias", "proj2.weight", "proj2.bias"}
        for i in range(cfg.encoder_layers):
            p = f"layers.{i}."
            for n in ("self_attn_layer_norm", "final_layer_norm", "self_attn.q_proj", "self_attn.k_proj",
                      "self_attn.v_proj", "self_attn.out_proj", "fc1", "fc2"):
                need.add(p + n + ".weight")
                need.add(p + n + ".bias")
        missing, extra = need - set(W), set(W) - need
        if missing or extra:
            raise ValueError(f"encoder weights: missing {sorted(missing)[:4]} extra {sorted(extra)[:4]}")

        def bf(t):
            return t.to(device=dev, dtype=torch.bfloat16).contiguous()

        def f32(t):
            return t.to(device=dev, dtype=torch.float32).contiguous()

        C1, d = cfg.downsample_hidden_size, cfg.d_model
        w = {}
        # conv weights: (O, kH=freq, kW=time, I) -> tap order (time, freq) for the time-major activations
        w["c1_w"] = f32(W["conv2d1.weight"].to(dev)[..., 0].permute(0, 2, 1))                   # [C1][kt][kf]
        w["c1_b"] = f32(W["conv2d1.bias"])
        self._conv_w = {}
        for i in (2, 3):
            self._conv_w[i] = bf(W[f"conv2d{i}.weight"].to(dev).permute(0, 2, 1, 3))                       # (O, kt, kf, I)
            w[f"c{i}_w"] = ops.conv_weight_chunked(self._conv_w[i])                                         # chunk-major K
            w[f"c{i}_b"] = f32(W[f"conv2d{i}.bias"])
        fq = W["conv_out.weight"].shape[1] // C1
        # conv_out input index is ch*freq + f (encoder.py:278); our rows are [f][ch]
        w["co_w"] = bf(W["conv_out.weight"].to(dev).reshape(d, C1, fq).permute(0, 2, 1)).reshape(d, fq * C1)
        w["pe"] = f32(self.positional_embedding(self.tokens_per_chunk))
        layers = []
        for i in range(cfg.encoder_layers):
            p = f"layers.{i}."
            a = p + "self_attn."
            layers.append({
                "ln1_w": f32(W[p + "self_attn_layer_norm.weight"]), "ln1_b": f32(W[p + "self_attn_layer_norm.bias"]),
                "wqkv": bf(torch.cat([W[a + "q_proj.weight"], W[a + "k_proj.weight"], W[a + "v_proj.weight"]], 0)),
                "bqkv": f32(torch.cat([W[a + "q_proj.bias"], W[a + "k_proj.bias"], W[a + "v_proj.bias"]], 0)),
                "wo": bf(W[a + "out_proj.weight"]), "bo": f32(W[a + "out_proj.bias"]),
                "ln2_w": f32(W[p + "final_layer_norm.weight"]), "ln2_b": f32(W[p + "final_layer_norm.bias"]),
                "w1": bf(W[p + "fc1.weight"]), "b1": f32(W[p + "fc1.bias"]),
                "w2": bf(W[p + "fc2.weight"]), "b2": f32(W[p + "fc2.bias"]),
            })
        w["layers"] = layers
        w["lnp_w"], w["lnp_b"] = f32(W["ln_post.weight"]), f32(W["ln_post.bias"])
        w["p1_w"], w["p1_b"] = bf(W["proj1.weight"]), f32(W["proj1.bias"])
        w["p2_w"], w["p2_b"] = bf(W["proj2.weight"]), f32(W["proj2.bias"])
        self.w, self.device, self.w3 = w, dev, None

    # ------------------------------------------------------------------ batch plan (host ints -> device index arrays)
    def _plan(self, n_frames: tuple):
        # one cache per lane: a plan's index tensors are freed by eviction, and memory freed on one lane's stream must not be
        # handed out while another lane's launches still read it
        from .decoder import current_lane
        plans = self._plans.setdefault(current_lane(), {})
        plan = plans.get(n_frames)
        if plan is not None:
            return plan
        import torch
        tpc, cs = self.tokens_per_chunk, self.chunk_size
        chunk_b, chunk_f0, row_map, tok_counts = [], [], [], []
        seg_start, seg_len = [], []
        tok = 0
        for b, nf in enumerate(n_frames):
            first = tok
            for f0 in range(0, nf, cs):
                valid = self._conv_output_length(min(cs, nf - f0))          # encoder.py:289-291
                chunk_b.append(b)
                chunk_f0.append(f0)
                row_map += [tok + t if t < valid else -1 for t in range(tpc)]
                tok += valid
            n_b = tok - first
            tok_counts.append(n_b)
            cu = self._window_cu_seqlens(n_b, self.window_tokens)           # encoder.py:297-309
            for s, e in zip(cu[:-1], cu[1:]):
                seg_start.append(first + s)
                seg_len.append(e - s)
        dev = self.device
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
        plan = {
            "n_chunks": len(chunk_b), "n_tokens": tok, "tok_counts": tok_counts,
            "chunk_b": i32(chunk_b), "chunk_f0": i32(chunk_f0), "row_map": i32(row_map),
            "seg_start": i32(seg_start), "seg_len": i32(seg_len), "n_segs": len(seg_len), "max_seg": max(seg_len),
            "kv_off": torch.tensor([s * 3 * self.config.d_model for s in seg_start], dtype=torch.int64, device=dev),
        }
        if len(plans) > 64:
            plans.clear()
        plans[n_frames] = plan
        return plan

    # ------------------------------------------------------------------ forward
    def forward_batch(self, mel, n_frames: Sequence[int]):
        """mel fp32 [B][128][Fmax] on the device (zero beyond each utterance's n_frames).
        Returns (features fp32 [N_total][output_dim], tokens per utterance)."""
        import torch
        if self.w is None:
            raise TypeError("AudioEncoder has no weights loaded")
        if self.precision == "fp32":
            return self._forward_batch_fp32(mel, n_frames)
        if self.precision != "bf16":
            raise ValueError(f"AudioEncoder.precision must be 'bf16' or 'fp32', got {self.precision!r}")
        cfg, w, dev = self.config, self.w, self.device
        pl = self._plan(tuple(int(n) for n in n_frames))
        Ct, N, d, C1 = pl["n_chunks"], pl["n_tokens"], cfg.d_model, cfg.downsample_hidden_size
        nh = cfg.encoder_attention_heads
        bf16 = torch.bfloat16
        act1 = torch.empty((Ct, 50, 64, C1), dtype=bf16, device=dev)
        ops.call("q3asr_conv1_gelu", ops.ptr(mel), mel.shape[2], ops.ptr(pl["chunk_b"]), ops.ptr(pl["chunk_f0"]), Ct,
                 ops.ptr(w["c1_w"]), ops.ptr(w["c1_b"]), ops.ptr(act1), C1, ops.sp())
        act2 = torch.empty((Ct, 25, 32, C1), dtype=bf16, device=dev)
        ops.conv3x3s2(act1, Ct, 50, 64, C1, w["c2_w"], C1, w["c2_b"], 1, act2)
        del act1
        act3 = torch.empty((Ct, 13, 16, C1), dtype=bf16, device=dev)
        ops.conv3x3s2(act2, Ct, 25, 32, C1, w["c3_w"], C1, w["c3_b"], 1, act3)
        del act2
        hidden = torch.empty((N, d), dtype=torch.float32, device=dev)
        # conv_out (no bias) + per-chunk positional embedding + pad-token strip (encoder.py:279-293)
        ops.gemm(act3, w["co_w"], Ct * self.tokens_per_chunk, d, 16 * C1, hidden, res=w["pe"], res_kind=1, ldres=d,
                 res_mod=self.tokens_per_chunk, out_fp32=True, row_map=pl["row_map"])
        del act3
        xn = torch.empty((N, d), dtype=bf16, device=dev)
        qkv = torch.empty((N, 3 * d), dtype=bf16, device=dev)
        att = torch.empty((N, d), dtype=bf16, device=dev)
        ff = torch.empty((N, cfg.encoder_ffn_dim), dtype=bf16, device=dev)
        scale = 64 ** -0.5
        for L in w["layers"]:
            ops.layernorm(hidden, L["ln1_w"], L["ln1_b"], xn, N, d, 1e-5)
            ops.gemm(xn, L["wqkv"], N, 3 * d, d, qkv, bias=L["bqkv"])
            ops.attn_fwd(ops.ptr(qkv), 3 * d, ops.off_ptr(qkv, d), ops.off_ptr(qkv, 2 * d), pl["kv_off"], 64, 3 * d,
                         att, d, pl["seg_start"], pl["seg_len"], pl["n_segs"], pl["max_seg"], nh, nh, 64, False, scale)
            ops.gemm(att, L["wo"], N, d, d, hidden, bias=L["bo"], res=hidden, res_kind=1, ldres=d, out_fp32=True)
            ops.layernorm(hidden, L["ln2_w"], L["ln2_b"], xn, N, d, 1e-5)
            ops.gemm(xn, L["w1"], N, cfg.encoder_ffn_dim, d, ff, bias=L["b1"], act=1)
            ops.gemm(ff, L["w2"], N, d, cfg.encoder_ffn_dim, hidden, bias=L["b2"], res=hidden, res_kind=1, ldres=d, out_fp32=True)
        ops.layernorm(hidden, w["lnp_w"], w["lnp_b"], xn, N, d, 1e-5)
        ops.gemm(xn, w["p1_w"], N, d, d, att, bias=w["p1_b"], act=1)
        out = torch.empty((N, cfg.output_dim), dtype=torch.float32, device=dev)
        ops.gemm(att, w["p2_w"], N, cfg.output_dim, d, out, bias=w["p2_b"], out_fp32=True)
        return out, pl["tok_counts"]

    # ------------------------------------------------------------------ reference-semantics forward (fp32 activations)
    def _tripled(self) -> dict:
        """[W | W | W] along K for every GEMM weight (and the convs' channel axis): the partner of a split activation."""
        if self.w3 is None:
            import torch
            w, C1 = self.w, self.config.downsample_hidden_size
            t3 = lambda m: torch.cat([m, m, m], dim=1).contiguous()              # noqa: E731
            w3 = {"layers": []}
            for i in (2, 3):
                k4 = self._conv_w[i]                                               # (O, kt, kf, I) bf16
                w3[f"c{i}_w"] = ops.conv_weight_chunked(torch.cat([k4, k4, k4], dim=-1))
            fq = w["co_w"].shape[1] // C1
            co = w["co_w"].reshape(-1, fq, C1)
            w3["co_w"] = torch.cat([co, co, co], dim=2).reshape(co.shape[0], fq * 3 * C1).contiguous()
            for L in w["layers"]:
                w3["layers"].append({k: t3(L[k]) for k in ("wqkv", "wo", "w1", "w2")})
            w3["p1_w"], w3["p2_w"] = t3(w["p1_w"]), t3(w["p2_w"])
            self.w3 = w3
        return self.w3

    def _forward_batch_fp32(self, mel, n_frames: Sequence[int]):
        """forward_batch with float32 activations end to end (the reference's arithmetic, encoder.py:249-323): every GEMM
        operand is the three-way bf16 split of an fp32 tensor, attention / LayerNorm / GELU / residuals are fp32."""
        import torch
        cfg, w, w3, dev = self.config, self.w, self._tripled(), self.device
        pl = self._plan(tuple(int(n) for n in n_frames))
        Ct, N, d, C1 = pl["n_chunks"], pl["n_tokens"], cfg.d_model, cfg.downsample_hidden_size
        nh, ffn = cfg.encoder_attention_heads, cfg.encoder_ffn_dim
        f32, bf16 = torch.float32, torch.bfloat16
        buf = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)         # noqa: E731

        def split(x, rows, cols):
            y = buf((rows, 3 * cols), bf16)
            ops.split3(x, rows, cols, y)
            return y
        a1 = buf((Ct, 50, 64, C1), f32)
        ops.call("q3asr_conv1_gelu_f32", ops.ptr(mel), mel.shape[2], ops.ptr(pl["chunk_b"]), ops.ptr(pl["chunk_f0"]), Ct,
                 ops.ptr(w["c1_w"]), ops.ptr(w["c1_b"]), ops.ptr(a1), C1, ops.sp())
        s1 = split(a1, Ct * 50 * 64, C1)
        del a1
        a2 = buf((Ct, 25, 32, C1), f32)
        ops.conv3x3s2(s1, Ct, 50, 64, 3 * C1, w3["c2_w"], C1, w["c2_b"], 1, a2, out_fp32=True)
        del s1
        s2 = split(a2, Ct * 25 * 32, C1)
        del a2
        a3 = buf((Ct, 13, 16, C1), f32)
        ops.conv3x3s2(s2, Ct, 25, 32, 3 * C1, w3["c3_w"], C1, w["c3_b"], 1, a3, out_fp32=True)
        del s2
        s3 = split(a3, Ct * 13 * 16, C1)                                         # token row = 16 x [hi | mid | lo] of 480 channels
        del a3
        hidden = buf((N, d), f32)
        ops.gemm(s3, w3["co_w"], Ct * self.tokens_per_chunk, d, 16 * 3 * C1, hidden, res=w["pe"], res_kind=1, ldres=d,
                 res_mod=self.tokens_per_chunk, out_fp32=True, row_map=pl["row_map"])
        del s3
        xn3 = buf((N, 3 * d), bf16)
        qkv = buf((N, 3 * d), f32)
        att = buf((N, d), f32)
        ff = buf((N, ffn), f32)
        scale = 64 ** -0.5

        def ln(wgt, bias):
            ops.call("q3asr_layernorm_f32_split3", ops.ptr(hidden), ops.ptr(wgt), ops.ptr(bias), ops.ptr(xn3), N, d, 1e-5, ops.sp())
        for L, L3 in zip(w["layers"], w3["layers"]):
            ln(L["ln1_w"], L["ln1_b"])
            ops.gemm(xn3, L3["wqkv"], N, 3 * d, 3 * d, qkv, bias=L["bqkv"], out_fp32=True)
            ops.call("q3asr_attn_fwd_f32", ops.ptr(qkv), 3 * d, d, ops.ptr(pl["seg_start"]), ops.ptr(pl["seg_len"]), pl["n_segs"],
                     pl["max_seg"], nh, scale, ops.ptr(att), d, ops.sp())
            ops.gemm(split(att, N, d), L3["wo"], N, d, 3 * d, hidden, bias=L["bo"], res=hidden, res_kind=1, ldres=d, out_fp32=True)
            ln(L["ln2_w"], L["ln2_b"])
            ops.gemm(xn3, L3["w1"], N, ffn, 3 * d, ff, bias=L["b1"], act=1, out_fp32=True)
            ops.gemm(split(ff, N, ffn), L3["w2"], N, d, 3 * ffn, hidden, bias=L["b2"], res=hidden, res_kind=1, ldres=d, out_fp32=True)
        ln(w["lnp_w"], w["lnp_b"])
        ops.gemm(xn3, w3["p1_w"], N, d, 3 * d, att, bias=w["p1_b"], act=1, out_fp32=True)
        out = buf((N, cfg.output_dim), f32)
        ops.gemm(split(att, N, d), w3["p2_w"], N, cfg.output_dim, 3 * d, out, bias=w["p2_b"], out_fp32=True)
        return out, pl["tok_counts"]

    def __call__(self, mel):
        """(n_mels, T) or (1, n_mels, T) -> (1, n_tokens, output_dim) (reference encoder.py:235-323)."""
        import torch
        if not isinstance(mel, torch.Tensor):
            mel = torch.as_tensor(mel)
        if mel.ndim == 3:
            mel = mel[0]
        mel = mel.to(device=self.device, dtype=torch.float32).contiguous()[None]
        out, _ = self.forward_batch(mel, [mel.shape[2]])
        return out[None]


def _read_safetensors(model_path: str | Path, prefix: str) -> dict:
    path = Path(model_path)
    if not path.is_dir():                                    # by-name download (encoder.py:342-344); needs network
        from huggingface_hub import snapshot_download
        path = Path(snapshot_download(repo_id=str(model_path)))
    from safetensors import safe_open
    out = {}
    with safe_open(str(path / "model.safetensors"), framework="pt", device="cpu") as f:
        for key in f.keys():
            if key.startswith(prefix):
                out[key[len(prefix):]] = f.get_tensor(key)
    return out


def load_encoder_weights(model: AudioEncoder, model_path: str | Path, device: Optional[str] = None) -> None:
    """Load ``audio_tower.*`` from model.safetensors (reference encoder.py:330-359)."""
    import torch
    dev = device or f"cuda:{torch.cuda.current_device()}"
    model.load_state(_read_safetensors(model_path, "audio_tower."), dev)
