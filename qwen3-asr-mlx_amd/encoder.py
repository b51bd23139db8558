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
        self.device = None
        self._plans: dict = {}

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
        for i in (2, 3):
            w[f"c{i}_w"] = ops.conv_weight_chunked(bf(W[f"conv2d{i}.weight"].to(dev).permute(0, 2, 1, 3)))   # (O, kt, kf, I) -> chunk-major K
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
        self.w, self.device = w, dev

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
