"""Oracle: audio encoder (torch-CPU float32) -- test infrastructure.

Restates reference encoder.py:21-323.  Weights are passed as a dict keyed as in
the checkpoint after stripping ``audio_tower.`` (encoder.py:349-356), in the
MLX layouts: Linear (out, in); Conv2d (O, kH, kW, I).

``gpu_rounding=False`` is the reference semantics (float32 activations against
bf16-valued weights, MLX promotion f32 (+) bf16 -> f32).  ``gpu_rounding=True``
additionally rounds activations to bf16 exactly where the HIP path does (every
MFMA operand), which lets the GPU parity tests use a tight tolerance.
PARITY UNPINNED at real weights (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .host import chunk_layout, window_cu_seqlens


def _r(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def sinusoid_table(n_pos: int, d_model: int) -> torch.Tensor:
    """encoder.py:29-40: inv_ts = exp(-i*ln(1e4)/(half-1)); [sin | cos]."""
    half = d_model // 2
    inc = math.log(10000.0) / (half - 1)
    inv = torch.exp(-torch.arange(half, dtype=torch.float32) * inc)
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


def _conv(x_nhwc: torch.Tensor, w_ohwi: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """MLX nn.Conv2d NHWC, weight (O,kH,kW,I), k3 s2 p1 (encoder.py:148-168)."""
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), w_ohwi.permute(0, 3, 1, 2), b, stride=2, padding=1)
    return y.permute(0, 2, 3, 1)


def _ln(x, w, b, eps=1e-5):
    """MLX nn.LayerNorm default eps 1e-5 (encoder.py:98,102,189)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _attention(q, k, v, cu, n_heads, rp):
    """encoder.py:78-85 with the block-diagonal mask of encoder.py:210-229:
    each window [cu[i], cu[i+1]) attends only within itself (masked entries
    carry -1e9 -> exactly 0 after softmax in float32)."""
    n, d = q.shape
    hd = d // n_heads
    out = torch.empty_like(q)
    for i in range(len(cu) - 1):
        s, e = cu[i], cu[i + 1]
        qh = q[s:e].reshape(e - s, n_heads, hd).transpose(0, 1)
        kh = k[s:e].reshape(e - s, n_heads, hd).transpose(0, 1)
        vh = v[s:e].reshape(e - s, n_heads, hd).transpose(0, 1)
        sc = (qh @ kh.transpose(1, 2)) * (hd ** -0.5)
        m = sc.max(dim=-1, keepdim=True).values
        p = torch.exp(sc - m)
        l = p.sum(dim=-1, keepdim=True)
        o = (_r(p, rp) @ vh) / l
        out[s:e] = o.transpose(0, 1).reshape(e - s, d)
    return out


def encoder_forward(mel: torch.Tensor, W: dict, n_heads: int = 16, n_window: int = 50,
                    n_window_infer: int = 800, gpu_rounding: bool = False,
                    return_hidden: bool = False) -> torch.Tensor:
    """mel float32 (n_mels, T) -> float32 (n_tokens, output_dim).  encoder.py:235-323."""
    g = gpu_rounding
    mel = mel.to(torch.float32)
    n_mels, T = mel.shape
    chunk = n_window * 2
    real, valid = chunk_layout(T, chunk)
    C = len(real)
    padded = torch.zeros(n_mels, C * chunk, dtype=torch.float32)
    padded[:, :T] = mel
    x = padded.reshape(n_mels, C, chunk).permute(1, 0, 2)[..., None]        # (C, mel, time, 1)
    x = _r(F.gelu(_conv(x, W["conv2d1.weight"], W["conv2d1.bias"])), g)
    x = _r(F.gelu(_conv(x, W["conv2d2.weight"], W["conv2d2.bias"])), g)
    x = _r(F.gelu(_conv(x, W["conv2d3.weight"], W["conv2d3.bias"])), g)
    B, fq, tm, ch = x.shape
    x = x.permute(0, 2, 3, 1).reshape(B, tm, ch * fq)                        # encoder.py:278
    x = x @ W["conv_out.weight"].T                                           # no bias
    pe = sinusoid_table(tm, x.shape[-1])
    x = x + pe[None]                                                         # encoder.py:284-286
    hidden = torch.cat([x[i, : valid[i]] for i in range(C)], dim=0)          # encoder.py:289-293
    n_tok = hidden.shape[0]
    cu = window_cu_seqlens(n_tok, tm * (n_window_infer // chunk))
    n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("layers."))
    for li in range(n_layers):
        p = f"layers.{li}."
        y = _r(_ln(hidden, W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"]), g)
        q = _r(y @ W[p + "self_attn.q_proj.weight"].T + W[p + "self_attn.q_proj.bias"], g)
        k = _r(y @ W[p + "self_attn.k_proj.weight"].T + W[p + "self_attn.k_proj.bias"], g)
        v = _r(y @ W[p + "self_attn.v_proj.weight"].T + W[p + "self_attn.v_proj.bias"], g)
        a = _r(_attention(q, k, v, cu, n_heads, g), g)
        hidden = hidden + (a @ W[p + "self_attn.out_proj.weight"].T + W[p + "self_attn.out_proj.bias"])
        y = _r(_ln(hidden, W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"]), g)
        y = _r(F.gelu(y @ W[p + "fc1.weight"].T + W[p + "fc1.bias"]), g)
        hidden = hidden + (y @ W[p + "fc2.weight"].T + W[p + "fc2.bias"])
    if return_hidden:
        return hidden
    y = _r(_ln(hidden, W["ln_post.weight"], W["ln_post.bias"]), g)
    y = _r(F.gelu(y @ W["proj1.weight"].T + W["proj1.bias"]), g)
    return y @ W["proj2.weight"].T + W["proj2.bias"]
