"""Oracle: Qwen3 text decoder with KV cache (torch-CPU) -- test infrastructure.

Restates reference decoder.py:20-253 under MLX bf16 semantics: every op's
output is rounded to bf16 (the checkpoint and the embeddings are bf16,
generate.py:48), reductions/softmax in float32.  Weights keyed as in the
checkpoint after stripping ``model.`` (decoder.py:282-288).
PARITY UNPINNED at real weights (see oracle/__init__.py).
"""
from __future__ import annotations

import torch


def bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def rms_norm(x, w, eps):
    """MLX nn.RMSNorm: x * rsqrt(mean(x^2) + eps) * w, float32 inside (decoder.py:124,177)."""
    xf = x.to(torch.float32)
    return bf(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w)


def rope(x, pos, theta):
    """MLX nn.RoPE(dims, traditional=False, base): pairs (i, i+D/2),
    angle = pos * base**(-2i/D) (decoder.py:126,151-153).  x (..., T, D); pos (T,)."""
    d = x.shape[-1]
    half = d // 2
    inv = theta ** (-torch.arange(half, dtype=torch.float32) / half)
    ang = pos.to(torch.float32)[:, None] * inv[None, :]
    c, s = torch.cos(ang), torch.sin(ang)
    x1, x2 = x[..., :half], x[..., half:]
    return bf(torch.cat([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1))


class DecoderOracle:
    """One sequence (the reference is batch-1 everywhere: generate.py:48,294)."""

    def __init__(self, W: dict, n_heads=16, n_kv_heads=8, head_dim=128, eps=1e-6,
                 theta=1_000_000.0, gpu_rounding=False):
        self.W = W
        self.nh, self.nkv, self.hd, self.eps, self.theta = n_heads, n_kv_heads, head_dim, eps, theta
        self.n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("layers."))
        self.gpu_rounding = gpu_rounding
        self.reset()

    def reset(self):
        self.k = [None] * self.n_layers          # decoder.py:20-64 (concat-grown cache)
        self.v = [None] * self.n_layers
        self.offset = 0

    def embed(self, ids) -> torch.Tensor:
        return self.W["embed_tokens.weight"][torch.as_tensor(ids, dtype=torch.long)]

    def forward(self, h: torch.Tensor, last_only: bool = True, keep_from: int = None) -> torch.Tensor:
        """h (T, hidden) bf16-valued float32; appends to the cache at
        self.offset and advances it.  Returns bf16-valued logits (T or 1, V)
        (decoder.py:243-253; the reference computes every position, only the
        last is consumed at generate.py:279)."""
        W = self.W
        T = h.shape[0]
        pos = torch.arange(self.offset, self.offset + T)
        g = self.nh // self.nkv
        for li in range(self.n_layers):
            p = f"layers.{li}."
            x = rms_norm(h, W[p + "input_layernorm.weight"], self.eps)
            q = bf(x @ W[p + "self_attn.q_proj.weight"].T).reshape(T, self.nh, self.hd)
            k = bf(x @ W[p + "self_attn.k_proj.weight"].T).reshape(T, self.nkv, self.hd)
            v = bf(x @ W[p + "self_attn.v_proj.weight"].T).reshape(T, self.nkv, self.hd)
            q = rms_norm(q, W[p + "self_attn.q_norm.weight"], self.eps).transpose(0, 1)   # (H,T,D)
            k = rms_norm(k, W[p + "self_attn.k_norm.weight"], self.eps).transpose(0, 1)
            v = v.transpose(0, 1)
            q = rope(q, pos, self.theta)
            k = rope(k, pos, self.theta)
            self.k[li] = k if self.k[li] is None else torch.cat([self.k[li], k], dim=1)
            self.v[li] = v if self.v[li] is None else torch.cat([self.v[li], v], dim=1)
            K, V = self.k[li], self.v[li]
            L = K.shape[1]
            # causal: query at absolute position offset+i sees keys j <= offset+i (decoder.py:71-80)
            allowed = torch.arange(L)[None, :] <= pos[:, None]
            Kq = K.repeat_interleave(g, dim=0)          # GQA: q head h -> kv head h//g
            Vq = V.repeat_interleave(g, dim=0)
            sc = (q @ Kq.transpose(1, 2)) * (self.hd ** -0.5)
            sc = sc.masked_fill(~allowed[None], float("-inf"))
            m = sc.max(dim=-1, keepdim=True).values
            pr = torch.exp(sc - m)
            l = pr.sum(-1, keepdim=True)
            if self.gpu_rounding:
                pr = bf(pr)
            a = bf((pr @ Vq) / l).transpose(0, 1).reshape(T, self.nh * self.hd)
            h = bf(h + bf(a @ W[p + "self_attn.o_proj.weight"].T))
            x = rms_norm(h, W[p + "post_attention_layernorm.weight"], self.eps)
            gt = bf(x @ W[p + "mlp.gate_proj.weight"].T)
            up = bf(x @ W[p + "mlp.up_proj.weight"].T)
            act = bf(gt * torch.sigmoid(gt) * up)
            h = bf(h + bf(act @ W[p + "mlp.down_proj.weight"].T))
        self.offset += T
        if keep_from is not None:               # teacher-forced checks: logits of rows keep_from.. only
            h = h[keep_from:]
        elif last_only:
            h = h[-1:]
        x = rms_norm(h, W["norm.weight"], self.eps)
        return bf(x @ W["embed_tokens.weight"].T)
