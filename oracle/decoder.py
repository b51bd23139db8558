"""Oracle: Qwen3 text decoder with KV cache (torch-CPU) -- test infrastructure.

Restates reference decoder.py:20-253 under MLX bf16 semantics: every op's
output is rounded to bf16 (the checkpoint and the embeddings are bf16,
generate.py:48), reductions/softmax in float32.  Weights keyed as in the
checkpoint after stripping ``model.`` (decoder.py:282-288).
PARITY UNPINNED at real weights (see oracle/__init__.py).

``fp8_rows`` (BASELINE.json configs[4], "fp8 weights (CDNA4 fp8 MFMA)"): the
first ``fp8_rows`` rows of a ``forward`` call -- the prompt rows a W8A8 prefill
handles -- run every projection of decoder.py:92-97,136-165 in e4m3 x e4m3
arithmetic: the weight quantised per output row with a power-of-two scale, the
activation operand (RMSNorm output, attention output, SwiGLU output) per token
with scale = amax / 448, products accumulated in float32 and scaled by
a_scale[m] * w_scale[n] before the usual bf16 rounding.  Rows beyond
``fp8_rows`` (the generated tokens of a teacher-forced pass) keep bf16
activations against the dequantised weights, as a W8A16 decode step does.
"""
from __future__ import annotations

import torch


def bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


E4M3_MAX = 448.0


def quantize_weight_rows_e4m3(W: torch.Tensor):
    """Per-output-row power-of-two scale s with max|W[n]| / s <= 448; returns (q, s) with q the e4m3 values as float32.
    Idempotent on an already dequantised weight (q * s reproduces it exactly)."""
    amax = W.abs().amax(dim=1).clamp_min(1e-30)
    s = torch.exp2(torch.ceil(torch.log2(amax / E4M3_MAX)))
    q = (W / s[:, None]).to(torch.float8_e4m3fn).to(torch.float32)
    return q, s


def quantize_act_rows_e4m3(x: torch.Tensor):
    """Per-token scale = amax / 448 (1 for an all-zero row), values rounded to e4m3 (RNE); float32 arithmetic in the
    order a one-pass row quantiser uses: scale = amax * (1/448), inv = 1 / scale, q = e4m3(x * inv)."""
    amax = x.abs().amax(dim=-1)
    sc = torch.where(amax > 0, amax * torch.tensor(1.0 / E4M3_MAX, dtype=torch.float32), torch.ones_like(amax))
    inv = 1.0 / sc
    q = (x * inv[..., None]).to(torch.float8_e4m3fn).to(torch.float32)
    return q, sc


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


def swiglu(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """decoder.py:96-97: ``down_proj(nn.silu(gate_proj(x)) * up_proj(x))`` on bf16 arrays.  ``nn.silu`` is
    ``x * mx.sigmoid(x)``; under the per-op dtype rule of the module docstring that is THREE bf16 results -- sigmoid(g),
    g * sigmoid(g), (.) * up -- not one rounding of the fp32 product (rounds 1-4 of this oracle fused them; the round-4
    review's weak point 3).  Both oracle flavours and every GPU SwiGLU epilogue (csrc/common.h ``swiglu_bf``) use these
    points.  ASSUMPTION (unverifiable here, mlx absent): the fused kernel mx.compile builds for nn.silu keeps each
    primitive's output in the array dtype, as MLX's compiled-kernel generator declares its temporaries."""
    return bf(bf(gate * bf(torch.sigmoid(gate))) * up)


class DecoderOracle:
    """One sequence (the reference is batch-1 everywhere: generate.py:48,294)."""

    def __init__(self, W: dict, n_heads=16, n_kv_heads=8, head_dim=128, eps=1e-6,
                 theta=1_000_000.0, gpu_rounding=False):
        self.W = W
        self.nh, self.nkv, self.hd, self.eps, self.theta = n_heads, n_kv_heads, head_dim, eps, theta
        self.n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("layers."))
        self.gpu_rounding = gpu_rounding
        self._w8 = {}                            # name -> (e4m3 values as float32, per-row scale), built on first use
        self.perturb = 0.0                       # test aid: relative Gaussian noise on every projection before its rounding
        self.reset()

    def reset(self):
        self.k = [None] * self.n_layers          # decoder.py:20-64 (concat-grown cache)
        self.v = [None] * self.n_layers
        self.offset = 0

    def embed(self, ids) -> torch.Tensor:
        return self.W["embed_tokens.weight"][torch.as_tensor(ids, dtype=torch.long)]

    def _lin(self, x: torch.Tensor, name: str, n8: int) -> torch.Tensor:
        """x @ W.T in float32 (unrounded); rows < n8 in e4m3 x e4m3 arithmetic (module docstring)."""
        y = x @ self.W[name].T
        if n8 > 0:
            if name not in self._w8:
                self._w8[name] = quantize_weight_rows_e4m3(self.W[name])
            q, s = self._w8[name]
            x8, sa = quantize_act_rows_e4m3(x[:n8])
            y[:n8] = (x8 @ q.T) * (s[None, :] * sa[:, None])
        if self.perturb:                         # what a different fp32 summation order / exp implementation does, scaled up
            y = y * (1.0 + self.perturb * torch.randn(y.shape))
        return y

    def forward(self, h: torch.Tensor, last_only: bool = True, keep_from: int = None, fp8_rows: int = 0,
                return_hidden: bool = False) -> torch.Tensor:
        """h (T, hidden) bf16-valued float32; appends to the cache at
        self.offset and advances it.  Returns bf16-valued logits (T or 1, V)
        (decoder.py:243-253; the reference computes every position, only the
        last is consumed at generate.py:279)."""
        W = self.W
        T = h.shape[0]
        n8 = min(int(fp8_rows), T)
        pos = torch.arange(self.offset, self.offset + T)
        g = self.nh // self.nkv
        for li in range(self.n_layers):
            p = f"layers.{li}."
            x = rms_norm(h, W[p + "input_layernorm.weight"], self.eps)
            q = bf(self._lin(x, p + "self_attn.q_proj.weight", n8)).reshape(T, self.nh, self.hd)
            k = bf(self._lin(x, p + "self_attn.k_proj.weight", n8)).reshape(T, self.nkv, self.hd)
            v = bf(self._lin(x, p + "self_attn.v_proj.weight", n8)).reshape(T, self.nkv, self.hd)
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
            h = bf(h + bf(self._lin(a, p + "self_attn.o_proj.weight", n8)))
            x = rms_norm(h, W[p + "post_attention_layernorm.weight"], self.eps)
            gt = bf(self._lin(x, p + "mlp.gate_proj.weight", n8))
            up = bf(self._lin(x, p + "mlp.up_proj.weight", n8))
            act = swiglu(gt, up)
            h = bf(h + bf(self._lin(act, p + "mlp.down_proj.weight", n8)))
        self.offset += T
        if return_hidden:                       # the residual stream after the last layer, before the final norm (test aid)
            return h
        if keep_from is not None:               # teacher-forced checks: logits of rows keep_from.. only
            h = h[keep_from:]
        elif last_only:
            h = h[-1:]
        x = rms_norm(h, W["norm.weight"], self.eps)
        return bf(x @ W["embed_tokens.weight"].T)
