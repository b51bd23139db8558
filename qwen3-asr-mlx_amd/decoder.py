"""Qwen3 text decoder on HIP kernels (mirror of reference decoder.py; same class / loader names).

Two regimes: *prefill* (MFMA-bound, all prompt rows of all sequences packed into one matrix) and
*decode* (HBM-bound, one row per sequence, weight-streaming skinny GEMMs with split-K slabs,
captured into a hipGraph).  bf16 everywhere an MLX op would produce bf16; fp32 accumulation.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path
from typing import Optional, Sequence

from . import _lib, ops
from .config import TextDecoderConfig
from .tokenizer import ENDOFTEXT_TOKEN_ID, IM_END_TOKEN_ID


# Lanes: independent requests may be in flight on different host threads / HIP streams (two batches interleave well:
# one decodes, latency-bound, while the other runs its MFMA-bound encoder and prefill).  Weights and read-only plans are
# shared; everything a launch writes (decode workspace, graph, KV cache, attention workspace, side stream) is per lane.
_lane = threading.local()


def current_lane() -> int:
    return getattr(_lane, "id", 0)


def set_lane(lane: int) -> None:
    """Bind the calling thread to a lane (default 0).  Call once per worker thread, before its first request."""
    _lane.id = int(lane)


class KVCache:
    """Pre-allocated per-layer key/value cache (the reference grows by concatenation,
    decoder.py:20-64).  ``offset`` is maintained by the caller exactly as in the reference
    (generate.py:275,298).  Buffers: k, v bf16 [layers][B][n_kv][Lmax][head_dim].
    ``update(layer_idx, key, value)`` keeps the reference's method (decoder.py:32-64) for callers that drive the layers
    themselves; the HIP path writes the same buffers from its RoPE / attention kernels and never calls it."""

    def __init__(self) -> None:
        self.offset: int = 0
        self.k = None
        self.v = None
        self.batch = 0
        self.max_len = 0
        self.attn_ws = None        # q3asr_attn_decode_fused workspace (chunk partials + arrival counters), zeroed once
        self._fill: dict = {}      # update(): tokens appended per layer (the reference's per-layer array length)

    def _alloc(self, layers: int, batch: int, n_kv: int, head_dim: int, max_len: int, device) -> None:
        """(Re)allocate; contents are kept when the batch is unchanged (a longer or deeper cache)."""
        import torch
        old_k, old_v = self.k, self.v
        shape = (layers, batch, n_kv, max_len, head_dim)
        self.k = torch.zeros(shape, dtype=torch.bfloat16, device=device)
        self.v = torch.zeros(shape, dtype=torch.bfloat16, device=device)
        if old_k is not None and batch == self.batch and old_k.shape[2] == n_kv and old_k.shape[4] == head_dim:
            nl, ln = min(layers, old_k.shape[0]), min(max_len, old_k.shape[3])
            self.k[:nl, :, :, :ln] = old_k[:nl, :, :, :ln]
            self.v[:nl, :, :, :ln] = old_v[:nl, :, :, :ln]
        self.batch, self.max_len = batch, max_len
        nbytes = _lib.load().q3asr_attn_decode_fused_workspace(batch, n_kv, max_len)
        self.attn_ws = torch.zeros((nbytes // 4,), dtype=torch.float32, device=device)

    def ensure(self, cfg: TextDecoderConfig, batch: int, max_len: int, device) -> None:
        if self.k is None or batch != self.batch or max_len > self.max_len or self.k.shape[0] < cfg.num_hidden_layers:
            self._alloc(cfg.num_hidden_layers, batch, cfg.num_key_value_heads, cfg.head_dim, max_len, device)

    def update(self, layer_idx: int, key, value):
        """Append key / value (B, n_kv_heads, T_new, head_dim) to layer ``layer_idx`` and return the full (key, value) of that
        layer, shapes (B, n_kv_heads, L, head_dim) -- reference decoder.py:32-64 (first call stores, later calls concatenate on
        axis 2).  The append position is the layer's own length, as in the reference; ``offset`` stays the caller's to advance
        (generate.py:275,298).  Views of the pre-allocated buffers are returned, not copies."""
        import torch
        key, value = torch.as_tensor(key), torch.as_tensor(value)
        if key.ndim != 4 or key.shape != value.shape:
            raise ValueError(f"KVCache.update: key {tuple(key.shape)} / value {tuple(value.shape)} must both be (B, n_kv_heads, T, head_dim)")
        if layer_idx < 0:
            raise IndexError(f"KVCache.update: layer_idx {layer_idx}")
        B, nkv, T, hd = key.shape
        n = max(self._fill.get(layer_idx, 0), self.offset)
        if self.k is not None and (nkv, hd) != (self.k.shape[2], self.k.shape[4]):
            raise ValueError(f"KVCache.update: heads / head_dim {(nkv, hd)} do not match the cache {(self.k.shape[2], self.k.shape[4])}")
        if self.k is not None and B != self.batch and (n > 0 or self._fill):
            raise ValueError(f"KVCache.update: batch {B} on a cache holding batch {self.batch}")
        layers = max(layer_idx + 1, 0 if self.k is None else self.k.shape[0])
        if self.k is None or B != self.batch or n + T > self.max_len or layers > self.k.shape[0]:
            # where the buffers live: the key's device if it is a GPU tensor, else the cache's own, else the current GPU (a host-only process --
            # the CPU test suite -- keeps them on the host: this method is plain tensor bookkeeping, no kernel runs here)
            dev = key.device if key.is_cuda else (self.k.device if self.k is not None else
                                                  (torch.device(f"cuda:{torch.cuda.current_device()}") if torch.cuda.is_available() else key.device))
            self._alloc(layers, B, nkv, hd, max(n + T, 2 * self.max_len if n + T > self.max_len else self.max_len, 64), dev)
        self.k[layer_idx, :, :, n: n + T] = key.to(self.k.device, torch.bfloat16)
        self.v[layer_idx, :, :, n: n + T] = value.to(self.v.device, torch.bfloat16)
        self._fill[layer_idx] = n + T
        return self.k[layer_idx, :, :, : n + T], self.v[layer_idx, :, :, : n + T]

    def _len(self, i: int) -> int:
        return max(self.offset, self._fill.get(i, 0))

    @property
    def keys(self):
        return [] if self.k is None else [self.k[i, :, :, : self._len(i)] for i in range(self.k.shape[0])]

    @property
    def values(self):
        return [] if self.v is None else [self.v[i, :, :, : self._len(i)] for i in range(self.v.shape[0])]


def create_additive_causal_mask(N: int, offset: int = 0):
    """(N, offset+N) additive mask, -1e9 above the diagonal (reference decoder.py:71-80).  The HIP
    attention kernels apply causality by index; kept for API parity."""
    import torch
    r = torch.arange(offset + N)
    l = torch.arange(offset, offset + N)
    return (l[:, None] < r[None]).to(torch.float32) * -1e9


class _Embedding:
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, ids):
        import torch
        return self.weight[torch.as_tensor(ids, device=self.weight.device, dtype=torch.long)]


class DecodeWorkspace:
    """Device state of one decode batch; pointers are stable so one hipGraph serves every step.
    `parent` / `row0`: rows row0 .. row0+B-1 of another workspace's per-sequence state (tokens, positions, counters, flags,
    output ids, logits) seen as a batch of their own with private activation buffers; `row_base` keys the sampler's draws by
    the request-wide row index, so a row draws the same token whichever way its batch is cut (tools/ab_chains.py decodes a
    batch as two launch chains that way: measured slower)."""

    def __init__(self, cfg: TextDecoderConfig, B: int, max_out: int, device, fp8: bool = False, parent: "DecodeWorkspace" = None,
                 row0: int = 0, geo4_override: Optional[dict] = None):
        import torch
        i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=device)          # noqa: E731
        bf = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=device)        # noqa: E731
        H, I = cfg.hidden_size, cfg.intermediate_size
        qd = cfg.num_attention_heads * cfg.head_dim
        qkvd = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
        self.B, self.max_out = B, max_out
        self.parent, self.row_base = parent, int(row0)       # row_base keys the sampler's draws by the request-wide row index
        if parent is None:
            self.cur_tok, self.pos, self.n_gen, self.finished, self.max_new = i32(B), i32(B), i32(B), i32(B), i32(B)
            self.out_tokens = i32(B, max_out)
            self.logits = bf(B, cfg.vocab_size)
        else:
            if not (0 <= row0 and row0 + B <= parent.B and max_out == parent.max_out):
                raise ValueError("chain workspace: rows outside the parent")
            r = slice(row0, row0 + B)
            self.cur_tok, self.pos, self.n_gen, self.finished, self.max_new = (parent.cur_tok[r], parent.pos[r], parent.n_gen[r],
                                                                               parent.finished[r], parent.max_new[r])
            self.out_tokens, self.logits = parent.out_tokens[r], parent.logits[r]
        self.row_b = torch.arange(B, dtype=torch.int32, device=device)
        self.h = bf(B, H)
        self.cfg = cfg
        self.MT = ops.mt_of(B)
        rows = 16 * self.MT
        self.x16, self.att16, self.act16 = bf(rows * H), bf(rows * qd), bf(rows * I)     # X16-layout operands
        for t in (self.x16, self.att16, self.act16):
            t.zero_()                                                                    # rows >= B stay zero
        self.geo = {"qkv": ops.gemm_geometry(qkvd, H, fp8=fp8), "o": ops.gemm_geometry(H, qd, fp8=fp8),
                    "down": ops.gemm_geometry(H, I, fp8=fp8), "gu": ops.gemm_geometry(2 * I, H, swiglu=True, fp8=fp8),
                    "head": ops.gemm_geometry(cfg.vocab_size, H, direct=True, fp8=fp8)}
        # skinny4 geometries (None: no instantiated geometry fits this shape, skinny3 runs)
        self.geo4 = {"qkv": ops.gemm_geometry4(qkvd, H, 0, fp8, B), "o": ops.gemm_geometry4(H, qd, 0, fp8, B),
                     "down": ops.gemm_geometry4(H, I, 0, fp8, B), "gu": ops.gemm_geometry4(2 * I, H, 2, fp8, B),
                     "head": ops.gemm_geometry4(cfg.vocab_size, H, 1, fp8, B)}

        for which, g in (geo4_override or {}).items():       # tuning knob (tools/tune_decode.py): {"qkv": (2, 4, 8), ...}
            self.geo4[which] = tuple(int(v) for v in g)

        def s4(which, K):
            g = self.geo4[which]
            return K // (32 * g[1] * g[2]) if g else 1
        n_slab = max(self.geo["qkv"][0] * qkvd, self.geo["o"][0] * H, self.geo["down"][0] * H,
                     s4("qkv", H) * qkvd, s4("o", qd) * H, s4("down", I) * H)
        self.slab = torch.empty((n_slab * B,), dtype=torch.float32, device=device)
        # batch <= 2 runs the residual + RMSNorm inside the consuming GEMM: second slab buffer and residual stream to ping-pong
        self.fused_norm = B <= 2
        self.slab2 = torch.empty_like(self.slab) if self.fused_norm else None
        self.h2 = torch.empty_like(self.h) if self.fused_norm else None
        self.graph = None
        self.graph_multi = None
        self.graph_key = None
        self.graph_uses = 0

    def __del__(self):
        for g in (getattr(self, "graph", None), getattr(self, "graph_multi", None)):
            if g:
                try:
                    _lib.load().q3asr_graph_destroy(g)
                except Exception:
                    pass


class TextDecoder:
    def __init__(self, config: TextDecoderConfig) -> None:
        self.config = config
        if config.head_dim != 128 or config.num_attention_heads != 2 * config.num_key_value_heads:
            raise NotImplementedError("HIP decoder kernels are specialised for head_dim 128 and GQA 2:1")
        self.w = None
        self.device = None
        self.embed_tokens = None
        self._plans: dict = {}
        self._ws: dict = {}
        self.decode_gemm = "sk4"     # "sk4": skinny4 where a geometry fits; "sk3": skinny3 everywhere (A/B; Qwen3ASR.from_pretrained(decode_gemm=...))
        self.geo4_override: dict = {}    # launch geometries by GEMM name (tools/tune_decode.py); applies to workspaces created afterwards
        self.fuse_norm = True        # batch <= 2: residual + RMSNorm inside the consuming qkv GEMM (False: separate launches, A/B; tools/ab_batch1.py)
        self.fuse_norm_gu = True     # ... and inside gate/up too
        self._caches: dict = {}
        self._api_ws = None

    # ------------------------------------------------------------------ weights
    def load_state(self, W: dict, device, weight_dtype: str = "bf16") -> None:
        """W: checkpoint tensors with the ``model.`` prefix stripped (reference decoder.py:282-290).
        Strict key set.  The lm_head is tied to embed_tokens (decoder.py:252).
        weight_dtype "fp8" (BASELINE.json configs[4]): projection and embedding weights are quantised on
        load to OCP e4m3 with a power-of-two scale per output row.  The decode step streams the fp8 bytes
        (half the HBM traffic, widened to bf16 in registers); the prefill projections run fp8 x fp8 on the
        CDNA4 fp8 MFMA (q3asr_gemm_fp8) with the activations quantised per token, and no bf16 copy of the
        layer weights is kept.  "fp8w" is the weight-only variant (prefill on bf16 MFMAs over the
        exactly-equal bf16 values, which it keeps beside the fp8 bytes): the comparison point for what the
        activation quantisation costs in accuracy."""
        if weight_dtype not in ("bf16", "fp8", "fp8w"):
            raise ValueError(f"weight_dtype must be 'bf16', 'fp8' or 'fp8w', got {weight_dtype!r}")
        self.weight_dtype = weight_dtype
        import torch
        cfg = self.config
        dev = torch.device(device)
        need = {"embed_tokens.weight", "norm.weight"}
        names = ("input_layernorm.weight", "post_attention_layernorm.weight", "self_attn.q_proj.weight",
                 "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                 "self_attn.q_norm.weight", "self_attn.k_norm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                 "mlp.down_proj.weight")
        for i in range(cfg.num_hidden_layers):
            need.update(f"layers.{i}.{n}" for n in names)
        missing, extra = need - set(W), set(W) - need
        if missing or extra:
            raise ValueError(f"decoder weights: missing {sorted(missing)[:4]} extra {sorted(extra)[:4]}")

        def bf(t):
            return t.to(device=dev, dtype=torch.bfloat16).contiguous()

        I = cfg.intermediate_size
        fp8 = weight_dtype in ("fp8", "fp8w")
        self.fp8_prefill = weight_dtype == "fp8"            # prefill projections on the fp8 MFMA (activations quantised per token)

        def stream_form(dst: dict, name: str, mat, keep_bf16: bool = False):
            """dst[name] = row-major operand of the prefill GEMM (bf16, or fp8 bytes in "fp8" mode), dst[name+'16'] = decode
            streaming layout, dst[name+'_s'] = per-row scale.  keep_bf16: the embedding, whose rows are also gathered."""
            if fp8:
                q8, scale, w_eff = ops.quantize_fp8_rows(mat)
                dst[name + "16"], dst[name + "_s"] = ops.to_w16f8(q8), scale
                dst[name] = w_eff.contiguous() if (keep_bf16 or not self.fp8_prefill) else q8.contiguous()
            else:
                dst[name], dst[name + "16"], dst[name + "_s"] = mat, ops.to_w16(mat), None

        w = {"norm": bf(W["norm.weight"]), "layers": []}
        stream_form(w, "embed", bf(W["embed_tokens.weight"]), keep_bf16=True)
        for i in range(cfg.num_hidden_layers):
            p = f"layers.{i}."
            gate, up = W[p + "mlp.gate_proj.weight"].to(dev), W[p + "mlp.up_proj.weight"].to(dev)
            gu = torch.stack([gate.reshape(I // 16, 16, -1), up.reshape(I // 16, 16, -1)], dim=1).reshape(2 * I, -1)
            L = {"ln1": bf(W[p + "input_layernorm.weight"]), "ln2": bf(W[p + "post_attention_layernorm.weight"]),
                 "qn": bf(W[p + "self_attn.q_norm.weight"]), "kn": bf(W[p + "self_attn.k_norm.weight"])}
            stream_form(L, "wqkv", bf(torch.cat([W[p + "self_attn.q_proj.weight"], W[p + "self_attn.k_proj.weight"],
                                                 W[p + "self_attn.v_proj.weight"]], 0)))
            stream_form(L, "wo", bf(W[p + "self_attn.o_proj.weight"]))
            stream_form(L, "wgu", bf(gu))
            stream_form(L, "wdown", bf(W[p + "mlp.down_proj.weight"]))
            w["layers"].append(L)
        half = cfg.head_dim // 2
        # MLX nn.RoPE(traditional=False): angle = pos * base**(-i/half) (decoder.py:126)
        w["inv_freq"] = (float(cfg.rope_theta) ** (-torch.arange(half, dtype=torch.float32) / half)).to(dev)
        self.w, self.device = w, dev
        self.embed_tokens = _Embedding(w["embed"])

    # ------------------------------------------------------------------ prefill (hot path)
    def _prefill_plan(self, seq_lens: tuple, Lmax: int, slots: Optional[tuple] = None, pos0: int = 0):
        """Index vectors of one packed prefill.  `slots`: cache rows the sequences go to (default 0..n-1); `pos0`: tokens
        already in the cache in front of every sequence (a multi-token call on a non-empty cache, decoder.py:151-158)."""
        slots = tuple(range(len(seq_lens))) if slots is None else slots
        key = (seq_lens, Lmax, slots, pos0)
        plans = self._plans.setdefault(current_lane(), {})   # per lane: eviction frees index tensors another lane could still be reading
        pl = plans.get(key)
        if pl is None:
            import numpy as np
            import torch
            cfg, dev = self.config, self.device
            lens = np.asarray(seq_lens, dtype=np.int64)
            ends = np.cumsum(lens)
            starts = ends - lens
            m = int(ends[-1])
            row_b = np.repeat(np.asarray(slots, dtype=np.int32), lens)
            row_pos = (np.arange(m, dtype=np.int64) - np.repeat(starts, lens) + pos0).astype(np.int32)
            i32 = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.int32)).to(dev)      # noqa: E731
            pl = {"row_b": i32(row_b), "row_pos": i32(row_pos), "seg_start": i32(starts), "seg_len": i32(lens),
                  "last": i32(ends - 1), "M": m, "max_len": int(lens.max()),
                  "k_off": i32(np.full(len(lens), pos0)) if pos0 else None, "max_klen": int(lens.max()) + pos0,
                  "kv_off": torch.from_numpy(np.asarray(slots, dtype=np.int64) * (cfg.num_key_value_heads * Lmax * cfg.head_dim)).to(dev)}
            if len(plans) > 64:
                plans.clear()
            plans[key] = pl
        return pl

    def _layers_prefill(self, h, cache: KVCache, seq_lens: Sequence[int], slots: Optional[Sequence[int]] = None, trace: Optional[list] = None,
                        pos0: int = 0):
        """h bf16 [M][H] (packed rows of all sequences, positions start at pos0: the cache already holds pos0 tokens of every
        sequence) -> updated in place.
        `trace` (tests): a list that receives, per layer, a dict of clones of every intermediate the layer hands from one
        kernel to the next, so that each stage can be checked against the oracle from the inputs the GPU itself saw."""
        def keep(li, **kw):
            if trace is not None:
                while len(trace) <= li:
                    trace.append({})
                trace[li].update({k: v.clone() for k, v in kw.items()})

        import torch
        cfg, w, dev = self.config, self.w, self.device
        if slots is not None:
            slots = tuple(int(s) for s in slots)
            if len(slots) != len(seq_lens) or len(set(slots)) != len(slots) or min(slots) < 0 or max(slots) >= cache.batch:
                raise ValueError(f"prefill: slots {slots} must be {len(seq_lens)} distinct rows of a cache of {cache.batch}")
        if pos0 < 0 or pos0 + max(seq_lens) > cache.max_len:
            raise ValueError(f"prefill: {pos0} cached + {max(seq_lens)} new tokens do not fit a cache of {cache.max_len}")
        pl = self._prefill_plan(tuple(int(t) for t in seq_lens), cache.max_len, slots, int(pos0))
        M, H, I = pl["M"], cfg.hidden_size, cfg.intermediate_size
        nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        qd, qkvd = nh * hd, (nh + 2 * nkv) * hd
        bf16 = torch.bfloat16
        x = torch.empty((M, H), dtype=bf16, device=dev)
        qkv = torch.empty((M, qkvd), dtype=bf16, device=dev)
        q = torch.empty((M, qd), dtype=bf16, device=dev)
        att = torch.empty((M, qd), dtype=bf16, device=dev)
        act = torch.empty((M, I), dtype=bf16, device=dev)
        eps, scale, Lmax = cfg.rms_norm_eps, hd ** -0.5, cache.max_len
        if getattr(self, "fp8_prefill", False):
            # BASELINE.json configs[4]: every projection fp8 x fp8 on the fp8 MFMA.  Activations are quantised per token where
            # the bf16 path would hand them to the GEMM: RMSNorm output (fused), attention output, SwiGLU output.
            u8, f32 = torch.uint8, torch.float32
            x8 = torch.empty((M, H), dtype=u8, device=dev)
            a8 = torch.empty((M, qd), dtype=u8, device=dev)
            c8 = torch.empty((M, I), dtype=u8, device=dev)
            sx, sa_, sc = (torch.empty((M,), dtype=f32, device=dev) for _ in range(3))
            for li, L in enumerate(w["layers"]):
                keep(li, h_in=h)
                ops.rmsnorm_fp8(h, L["ln1"], x8, sx, M, H, eps)
                ops.gemm_fp8(x8, sx, L["wqkv"], L["wqkv_s"], M, qkvd, H, qkv)
                keep(li, x8_1=x8, sx_1=sx, qkv=qkv)
                ops.call("q3asr_qk_norm_rope_kv", ops.ptr(qkv), 0, qkvd, 0, 0, ops.ptr(pl["row_b"]), ops.ptr(pl["row_pos"]),
                         ops.ptr(L["qn"]), ops.ptr(L["kn"]), ops.ptr(w["inv_freq"]), eps, ops.ptr(q), ops.ptr(cache.k[li]),
                         ops.ptr(cache.v[li]), M, nh, nkv, hd, Lmax, ops.sp())
                ops.attn_fwd(ops.ptr(q), qd, ops.ptr(cache.k[li]), ops.ptr(cache.v[li]), pl["kv_off"], Lmax * hd, hd, att, qd,
                             pl["seg_start"], pl["seg_len"], len(seq_lens), pl["max_len"], nh, nkv, hd, True, scale,
                             seg_k_off=pl["k_off"], max_klen=pl["max_klen"])
                ops.quantize_rows_fp8(att, a8, sa_, M, qd)
                ops.gemm_fp8(a8, sa_, L["wo"], L["wo_s"], M, H, qd, h, res=h)
                keep(li, q=q, att=att, a8=a8, sa=sa_, h_mid=h)
                ops.rmsnorm_fp8(h, L["ln2"], x8, sx, M, H, eps)
                ops.gemm_fp8(x8, sx, L["wgu"], L["wgu_s"], M, 2 * I, H, act, act=2)
                ops.quantize_rows_fp8(act, c8, sc, M, I)
                keep(li, x8_2=x8, sx_2=sx, act=act, c8=c8, sc=sc)
                ops.gemm_fp8(c8, sc, L["wdown"], L["wdown_s"], M, H, I, h, res=h)
                keep(li, h_out=h)
            return pl
        for li, L in enumerate(w["layers"]):
            keep(li, h_in=h)
            ops.rmsnorm(h, L["ln1"], x, M, H, eps)
            ops.gemm(x, L["wqkv"], M, qkvd, H, qkv)
            keep(li, x_1=x, qkv=qkv)
            ops.call("q3asr_qk_norm_rope_kv", ops.ptr(qkv), 0, qkvd, 0, 0, ops.ptr(pl["row_b"]), ops.ptr(pl["row_pos"]),
                     ops.ptr(L["qn"]), ops.ptr(L["kn"]), ops.ptr(w["inv_freq"]), eps, ops.ptr(q), ops.ptr(cache.k[li]),
                     ops.ptr(cache.v[li]), M, nh, nkv, hd, Lmax, ops.sp())
            ops.attn_fwd(ops.ptr(q), qd, ops.ptr(cache.k[li]), ops.ptr(cache.v[li]), pl["kv_off"], Lmax * hd, hd, att, qd,
                         pl["seg_start"], pl["seg_len"], len(seq_lens), pl["max_len"], nh, nkv, hd, True, scale,
                             seg_k_off=pl["k_off"], max_klen=pl["max_klen"])
            ops.gemm(att, L["wo"], M, H, qd, h, res=h, res_kind=2, ldres=H)
            keep(li, q=q, att=att, h_mid=h)
            ops.rmsnorm(h, L["ln2"], x, M, H, eps)
            ops.gemm(x, L["wgu"], M, 2 * I, H, act, act=2)
            keep(li, x_2=x, act=act)
            ops.gemm(act, L["wdown"], M, H, I, h, res=h, res_kind=2, ldres=H)
            keep(li, h_out=h)
        return pl

    def prefill(self, embeds, seq_lens: Sequence[int], cache: KVCache, logits_out=None, slots: Optional[Sequence[int]] = None):
        """Packed prompt embeddings bf16 [sum T_b][H] -> bf16 logits of each sequence's LAST
        position [B][V] (the reference computes all T rows, decoder.py:251-252, and consumes only the
        last, generate.py:279).  Fills cache positions 0..T_b-1 of cache rows `slots` (default 0..B-1; other rows of the
        cache are left alone, so a sequence can be admitted while its neighbours are mid-decode)."""
        import torch
        cfg, w = self.config, self.w
        B = len(seq_lens)
        h = embeds.clone()
        pl = self._layers_prefill(h, cache, seq_lens, slots)
        if logits_out is None:
            logits_out = torch.empty((B, cfg.vocab_size), dtype=torch.bfloat16, device=self.device)
        self._lm_head_rows(h, pl["last"], pl["M"], B, logits_out)
        return logits_out

    def _lm_head_rows(self, h, rows, n_rows_h: int, M: int, out) -> None:
        """Final RMSNorm of rows `rows` (int32 [M], device) of h and the tied head (decoder.py:251-252) for M <= 64 rows: the
        decode step's own kernels (gathered norm straight into the X16 operand layout, then the weight-streaming GEMM over
        the re-tiled embedding), so the first token's logits come out of the same arithmetic as every later one's."""
        import torch
        cfg, w, H = self.config, self.w, self.config.hidden_size
        fp8 = getattr(self, "weight_dtype", "bf16") != "bf16"
        for m0 in range(0, M, 64):
            mm = min(64, M - m0)
            MT = ops.mt_of(mm)
            x16 = torch.zeros((16 * MT * H,), dtype=torch.bfloat16, device=self.device)
            scratch = torch.empty((mm, H), dtype=torch.bfloat16, device=self.device)
            ops.call("q3asr_embed_norm_x16", ops.ptr(rows[m0:]), ops.ptr(h), int(n_rows_h), ops.ptr(scratch), ops.ptr(w["norm"]), ops.ptr(x16),
                     mm, H, cfg.rms_norm_eps, ops.sp())
            g4 = ops.gemm_geometry4(cfg.vocab_size, H, 1, fp8, mm) if self.decode_gemm == "sk4" else None
            if g4 is not None:
                ops.skinny4(x16, mm, w["embed16"], cfg.vocab_size, H, 1, *g4, out=out[m0:], scale=w["embed_s"])
            else:
                S, waves, rt, ksplit = ops.gemm_geometry(cfg.vocab_size, H, direct=True, fp8=fp8)
                ops.skinny3(x16, mm, w["embed16"], cfg.vocab_size, H, 1, S, waves, out=out[m0:], rt=rt, ksplit=ksplit, scale=w["embed_s"])

    # ------------------------------------------------------------------ decode step (hot path)
    def workspace(self, B: int, max_out: int) -> DecodeWorkspace:
        key = (current_lane(), B, max_out)
        ws = self._ws.get(key)
        if ws is None:
            mine = [k for k in self._ws if k[0] == key[0]]
            if len(mine) > 8:                                # evict this lane's entries only: another lane may be mid-request
                for k in mine:
                    del self._ws[k]
            ws = self._ws[key] = DecodeWorkspace(self.config, B, max_out, self.device, fp8=getattr(self, "weight_dtype", "bf16") != "bf16",
                                                 geo4_override=self.geo4_override)
        return ws

    def batch_cache(self, B: int, Lmax: int) -> KVCache:
        """Decoder-owned cache with stable device pointers (so a captured decode graph stays valid
        across calls with the same batch geometry)."""
        key = (current_lane(), B, Lmax)
        c = self._caches.get(key)
        if c is None:
            mine = [k for k in self._caches if k[0] == key[0]]
            if len(mine) > 2:                                # this lane's entries only
                for k in mine:
                    del self._caches[k]
            c = self._caches[key] = KVCache()
            c.ensure(self.config, B, Lmax, self.device)
        c.offset = 0
        return c

    def decode_step(self, ws: DecodeWorkspace, cache: KVCache, sampling: dict) -> None:
        """One token for every sequence: embed(cur_tok) -> 28 layers at position pos[b] -> tied head ->
        sample -> bookkeeping.  7 launches per layer (5 at batch <= 2), no host interaction (graph-capturable)."""
        cfg, w = self.config, self.w
        B, H, I = ws.B, cfg.hidden_size, cfg.intermediate_size
        nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        qd, qkvd = nh * hd, (nh + 2 * nkv) * hd
        eps, scale, Lmax = cfg.rms_norm_eps, hd ** -0.5, cache.max_len
        layers = w["layers"]
        ops.call("q3asr_embed_norm_x16", ops.ptr(ws.cur_tok), ops.ptr(w["embed"]), cfg.vocab_size, ops.ptr(ws.h), ops.ptr(layers[0]["ln1"]),
                 ops.ptr(ws.x16), B, H, eps, ops.sp())
        self._layers_decode(ws, cache, ws.pos)
        self._gemm3(ws, "head", ws.x16, w, "embed", cfg.vocab_size, H, 1, out=ws.logits)
        self.sample_step(ws, ws.logits, sampling)

    def _gemm3(self, ws: DecodeWorkspace, which: str, X16, holder: dict, name: str, N, K, epi, out=None, slab=None) -> int:
        g4 = ws.geo4[which] if self.decode_gemm == "sk4" else None
        if g4 is not None:
            tpw, nks, waves = g4
            ops.skinny4(X16, ws.B, holder[name + "16"], N, K, epi, tpw, nks, waves,
                        slab=(slab if slab is not None else ws.slab) if epi == 0 else None, out=out, scale=holder[name + "_s"])
            return K // (32 * nks * waves)
        S, waves, rt, ksplit = ws.geo[which]
        ops.skinny3(X16, ws.B, holder[name + "16"], N, K, epi, S, waves, slab=(slab if slab is not None else ws.slab) if epi == 0 else None,
                    out=out, rt=rt, ksplit=ksplit, scale=holder[name + "_s"])
        return S

    def _gemm3_norm(self, ws: DecodeWorkspace, which: str, in_slab, s_in, h_in, h_out, norm_w, holder: dict, name: str, N, K, epi,
                    out=None, slab=None) -> int:
        """`_gemm3` with residual + RMSNorm of the operand fused in (q3asr_skinny3_gemm_norm: 4 waves per workgroup)."""
        g4 = ws.geo4[which] if self.decode_gemm == "sk4" else None
        if g4 is not None:
            tpw, nks, waves = g4
            ops.skinny4_norm(in_slab, s_in, ws.B * K, h_in, h_out, norm_w, self.config.rms_norm_eps, ws.B, holder[name + "16"], N, K, epi,
                             tpw, nks, waves, slab=slab if epi == 0 else None, out=out, scale=holder[name + "_s"])
            return K // (32 * nks * waves)
        S, waves, rt, ksplit = ws.geo[which]
        ops.skinny3_norm(in_slab, s_in, ws.B * K, h_in, h_out, norm_w, self.config.rms_norm_eps, ws.B, holder[name + "16"], N, K, epi, S,
                         slab=slab if epi == 0 else None, out=out, rt=rt, ksplit=ksplit, scale=holder[name + "_s"])
        return S

    def _layers_decode(self, ws: DecodeWorkspace, cache: KVCache, pos, trace: Optional[list] = None) -> None:
        """7 launches per layer, every GEMM operand in its streaming layout (W16 / X16):
        qkv GEMM (slabs) -> attention with fused QK-norm / RoPE / KV append -> o GEMM (slabs) ->
        residual + RMSNorm -> gate/up GEMM with SwiGLU epilogue -> down GEMM (slabs) -> residual + RMSNorm.
        ws.x16 holds the normed input on entry and the final-norm output on exit.
        `trace` (tests; the 7-launch form only): a list that receives, per layer, clones of what every launch handed to the
        next -- operands back in row-major form, split-K slabs as written -- so that each stage can be checked against the
        oracle from the inputs the GPU itself saw (tests/test_parity_gpu.py)."""
        cfg, w = self.config, self.w
        if trace is not None and ws.fused_norm and self.fuse_norm:
            raise ValueError("decode trace: the batch <= 2 form fuses the norms into the GEMMs; trace a batch of 3 or more")

        def keep(li, **kw):
            if trace is not None:
                while len(trace) <= li:
                    trace.append({})
                trace[li].update({k: (v.clone() if hasattr(v, "clone") else v) for k, v in kw.items()})
        B, H, I = ws.B, cfg.hidden_size, cfg.intermediate_size
        nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        qd, qkvd = nh * hd, (nh + 2 * nkv) * hd
        eps, scale, Lmax = cfg.rms_norm_eps, hd ** -0.5, cache.max_len
        layers = w["layers"]
        if ws.fused_norm and self.fuse_norm:
            # batch <= 2: 5 launches per layer.  Slabs: o / down -> ws.slab, read by the fused prologues of gate/up and of the
            # next layer's qkv, whose own slabs go to ws.slab2 (a kernel must not write the buffer its neighbours still read);
            # the residual stream ping-pongs between ws.h and ws.h2 for the same reason.
            h_cur, h_oth, s_d = ws.h, ws.h2, 0
            for li, L in enumerate(layers):
                if li == 0:
                    s_qkv = self._gemm3(ws, "qkv", ws.x16, L, "wqkv", qkvd, H, 0, slab=ws.slab2)
                else:
                    s_qkv = self._gemm3_norm(ws, "qkv", ws.slab, s_d, h_cur, h_oth, L["ln1"], L, "wqkv", qkvd, H, 0, slab=ws.slab2)
                    h_cur, h_oth = h_oth, h_cur
                ops.call("q3asr_attn_decode_fused", ops.ptr(ws.slab2), qkvd, s_qkv, B * qkvd, ops.ptr(L["qn"]), ops.ptr(L["kn"]),
                         ops.ptr(w["inv_freq"]), eps, ops.ptr(cache.k[li]), ops.ptr(cache.v[li]), ops.ptr(pos), ops.ptr(ws.att16),
                         B, nh, nkv, hd, Lmax, scale, ops.ptr(cache.attn_ws), ops.sp())
                s_o = self._gemm3(ws, "o", ws.att16, L, "wo", H, qd, 0)
                if self.fuse_norm_gu:
                    self._gemm3_norm(ws, "gu", ws.slab, s_o, h_cur, h_oth, L["ln2"], L, "wgu", 2 * I, H, 2, out=ws.act16)
                    h_cur, h_oth = h_oth, h_cur
                else:
                    ops.call("q3asr_slab_resnorm_x16", ops.ptr(ws.slab), s_o, B * H, ops.ptr(h_cur), ops.ptr(h_cur),
                             ops.ptr(L["ln2"]), ops.ptr(ws.x16), B, H, eps, ops.sp())
                    self._gemm3(ws, "gu", ws.x16, L, "wgu", 2 * I, H, 2, out=ws.act16)
                s_d = self._gemm3(ws, "down", ws.act16, L, "wdown", H, I, 0)
            ops.call("q3asr_slab_resnorm_x16", ops.ptr(ws.slab), s_d, B * H, ops.ptr(h_cur), ops.ptr(h_cur),
                     ops.ptr(w["norm"]), ops.ptr(ws.x16), B, H, eps, ops.sp())
            return
        for li, L in enumerate(layers):
            if trace is not None:
                keep(li, x_in=ops.from_x16(ws.x16, B, ws.MT), h_in=ws.h)
            s_qkv = self._gemm3(ws, "qkv", ws.x16, L, "wqkv", qkvd, H, 0)
            if trace is not None:
                keep(li, qkv_slabs=ws.slab[: s_qkv * B * qkvd].view(s_qkv, B, qkvd))
            ops.call("q3asr_attn_decode_fused", ops.ptr(ws.slab), qkvd, s_qkv, B * qkvd, ops.ptr(L["qn"]), ops.ptr(L["kn"]),
                     ops.ptr(w["inv_freq"]), eps, ops.ptr(cache.k[li]), ops.ptr(cache.v[li]), ops.ptr(pos), ops.ptr(ws.att16),
                     B, nh, nkv, hd, Lmax, scale, ops.ptr(cache.attn_ws), ops.sp())
            if trace is not None:
                keep(li, att=ops.from_x16(ws.att16, B, ws.MT))
            s_o = self._gemm3(ws, "o", ws.att16, L, "wo", H, qd, 0)
            if trace is not None:
                keep(li, o_slabs=ws.slab[: s_o * B * H].view(s_o, B, H))
            ops.call("q3asr_slab_resnorm_x16", ops.ptr(ws.slab), s_o, B * H, ops.ptr(ws.h), ops.ptr(ws.h),
                     ops.ptr(L["ln2"]), ops.ptr(ws.x16), B, H, eps, ops.sp())
            if trace is not None:
                keep(li, h_mid=ws.h, x_2=ops.from_x16(ws.x16, B, ws.MT))
            self._gemm3(ws, "gu", ws.x16, L, "wgu", 2 * I, H, 2, out=ws.act16)
            if trace is not None:
                keep(li, act=ops.from_x16(ws.act16, B, ws.MT))
            s_d = self._gemm3(ws, "down", ws.act16, L, "wdown", H, I, 0)
            if trace is not None:
                keep(li, down_slabs=ws.slab[: s_d * B * H].view(s_d, B, H))
            nxt = layers[li + 1]["ln1"] if li + 1 < len(layers) else w["norm"]
            ops.call("q3asr_slab_resnorm_x16", ops.ptr(ws.slab), s_d, B * H, ops.ptr(ws.h), ops.ptr(ws.h),
                     ops.ptr(nxt), ops.ptr(ws.x16), B, H, eps, ops.sp())
            if trace is not None:
                keep(li, h_out=ws.h, x_next=ops.from_x16(ws.x16, B, ws.MT))

    def gemm_only_step(self, ws: DecodeWorkspace) -> int:
        """bench.py roofline leg: exactly the weight-streaming GEMM launches of one decode step (4 per
        layer + the tied head), same weights / operands / launch geometry, nothing else.  Returns the count."""
        cfg, w = self.config, self.w
        H, I = cfg.hidden_size, cfg.intermediate_size
        qd = cfg.num_attention_heads * cfg.head_dim
        qkvd = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
        for L in w["layers"]:
            self._gemm3(ws, "qkv", ws.x16, L, "wqkv", qkvd, H, 0)
            self._gemm3(ws, "o", ws.att16, L, "wo", H, qd, 0)
            self._gemm3(ws, "gu", ws.x16, L, "wgu", 2 * I, H, 2, out=ws.act16)
            self._gemm3(ws, "down", ws.act16, L, "wdown", H, I, 0)
        self._gemm3(ws, "head", ws.x16, w, "embed", cfg.vocab_size, H, 1, out=ws.logits)
        return 4 * len(w["layers"]) + 1

    def sample_step(self, ws: DecodeWorkspace, logits, s: dict) -> None:
        ops.call("q3asr_sample_step", ops.ptr(logits), logits.stride(0), ws.B, self.config.vocab_size, ops.ptr(ws.out_tokens),
                 ws.max_out, ops.ptr(ws.n_gen), ops.ptr(ws.finished), ops.ptr(ws.cur_tok), ops.ptr(ws.pos), ops.ptr(ws.max_new),
                 float(s["temperature"]), float(s["top_p"]), int(s["top_k"]), float(s["repetition_penalty"]),
                 int(s["repetition_context_size"]), int(s.get("seed", 0)), ENDOFTEXT_TOKEN_ID, IM_END_TOKEN_ID, int(getattr(ws, "row_base", 0)),
                 ops.sp())

    def sample_rows(self, ws: DecodeWorkspace, logits_row, row: int, s: dict) -> None:
        """sample_step for ONE row of the workspace from a separate logits vector (a sequence admitted into a running batch
        draws its first token without touching its neighbours): same kernel, batch 1, pointers offset to `row`."""
        V = self.config.vocab_size
        ops.call("q3asr_sample_step", ops.ptr(logits_row), V, 1, V, ops.ptr(ws.out_tokens[row]),
                 ws.max_out, ops.ptr(ws.n_gen[row:]), ops.ptr(ws.finished[row:]), ops.ptr(ws.cur_tok[row:]), ops.ptr(ws.pos[row:]),
                 ops.ptr(ws.max_new[row:]), float(s["temperature"]), float(s["top_p"]), int(s["top_k"]),
                 float(s["repetition_penalty"]), int(s["repetition_context_size"]), int(s.get("seed", 0)) + 0x9E3779B97F4A7C15 * (row + 1) & 0x7FFFFFFFFFFFFFFF,
                 ENDOFTEXT_TOKEN_ID, IM_END_TOKEN_ID, 0, ops.sp())

    # ------------------------------------------------------------------ hipGraph
    MULTI_STEPS = 4     # decode steps per launch of the second graph (a replay boundary costs ~8 us of idle GPU)

    def capture_step(self, ws: DecodeWorkspace, cache: KVCache, sampling: dict) -> None:
        import torch
        # the graph bakes raw pointers: key it on the addresses it captured, not on Python object identity
        # (every buffer the captured launches address: K/V blocks AND the attention workspace -- a re-created cache can get
        # its large K/V blocks back from the caching allocator while the small workspace lands elsewhere -- and every switch
        # that changes which launches are recorded)
        key = (cache.k.data_ptr(), cache.v.data_ptr(), cache.attn_ws.data_ptr(), cache.max_len, self.decode_gemm,
               bool(self.fuse_norm), bool(self.fuse_norm_gu), tuple(sorted(sampling.items())))
        if ws.graph is not None and ws.graph_key == key:
            ws.graph_uses += 1
            if ws.graph_multi is None and ws.graph_uses >= 2:        # a workspace that keeps coming back earns the long graph
                ws.graph_multi = self._capture(ws, cache, sampling, self.MULTI_STEPS)
            return
        for g in (ws.graph, ws.graph_multi):
            if g is not None:
                _lib.call("q3asr_graph_destroy", g)
        ws.graph = ws.graph_multi = None
        ws.graph, ws.graph_key, ws.graph_uses = self._capture(ws, cache, sampling, 1), key, 1

    def _capture(self, ws: DecodeWorkspace, cache: KVCache, sampling: dict, n_steps: int):
        """Record n_steps consecutive decode steps into one graph (stream capture records, it does not run them)."""
        import torch
        stream = torch.cuda.current_stream()
        _lib.call("q3asr_graph_begin", C.c_void_p(stream.cuda_stream))
        try:
            for _ in range(n_steps):
                self.decode_step(ws, cache, sampling)
        finally:
            g = C.c_void_p()
            _lib.call("q3asr_graph_end", C.c_void_p(stream.cuda_stream), C.byref(g))
        return g

    def replay_step(self, ws: DecodeWorkspace, multi: bool = False) -> int:
        """Replay one captured step, or MULTI_STEPS of them when `multi` and the long graph exists.  Returns the steps run."""
        import torch
        use_multi = multi and ws.graph_multi is not None
        _lib.call("q3asr_graph_launch", ws.graph_multi if use_multi else ws.graph, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return self.MULTI_STEPS if use_multi else 1

    # ------------------------------------------------------------------ reference-shaped call
    def __call__(self, inputs, cache: Optional[KVCache] = None, is_embeds: bool = False):
        """Token ids (B,T) or embeddings (B,T,H) -> bf16 logits (B,T,V) for EVERY position, appending
        to ``cache`` at ``cache.offset`` (reference decoder.py:219-253; the caller advances offset).  T > 1 on a non-empty
        cache -- the case create_additive_causal_mask(N, offset) exists for -- runs the prefill kernels with every position
        shifted by the offset; T == 1 on a non-empty cache runs the decode-step kernels."""
        import torch
        if self.w is None:
            raise TypeError("TextDecoder has no weights loaded")
        cfg, w, dev = self.config, self.w, self.device
        t = torch.as_tensor(inputs) if not isinstance(inputs, torch.Tensor) else inputs
        h = (t.to(dev, torch.bfloat16) if is_embeds else self.embed_tokens(t.to(dev)))
        if h.ndim == 2:
            h = h[None]
        B, T, H = h.shape
        if T < 1 or B < 1:
            raise ValueError(f"decoder call: empty input {tuple(h.shape)}")
        if B > 64:
            raise NotImplementedError("reference-shaped call: at most 64 sequences per call; use prefill()/generate_batch for larger batches")
        hp = h.reshape(B * T, H).contiguous().clone()
        cache = cache if cache is not None else KVCache()
        off = cache.offset
        if cache.batch != B and off > 0:
            raise ValueError(f"decoder call: batch {B} on a cache holding {off} tokens of batch {cache.batch}")
        if cache.k is None or off + T > cache.max_len or cache.batch != B or cache.k.shape[0] < cfg.num_hidden_layers:
            cache.ensure(cfg, B, max(off + T, 2 * cache.max_len if cache.batch == B else 0, 64), dev)
        if T == 1 and off > 0:
            x = self._single_step_hidden(hp, cache, off)
        else:
            self._layers_prefill(hp, cache, [T] * B, pos0=off)
            x = torch.empty_like(hp)
            ops.rmsnorm(hp, w["norm"], x, B * T, H, cfg.rms_norm_eps)
        logits = torch.empty((B * T, cfg.vocab_size), dtype=torch.bfloat16, device=dev)
        ops.gemm(x, w["embed"], B * T, cfg.vocab_size, H, logits)      # every position (decoder.py:252), bf16 rows of the tied embedding
        return logits.view(B, T, cfg.vocab_size)

    def _single_step_hidden(self, h, cache: KVCache, off: int):
        """One token per sequence (h bf16 [B][H]) at position `off` through the decode kernels; returns the final-norm output."""
        import torch
        cfg, w = self.config, self.w
        B = h.shape[0]
        ws = self._api_ws
        if ws is None or ws.B != B:
            ws = self._api_ws = DecodeWorkspace(cfg, B, 1, self.device, fp8=getattr(self, "weight_dtype", "bf16") != "bf16",
                                                geo4_override=self.geo4_override)
        H, eps = cfg.hidden_size, cfg.rms_norm_eps
        ws.pos.fill_(off)
        layers = w["layers"]
        ops.call("q3asr_embed_norm_x16", None, ops.ptr(h), B, ops.ptr(ws.h), ops.ptr(layers[0]["ln1"]), ops.ptr(ws.x16), B, H,
                 eps, ops.sp())
        self._layers_decode(ws, cache, ws.pos)
        return ops.from_x16(ws.x16, B, ws.MT).contiguous()


def load_decoder_weights(decoder: TextDecoder, model_path: str | Path, device: Optional[str] = None,
                         weight_dtype: str = "bf16") -> None:
    """Load ``model.*`` from model.safetensors (reference decoder.py:260-291)."""
    import torch
    from .encoder import _read_safetensors
    dev = device or f"cuda:{torch.cuda.current_device()}"
    decoder.load_state(_read_safetensors(model_path, "model."), dev, weight_dtype)
