"""Thin Python wrappers over the C-ABI (argument marshalling only; no arithmetic here)."""
from __future__ import annotations

import ctypes as C

from . import _lib

ptr, sp, call = _lib.ptr, _lib.stream_ptr, _lib.call


def off_ptr(t, elem_offset: int):
    return C.c_void_p(t.data_ptr() + elem_offset * t.element_size())


def gemm(A, W, M, N, K, out, *, lda=None, ldw=None, bias=None, act=0, res=None, res_kind=0, ldres=0, res_mod=0,
         out_fp32=False, ldo=None, row_map=None):
    call("q3asr_gemm_bf16", ptr(A), lda if lda is not None else K, ptr(W), ldw if ldw is not None else K, M, N, K,
         ptr(bias), act, ptr(res), res_kind, ldres, res_mod, ptr(out), 1 if out_fp32 else 0,
         ldo if ldo is not None else (N // 2 if act == 2 else N), ptr(row_map), sp())


def conv_weight_chunked(w):
    """[Cout][3][3][Cin] (tap order = the activations' two spatial dims) -> chunk-major K [Cout][ceil(Cin/64) * 9 * 64]
    with zero channels beyond Cin (csrc/gemm.hip, q3asr_conv3x3s2_bf16).  One-time re-tiling of static weights."""
    import torch
    Cout, Cin = w.shape[0], w.shape[-1]
    nch = (Cin + 63) // 64
    t = torch.zeros((Cout, 9, nch * 64), dtype=w.dtype, device=w.device)
    t[:, :, :Cin] = w.reshape(Cout, 9, Cin)
    return t.reshape(Cout, 9, nch, 64).permute(0, 2, 1, 3).reshape(Cout, nch * 576).contiguous()


def conv3x3s2(x, imgs, Hin, Win, Cin, w, Cout, bias, act, out, out_fp32=False):
    """w: conv_weight_chunked layout."""
    call("q3asr_conv3x3s2_bf16", ptr(x), imgs, Hin, Win, Cin, ptr(w), Cout, ptr(bias), act, ptr(out), 1 if out_fp32 else 0, Cout, None, sp())


def gemm_fp8(A8, a_scale, W8, w_scale, M, N, K, out, act=0, res=None, ldo=None):
    """fp8 x fp8 GEMM (csrc/gemm.hip, q3asr_gemm_fp8): bytes [M][K] x bytes [N][K] -> bf16; res: bf16 residual (MLX add)."""
    call("q3asr_gemm_fp8", ptr(A8), K, ptr(a_scale), ptr(W8), K, ptr(w_scale), M, N, K, act, ptr(res), 2 if res is not None else 0,
         N if res is not None else 0, ptr(out), ldo if ldo is not None else (N // 2 if act == 2 else N), sp())


def rmsnorm_fp8(x, w, y8, scale, M, D, eps):
    call("q3asr_rmsnorm_bf16_fp8", ptr(x), D, ptr(w), ptr(y8), D, ptr(scale), M, D, eps, sp())


def quantize_rows_fp8(x, y8, scale, M, D):
    call("q3asr_quantize_rows_fp8", ptr(x), D, ptr(y8), D, ptr(scale), M, D, sp())


def split3(x, M, K, y, ldx=None):
    """fp32 [M][K] -> bf16 [M][3K] = [hi | mid | lo] (csrc/parity.hip)."""
    call("q3asr_split3_f32", ptr(x), ldx or K, ptr(y), M, K, sp())


def layernorm(x, w, b, y, M, D, eps):
    call("q3asr_layernorm_f32_bf16", ptr(x), ptr(w), ptr(b), ptr(y), M, D, eps, sp())


def rmsnorm(x, w, y, M, D, eps, ldx=None, ldy=None, row_idx=None):
    call("q3asr_rmsnorm_bf16", ptr(x), ldx or D, ptr(w), ptr(y), ldy or D, ptr(row_idx), M, D, eps, sp())


def attn_fwd(Q, q_stride, K, V, kv_seg_off, kv_head_stride, kv_row_stride, O, o_stride, seg_q_start, seg_len, n_segs,
             max_len, n_heads, n_kv_heads, head_dim, causal, scale, seg_k_off=None, max_klen=None):
    """seg_k_off / max_klen: keys already cached in front of each segment's first query row (include/q3asr.h)."""
    call("q3asr_attn_fwd", Q, q_stride, K, V, ptr(kv_seg_off), kv_head_stride, kv_row_stride, ptr(O), o_stride,
         ptr(seg_q_start), ptr(seg_len), ptr(seg_k_off), n_segs, max_len, max_len if max_klen is None else max_klen,
         n_heads, n_kv_heads, head_dim, 1 if causal else 0, scale, sp())


# ---------------------------------------------------------------- decode-step streaming layouts
def mt_of(M: int) -> int:
    return 1 if M <= 16 else (2 if M <= 32 else (3 if M <= 48 else 4))


def to_w16(W):
    """[N][K] -> W16 [N/16][K/32][64][8] (csrc/decode.hip); one-time re-tiling of static weights."""
    N, K = W.shape
    return W.reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def to_x16(X, MT=None):
    """[M][K] -> X16 [K/32][MT][64][8], rows padded with zeros (tests / API entry only)."""
    import torch
    M, K = X.shape
    MT = MT or mt_of(M)
    pad = torch.zeros((16 * MT, K), dtype=X.dtype, device=X.device)
    pad[:M] = X
    return pad.reshape(MT, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()


def from_x16(X16, M, MT=None):
    """Inverse of to_x16 for any tensor holding K/32 * MT KiB-fragments (tests / API exit only)."""
    MT = MT or mt_of(M)
    v = X16.reshape(-1, MT, 4, 16, 8)
    return v.permute(1, 3, 0, 2, 4).reshape(16 * MT, v.shape[0] * 32)[:M]


def skinny3(X16, M, W16, N, K, epi, n_splits=1, waves=4, slab=None, out=None, ldo=None, rt=None, ksplit=False, scale=None):
    """W16: bf16 W16 tensor, or (with `scale` fp32 [N]) fp8 bytes in W16F8 layout."""
    rt = rt if rt is not None else (2 if epi == 2 else 1)
    call("q3asr_skinny3_gemm", ptr(X16), M, ptr(W16), ptr(scale), N, K, n_splits, epi, waves, rt, 1 if ksplit else 0, ptr(slab), ptr(out),
         ldo if ldo is not None else N, sp())


def skinny3_norm(in_slab, in_n_slabs, in_stride, h_in, h_out, norm_w, eps, M, W16, N, K, epi, n_splits=1, slab=None, out=None, ldo=None,
                 rt=None, ksplit=False, scale=None):
    """skinny3 with the residual add + RMSNorm of its activation operand fused into the prologue (M <= 2, 4 waves)."""
    rt = rt if rt is not None else (2 if epi == 2 else 1)
    call("q3asr_skinny3_gemm_norm", ptr(in_slab), in_n_slabs, in_stride, ptr(h_in), ptr(h_out), ptr(norm_w), eps, M, ptr(W16), ptr(scale),
         N, K, n_splits, epi, 4, rt, 1 if ksplit else 0, ptr(slab), ptr(out), ldo if ldo is not None else N, sp())


def skinny4(X16, M, W16, N, K, epi, tpw, nks, waves, slab=None, out=None, ldo=None, scale=None):
    """q3asr_skinny4_gemm: `waves` waves per `tpw` weight tiles, `nks` k-steps per wave, K / (32 nks waves) slabs."""
    call("q3asr_skinny4_gemm", ptr(X16), M, ptr(W16), ptr(scale), N, K, epi, tpw, nks, waves, ptr(slab), ptr(out),
         ldo if ldo is not None else N, sp())


def skinny4_norm(in_slab, in_n_slabs, in_stride, h_in, h_out, norm_w, eps, M, W16, N, K, epi, tpw, nks, waves, slab=None, out=None, ldo=None,
                 scale=None):
    """skinny4 with the residual add + RMSNorm of its activation operand fused into the prologue (M <= 2)."""
    call("q3asr_skinny4_gemm_norm", ptr(in_slab), in_n_slabs, in_stride, ptr(h_in), ptr(h_out), ptr(norm_w), eps, M, ptr(W16), ptr(scale),
         N, K, epi, tpw, nks, waves, ptr(slab), ptr(out), ldo if ldo is not None else N, sp())


def pick_splits3(N: int, K: int, waves: int = 4) -> int:
    """Smallest split-K factor that gives >= 256 workgroups (one per CU), K slice >= 128."""
    tiles = (N // 16 + waves - 1) // waves
    valid = [s for s in (1, 2, 3, 4, 6, 8, 12, 16) if K % (32 * s) == 0 and K // s >= 128]
    for s in valid:
        if tiles * s >= 256:
            return s
    return valid[-1]


# Launch geometry of the decode-step GEMMs, chosen INSIDE the captured decode step (tools/tune_decode.py: coordinate
# descent over (n_splits, waves, tiles/wave, wg_ksplit), graph replays at batch 32, full-size weights): 1843 -> 1784 us
# per token against the table from isolated launches (profiles/r01_skinny_sweep.md) -- fewer slabs also shorten the
# kernels that consume them.  (N, K) -> (n_splits, waves, tiles/wave, wg_ksplit)
_TUNED = {
    (4096, 2048): (4, 4, 2, True),       # qkv
    (2048, 2048): (4, 4, 2, True),       # o
    (2048, 6144): (8, 4, 1, True),       # down
    (12288, 2048): (1, 4, 2, True),      # gate/up (SwiGLU epilogue, complete sums per workgroup)
    (151936, 2048): (1, 4, 2, True),     # tied head (5.3 TB/s)
}


# fp8 weight streams are half as long: the same tuner prefers two tiles per wave for `down` and four for the head
# (tune_decode.py 40 96 all fp8: 1961 -> 1918 us per token)
_TUNED_F8 = {
    (2048, 6144): (8, 4, 2, True),
    (151936, 2048): (1, 2, 4, True),
}


# skinny4 geometries (tiles per workgroup, k-steps per wave, waves) from tools/micro/sk4.hip: graphs of 28 dependent launches
# over distinct matrices, batch 32 and batch 1; K / (32 k-steps waves) slabs.  (N, K) -> geometry
_TUNED4 = {
    (4096, 2048): (2, 4, 8),             # qkv: 2 slabs (isolated: (2,2,8) with 4 slabs is 0.3 us faster; inside the captured step the
                                         # attention prologue then adds 4 slabs instead of 2: 251.5 -> 249.1 ms per request, batch 1 162.5 -> 160.8)
    (2048, 2048): (2, 2, 8),             # o
    (2048, 6144): (2, 3, 16),            # down: 4 slabs
    (12288, 2048): (4, 4, 16),           # gate/up (SwiGLU epilogue: one slab)
    (151936, 2048): (2, 8, 8),           # tied head
}
_TUNED4_F8 = {
    (4096, 2048): (2, 4, 8),
    (2048, 2048): (2, 2, 8),
    (2048, 6144): (2, 6, 8),
    (12288, 2048): (4, 4, 16),
    (151936, 2048): (4, 8, 8),
}
_SK4_BY_EPI = {0: ((2, 2, 8), (2, 4, 8), (2, 6, 8), (2, 3, 16)), 1: ((2, 2, 8), (2, 8, 8), (4, 8, 8)), 2: ((2, 2, 8), (4, 4, 16), (2, 4, 16))}


def gemm_geometry4(N: int, K: int, epi: int, fp8: bool = False, M: int = 32):
    """(tiles_per_wg, ksteps_per_wave, waves) for skinny4 on an (N, K) weight, or None when no instantiated geometry
    fits (the caller falls back to skinny3).  The choice never depends on M except for the tile count of the 16-wave
    SwiGLU geometry above 32 rows, which does not change the arithmetic."""
    def fits(g):
        tpw, nks, waves = g
        if g not in _SK4_BY_EPI[epi] or K % (32 * nks * waves) or (N // 16) % tpw or (fp8 and nks % 2):
            return False
        return epi == 0 or K == 32 * nks * waves
    g = (_TUNED4_F8 if fp8 else _TUNED4).get((N, K))
    if g is None or not fits(g):
        g = (2, 2, 8)
        if not fits(g):
            return None
    if M > 32 and g[2] * g[0] * mt_of(M) > 160:       # partial sums of 3 / 4 m-tiles must fit LDS: halve the tiles
        g = (g[0] // 2, g[1], g[2])
        if not fits(g):
            return None
    return g


def gemm_geometry(N: int, K: int, swiglu: bool = False, direct: bool = False, fp8: bool = False):
    """(n_splits, waves, tiles_per_wave, wg_ksplit) for skinny3 on an (N, K) weight."""
    g = (_TUNED_F8.get((N, K)) if fp8 else None) or _TUNED.get((N, K))
    if g is not None and (g[0] == 1 or not (swiglu or direct)):
        return g
    if swiglu:                                        # complete sums needed: waves split K inside the workgroup
        for waves in (4, 2, 1):
            if K % (32 * waves) == 0 and (N // 16) % 2 == 0:
                return (1, waves, 2, waves > 1)
    if direct:
        return (1, 4, 1, False)
    return (pick_splits3(N, K, 2), 2, 1, False)


# ---------------------------------------------------------------- fp8 weights (BASELINE.json configs[4])
def quantize_fp8_rows(W):
    """Per-output-row power-of-two scale + OCP e4m3 values.  Returns (q8 uint8 [N][K], scale fp32 [N],
    W_eff bf16 [N][K]) with W_eff = scale * fp8 EXACTLY representable in bf16, so the prefill GEMM (bf16
    W_eff) and the decode GEMM (fp8 + scale) use identical effective weights."""
    import torch
    Wf = W.float()
    amax = Wf.abs().amax(dim=1).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = (Wf / scale[:, None]).to(torch.float8_e4m3fn)
    w_eff = (q.float() * scale[:, None]).to(torch.bfloat16)
    return q.view(torch.uint8), scale.contiguous(), w_eff


def to_w16f8(q8):
    """uint8 [N][K] -> W16F8 [N/16][K/64][64 lanes][16 B] (lane = 16 g + li; byte = 8 half + j)."""
    N, K = q8.shape
    return q8.reshape(N // 16, 16, K // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous()
