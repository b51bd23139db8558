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


def conv3x3s2(x, imgs, Hin, Win, Cin, w, Cout, bias, act, out):
    call("q3asr_conv3x3s2_bf16", ptr(x), imgs, Hin, Win, Cin, ptr(w), Cout, ptr(bias), act, ptr(out), 0, Cout, None, sp())


def layernorm(x, w, b, y, M, D, eps):
    call("q3asr_layernorm_f32_bf16", ptr(x), ptr(w), ptr(b), ptr(y), M, D, eps, sp())


def rmsnorm(x, w, y, M, D, eps, ldx=None, ldy=None, row_idx=None):
    call("q3asr_rmsnorm_bf16", ptr(x), ldx or D, ptr(w), ptr(y), ldy or D, ptr(row_idx), M, D, eps, sp())


def attn_fwd(Q, q_stride, K, V, kv_seg_off, kv_head_stride, kv_row_stride, O, o_stride, seg_q_start, seg_len, n_segs,
             max_len, n_heads, n_kv_heads, head_dim, causal, scale):
    call("q3asr_attn_fwd", Q, q_stride, K, V, ptr(kv_seg_off), kv_head_stride, kv_row_stride, ptr(O), o_stride,
         ptr(seg_q_start), ptr(seg_len), n_segs, max_len, n_heads, n_kv_heads, head_dim, 1 if causal else 0, scale, sp())


def skinny_gemm(X, M, W, N, K, n_splits, slab, out, ldx=None, ldo=None):
    """n_splits > 1: fp32 partials to `slab` [n_splits][M][N]; n_splits == 1: bf16 `out` [M][ldo]."""
    call("q3asr_skinny_gemm", ptr(X), ldx or K, M, ptr(W), K, N, K, n_splits, ptr(slab) if n_splits > 1 else None,
         ptr(out) if n_splits == 1 else None, ldo or N, sp())


def pick_splits(N: int, K: int, target_wgs: int = 512) -> int:
    """Smallest split-K factor (K slice a multiple of 256) that puts >= target_wgs workgroups on the
    256 CUs; the largest valid one if none reaches it."""
    tiles = (N + 63) // 64
    valid = [s for s in range(1, K // 256 + 1) if K % (s * 256) == 0]
    for s in valid:
        if tiles * s >= target_wgs:
            return s
    return valid[-1]
