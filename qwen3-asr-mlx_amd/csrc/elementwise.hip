// Small fused kernels around the GEMMs: conv stem layer 1 (direct), prompt
// embedding merge, per-head QK-RMSNorm + RoPE + KV-cache append.  All HBM-bound.
#include "common.h"

// ---------------------------------------------------------------------------
// conv2d1: Conv2d(1 -> C1, k3 s2 p1) + exact GELU over 100-frame mel chunks
// (encoder.py:148-154,254-273).  Input is the normalised log-mel [B][128][Fmax]
// fp32; chunk ci covers frames [f0, f0+100) of utterance chunk_b[ci], zero beyond
// the utterance (encoder.py:262-266; the normalise kernel wrote those zeros).
// Output layout is TIME-major NHWC: act[ci][ot 50][of 64][C1] bf16, so that after
// three convs a token's 16x480 features are one contiguous GEMM row (the
// reference's transpose at encoder.py:278 is folded into conv_out's weight order).
// K = 9 per output: direct fp32 FMAs; traffic = 2*C1 bytes written per output pixel.
// ---------------------------------------------------------------------------
#define C1_TIN 100
#define C1_FIN 128

template <bool F32OUT>      // F32OUT: reference-semantics mode (csrc/parity.hip) keeps the activation in float32
__global__ __launch_bounds__(256) void conv1_gelu_kernel(const float* __restrict__ mel, int Fmax,
                                                         const int* __restrict__ chunk_b, const int* __restrict__ chunk_f0,
                                                         const float* __restrict__ w9 /*[C1][kt][kf]*/,
                                                         const float* __restrict__ bias, void* __restrict__ out_, int C1) {
    // One workgroup per (chunk, output time step): 64 output frequencies x C1 channels.  A thread owns one
    // channel octet for the whole row: its 72 weights and 8 biases live in registers, the 3 input rows in
    // LDS are read as broadcasts (all lanes of a wave work on the same frequency at a time).
    __shared__ float in_s[3 * 130];      // [kt][f + 1], f = -1 .. 128
    const int ci = blockIdx.y, ot = blockIdx.x;
    const int TOUT = (C1_TIN - 1) / 2 + 1, FOUT = (C1_FIN - 1) / 2 + 1;
    const int b = chunk_b[ci], f0 = chunk_f0[ci];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * 130; i += 256) {
        const int kt = i / 130, fp = i - kt * 130;
        const int t = 2 * ot - 1 + kt, f = fp - 1;
        float v = 0.f;
        if (t >= 0 && t < C1_TIN && f >= 0 && f < C1_FIN && f0 + t < Fmax)
            v = mel[((long)b * C1_FIN + f) * Fmax + f0 + t];
        in_s[i] = v;
    }
    const int octs = C1 / 8;
    const int groups = 256 / octs;                          // frequency lanes processed concurrently (4 for C1 = 480)
    const int oc = tid % octs, fg = tid / octs;
    float w[8][9], bb[8];
    if (fg < groups) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bb[e] = bias[oc * 8 + e];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) w[e][tap] = w9[(oc * 8 + e) * 9 + tap];
        }
    }
    __syncthreads();
    if (fg >= groups) return;
    const long obase_i = ((long)(ci * TOUT + ot) * FOUT) * C1 + oc * 8;
    bf16* obase = (bf16*)out_ + obase_i;
    float* obase32 = (float*)out_ + obase_i;
    for (int of = fg; of < FOUT; of += groups) {
        float x[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) x[kt * 3 + kf] = in_s[kt * 130 + 2 * of + kf];   // f = 2*of-1+kf -> index +1
        bf16x8 o;
        float o32[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = bb[e];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a = fmaf(w[e][tap], x[tap], a);
            o32[e] = gelu_erf(a);
            o[e] = f2bf(o32[e]);
        }
        if constexpr (F32OUT) {
            *reinterpret_cast<f32x4*>(obase32 + (long)of * C1) = f32x4{o32[0], o32[1], o32[2], o32[3]};
            *reinterpret_cast<f32x4*>(obase32 + (long)of * C1 + 4) = f32x4{o32[4], o32[5], o32[6], o32[7]};
        } else {
            *reinterpret_cast<bf16x8*>(obase + (long)of * C1) = o;
        }
    }
}

extern "C" int q3asr_conv1_gelu(const float* mel, int Fmax, const int32_t* chunk_b, const int32_t* chunk_f0, int n_chunks,
                                const float* w9, const float* bias, void* out, int C1, void* stream) {
    Q3_REQUIRE(mel && chunk_b && chunk_f0 && w9 && bias && out, "conv1_gelu: null pointer");
    Q3_REQUIRE(n_chunks > 0 && Fmax > 0 && C1 > 0 && C1 % 8 == 0 && C1 <= 2048, "conv1_gelu: bad sizes (n_chunks=%d C1=%d, C1 <= 2048)", n_chunks, C1);
    hipLaunchKernelGGL(conv1_gelu_kernel<false>, dim3(50, n_chunks), dim3(256), 0, (hipStream_t)stream, mel, Fmax, chunk_b,
                       chunk_f0, w9, bias, out, C1);
    return q3_check_launch("conv1_gelu");
}

// the same with float32 output [n_chunks][50][64][C1] (reference-semantics mode)
extern "C" int q3asr_conv1_gelu_f32(const float* mel, int Fmax, const int32_t* chunk_b, const int32_t* chunk_f0, int n_chunks,
                                    const float* w9, const float* bias, float* out, int C1, void* stream) {
    Q3_REQUIRE(mel && chunk_b && chunk_f0 && w9 && bias && out, "conv1_gelu_f32: null pointer");
    Q3_REQUIRE(n_chunks > 0 && Fmax > 0 && C1 > 0 && C1 % 8 == 0 && C1 <= 2048, "conv1_gelu_f32: bad sizes (n_chunks=%d C1=%d, C1 <= 2048)", n_chunks, C1);
    hipLaunchKernelGGL(conv1_gelu_kernel<true>, dim3(50, n_chunks), dim3(256), 0, (hipStream_t)stream, mel, Fmax, chunk_b,
                       chunk_f0, w9, bias, (void*)out, C1);
    return q3_check_launch("conv1_gelu_f32");
}

// ---------------------------------------------------------------------------
// prepare_inputs (generate.py:20-81): out[m] = audio_src[m] >= 0 ? bf16(enc[audio_src[m]]) : E[ids[m]].
// The index vectors live in device memory the C-ABI cannot validate: an encoder row >= n_enc or a token id outside [0, V)
// yields a zero row instead of an out-of-bounds read.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_merge_kernel(const int* __restrict__ ids, const int* __restrict__ audio_src,
                                                          const bf16* __restrict__ E, const float* __restrict__ enc,
                                                          bf16* __restrict__ out, int H, int V, int n_enc) {
    const long m = blockIdx.x;
    const int a = audio_src[m];
    bf16* o = out + m * H;
    if ((a >= 0 && a >= n_enc) || (a < 0 && (unsigned)ids[m] >= (unsigned)V)) {          // uniform per workgroup
        for (int i = threadIdx.x * 8; i < H; i += 2048) *reinterpret_cast<uint4*>(o + i) = uint4{0u, 0u, 0u, 0u};
        return;
    }
    if (a >= 0) {
        const float* src = enc + (long)a * H;
        for (int i = threadIdx.x * 4; i < H; i += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
            bf16x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = f2bf(v[e]);
            *reinterpret_cast<bf16x4*>(o + i) = r;
        }
    } else {
        const bf16* src = E + (long)ids[m] * H;
        for (int i = threadIdx.x * 8; i < H; i += 2048)
            *reinterpret_cast<uint4*>(o + i) = *reinterpret_cast<const uint4*>(src + i);
    }
}

extern "C" int q3asr_embed_merge(const int32_t* ids, const int32_t* audio_src, const void* E, int V, const float* enc, int n_enc,
                                 void* out, int M, int H, void* stream) {
    Q3_REQUIRE(ids && audio_src && E && out, "embed_merge: null pointer");
    Q3_REQUIRE(M > 0 && H % 8 == 0 && V > 0 && n_enc >= 0 && (enc || n_enc == 0), "embed_merge: M=%d H=%d V=%d n_enc=%d", M, H, V, n_enc);
    hipLaunchKernelGGL(embed_merge_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, ids, audio_src, (const bf16*)E, enc,
                       (bf16*)out, H, V, n_enc);
    return q3_check_launch("embed_merge");
}

// ---------------------------------------------------------------------------
// Per-head q/k RMSNorm(128) -> RoPE (half-split pairs (i, i+64), base theta) -> q out,
// k/v appended in place to the KV cache (decoder.py:140-156, KVCache.update :32-64 made
// O(1)).  One wave per (token row, head slot); lane i owns the rotation pair (i, i+64).
// Input is either the bf16 qkv GEMM output or S fp32 split-K slabs (decode path), whose
// sum is first rounded to bf16 (the q_proj output dtype).
// cache layout: [B][n_kv][Lmax][128] bf16, K and V separately (one layer).
// ---------------------------------------------------------------------------
template <bool SLAB>
__global__ __launch_bounds__(256) void qk_norm_rope_kv_kernel(const void* __restrict__ qkv, long ld, int n_slabs, long slab_stride,
                                                              const int* __restrict__ row_b, const int* __restrict__ row_pos,
                                                              const bf16* __restrict__ qn_w, const bf16* __restrict__ kn_w,
                                                              const float* __restrict__ inv_freq, float eps,
                                                              bf16* __restrict__ q_out, bf16* __restrict__ k_cache,
                                                              bf16* __restrict__ v_cache, int M, int nh, int nkv, int Lmax) {
    const int slots = nh + 2 * nkv;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long)M * slots) return;
    const int m = (int)(w / slots), hs = (int)(w - (long)m * slots);
    const int lane = threadIdx.x & 63;
    float x1, x2;
    if (SLAB) {
        const float* base = (const float*)qkv + (long)m * ld + hs * 128;
        float a = 0.f, c = 0.f;
        for (int s = 0; s < n_slabs; ++s) {
            a += base[s * slab_stride + lane];
            c += base[s * slab_stride + lane + 64];
        }
        x1 = round_bf(a);
        x2 = round_bf(c);
    } else {
        const bf16* base = (const bf16*)qkv + (long)m * ld + hs * 128;
        x1 = bf2f(base[lane]);
        x2 = bf2f(base[lane + 64]);
    }
    const int b = row_b[m], pos = row_pos[m];
    // index vectors come from device memory the C-ABI cannot validate: a position outside the cache (or a negative row)
    // drops the cache write instead of storing out of bounds (r1 fault record: a freed index vector held positions where
    // rows were expected)
    const bool in_cache = b >= 0 && (unsigned)pos < (unsigned)Lmax;
    if (hs >= nh && !in_cache) return;
    if (hs >= nh + nkv) {                                   // v: plain append
        bf16* dst = v_cache + (((long)b * nkv + (hs - nh - nkv)) * Lmax + pos) * 128;
        dst[lane] = f2bf(x1);
        dst[lane + 64] = f2bf(x2);
        return;
    }
    const bf16* nw = hs < nh ? qn_w : kn_w;
    const float ss = wave_sum(x1 * x1 + x2 * x2) * (1.0f / 128.0f);
    const float r = rsqrtf(ss + eps);
    const float y1 = round_bf(x1 * r * bf2f(nw[lane]));
    const float y2 = round_bf(x2 * r * bf2f(nw[lane + 64]));
    const float ang = (float)pos * inv_freq[lane];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    const bf16 o1 = f2bf(y1 * cs - y2 * sn);
    const bf16 o2 = f2bf(y1 * sn + y2 * cs);
    bf16* dst = hs < nh ? q_out + (long)m * nh * 128 + hs * 128
                        : k_cache + (((long)b * nkv + (hs - nh)) * Lmax + pos) * 128;
    dst[lane] = o1;
    dst[lane + 64] = o2;
}

// Prefill form of the same op on bf16 rows: one workgroup per token, 8 lanes per head with 16-byte accesses, the
// token's 64 (cos, sin) pairs computed once in LDS and shared by its heads.  Same arithmetic and roundings as the
// wave-per-head kernel above (bit-identical results).
__global__ __launch_bounds__(256) void qk_norm_rope_kv_rows_kernel(const bf16* __restrict__ qkv, long ld,
                                                                   const int* __restrict__ row_b, const int* __restrict__ row_pos,
                                                                   const bf16* __restrict__ qn_w, const bf16* __restrict__ kn_w,
                                                                   const float* __restrict__ inv_freq, float eps,
                                                                   bf16* __restrict__ q_out, bf16* __restrict__ k_cache,
                                                                   bf16* __restrict__ v_cache, int nh, int nkv, int Lmax) {
    __shared__ float s_cs[64], s_sn[64];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int b = row_b[m], pos = row_pos[m];
    if (tid < 64) {
        float sn, cs;
        sincosf((float)pos * inv_freq[tid], &sn, &cs);
        s_cs[tid] = cs;
        s_sn[tid] = sn;
    }
    __syncthreads();
    const int slots = nh + 2 * nkv, sub = tid & 7;
    for (int hs = tid >> 3; hs < slots; hs += 32) {
        const bf16* src = qkv + (long)m * ld + hs * 128 + 8 * sub;
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(src), v2 = *reinterpret_cast<const bf16x8*>(src + 64);
        if (hs >= nh && !(b >= 0 && (unsigned)pos < (unsigned)Lmax)) continue;   // outside the cache: no store (see above)
        if (hs >= nh + nkv) {                               // v: plain append
            bf16* dst = v_cache + (((long)b * nkv + (hs - nh - nkv)) * Lmax + pos) * 128 + 8 * sub;
            *reinterpret_cast<bf16x8*>(dst) = v1;
            *reinterpret_cast<bf16x8*>(dst + 64) = v2;
            continue;
        }
        float x1[8], x2[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x1[e] = bf2f(v1[e]);
            x2[e] = bf2f(v2[e]);
        }
        // the wave-per-head kernel sums x1^2 + x2^2 per lane (pair j, j + 64) and then across lanes with the xor tree
        // 32, 16, 8, 4, 2, 1 over j: reproduce that order exactly
        float pj[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pj[e] = x1[e] * x1[e] + x2[e] * x2[e];
        // lanes j = 8 sub + e: tree level 32/16/8 pairs lanes that differ in sub bits 2/1/0; levels 4/2/1 pair e bits
#pragma unroll
        for (int o = 4; o >= 1; o >>= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) pj[e] += __shfl_xor(pj[e], o, 64);
        pj[0] += pj[4]; pj[1] += pj[5]; pj[2] += pj[6]; pj[3] += pj[7];
        pj[0] += pj[2]; pj[1] += pj[3];
        ss = (pj[0] + pj[1]) * (1.0f / 128.0f);
        const float r = rsqrtf(ss + eps);
        const bf16* nw = (hs < nh ? qn_w : kn_w) + 8 * sub;
        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(nw), w2 = *reinterpret_cast<const bf16x8*>(nw + 64);
        bf16x8 o1, o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y1 = round_bf(x1[e] * r * bf2f(w1[e])), y2 = round_bf(x2[e] * r * bf2f(w2[e]));
            const float cs = s_cs[8 * sub + e], sn = s_sn[8 * sub + e];
            o1[e] = f2bf(y1 * cs - y2 * sn);
            o2[e] = f2bf(y1 * sn + y2 * cs);
        }
        bf16* dst = (hs < nh ? q_out + (long)m * nh * 128 + hs * 128
                             : k_cache + (((long)b * nkv + (hs - nh)) * Lmax + pos) * 128) + 8 * sub;
        *reinterpret_cast<bf16x8*>(dst) = o1;
        *reinterpret_cast<bf16x8*>(dst + 64) = o2;
    }
}

extern "C" int q3asr_qk_norm_rope_kv(const void* qkv, int is_slab, int64_t ld, int n_slabs, int64_t slab_stride,
                                     const int32_t* row_b, const int32_t* row_pos, const void* qn_w, const void* kn_w,
                                     const float* inv_freq, float eps, void* q_out, void* k_cache, void* v_cache, int M,
                                     int nh, int nkv, int head_dim, int Lmax, void* stream) {
    Q3_REQUIRE(qkv && row_b && row_pos && qn_w && kn_w && inv_freq && q_out && k_cache && v_cache, "qk_norm_rope_kv: null pointer");
    Q3_REQUIRE(head_dim == 128, "qk_norm_rope_kv: head_dim=%d unsupported (128 only)", head_dim);
    Q3_REQUIRE(M > 0 && nh > 0 && nkv > 0 && Lmax > 0 && ld >= (int64_t)(nh + 2 * nkv) * 128, "qk_norm_rope_kv: bad sizes");
    const long waves = (long)M * (nh + 2 * nkv);
    dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (is_slab) {
        Q3_REQUIRE(n_slabs > 0, "qk_norm_rope_kv: n_slabs=%d", n_slabs);
        hipLaunchKernelGGL(qk_norm_rope_kv_kernel<true>, grid, blk, 0, s, qkv, (long)ld, n_slabs, (long)slab_stride, row_b, row_pos,
                           (const bf16*)qn_w, (const bf16*)kn_w, inv_freq, eps, (bf16*)q_out, (bf16*)k_cache, (bf16*)v_cache, M, nh, nkv, Lmax);
    } else if (ld % 8 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)q_out & 15) == 0 && ((uintptr_t)k_cache & 15) == 0 &&
               ((uintptr_t)v_cache & 15) == 0 && ((uintptr_t)qn_w & 15) == 0 && ((uintptr_t)kn_w & 15) == 0) {
        hipLaunchKernelGGL(qk_norm_rope_kv_rows_kernel, dim3(M), blk, 0, s, (const bf16*)qkv, (long)ld, row_b, row_pos,
                           (const bf16*)qn_w, (const bf16*)kn_w, inv_freq, eps, (bf16*)q_out, (bf16*)k_cache, (bf16*)v_cache, nh, nkv, Lmax);
    } else {
        hipLaunchKernelGGL(qk_norm_rope_kv_kernel<false>, grid, blk, 0, s, qkv, (long)ld, 0, 0L, row_b, row_pos,
                           (const bf16*)qn_w, (const bf16*)kn_w, inv_freq, eps, (bf16*)q_out, (bf16*)k_cache, (bf16*)v_cache, M, nh, nkv, Lmax);
    }
    return q3_check_launch("qk_norm_rope_kv");
}
