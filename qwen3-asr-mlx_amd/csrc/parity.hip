// Reference-semantics ("parity") mode of the audio encoder: the reference keeps float32 activations through the whole
// encoder (audio.py:278 hands float32 mel to encoder.py:272-323 and nothing casts; MLX promotes f32 (+) bf16 -> f32), while
// the fast path rounds every MFMA operand to bf16.  Here an fp32 activation x is split into three bf16 pieces
//     x = hi + mid + lo,  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)        (3 x 8 = 24 significant bits)
// laid side by side along K, [M][3K] = [hi | mid | lo], and multiplied by [W | W | W] on the same bf16 MFMA GEMM: every
// bf16 x bf16 product is exact in fp32 and the accumulation is fp32, so the result is an fp32-accurate x . W^T at three
// times the MFMA work.  LayerNorm, GELU, softmax and the residual stream stay fp32; attention runs on fp32 q/k/v.
// Used for parity measurements (tests/test_parity_gpu.py), not for throughput.
#include "common.h"

__device__ __forceinline__ void split3(float x, bf16& hi, bf16& mid, bf16& lo) {
    hi = f2bf(x);
    const float r1 = x - bf2f(hi);            // exact
    mid = f2bf(r1);
    lo = f2bf(r1 - bf2f(mid));                 // exact difference, then the last rounding
}

// x fp32 [M][ldx] (K columns used) -> y bf16 [M][3K]
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long ldx, bf16* __restrict__ y, long M, int K) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;           // one 4-wide vector per thread
    const int vpr = K >> 2;
    if (v >= M * vpr) return;
    const long m = v / vpr;
    const int k = (int)(v - m * vpr) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + m * ldx + k);
    bf16x4 h, mi, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bf16 p, q, r;
        split3(a[e], p, q, r);
        h[e] = p; mi[e] = q; l[e] = r;
    }
    bf16* yr = y + m * 3L * K + k;
    *reinterpret_cast<bf16x4*>(yr) = h;
    *reinterpret_cast<bf16x4*>(yr + K) = mi;
    *reinterpret_cast<bf16x4*>(yr + 2L * K) = l;
}

extern "C" int q3asr_split3_f32(const float* x, int64_t ldx, void* y, int64_t M, int K, void* stream) {
    Q3_REQUIRE(x && y && M > 0 && K > 0 && K % 4 == 0 && ldx >= K && ldx % 4 == 0, "split3: M=%ld K=%d ldx=%ld", (long)M, K, (long)ldx);
    const long n = M * (K / 4);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (bf16*)y, (long)M, K);
    return q3_check_launch("split3");
}

// LayerNorm (encoder.py:112,117,319; eps 1e-5) with the fp32 result split into three bf16 pieces: y [M][3D].
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_f32_split3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ b, bf16* __restrict__ y, int M, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    constexpr int D = 256 * VPL;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long)row * D);
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = xr[lane + 64 * i];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q = fmaf(d, d, q);
        }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    bf16* yr = y + (long)row * 3 * D;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const f32x4 wv = reinterpret_cast<const f32x4*>(w)[lane + 64 * i];
        const f32x4 bv = reinterpret_cast<const f32x4*>(b)[lane + 64 * i];
        bf16x4 h, mi, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16 p, qq, r;
            split3((v[i][e] - mean) * rstd * wv[e] + bv[e], p, qq, r);
            h[e] = p; mi[e] = qq; l[e] = r;
        }
        const int c = 4 * (lane + 64 * i);
        *reinterpret_cast<bf16x4*>(yr + c) = h;
        *reinterpret_cast<bf16x4*>(yr + D + c) = mi;
        *reinterpret_cast<bf16x4*>(yr + 2 * D + c) = l;
    }
}

extern "C" int q3asr_layernorm_f32_split3(const float* x, const float* w, const float* b, void* y, int M, int D, float eps, void* stream) {
    Q3_REQUIRE(x && w && b && y, "layernorm_split3: null pointer");
    Q3_REQUIRE(M > 0 && (D == 256 || D == 512 || D == 1024 || D == 2048), "layernorm_split3: M=%d D=%d unsupported", M, D);
    dim3 grid((M + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    switch (D) {
        case 256: hipLaunchKernelGGL(layernorm_f32_split3_kernel<1>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
        case 512: hipLaunchKernelGGL(layernorm_f32_split3_kernel<2>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
        case 1024: hipLaunchKernelGGL(layernorm_f32_split3_kernel<4>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
        default: hipLaunchKernelGGL(layernorm_f32_split3_kernel<8>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
    }
    return q3_check_launch("layernorm_split3");
}

// Windowed bidirectional attention on fp32 q/k/v (encoder.py:78-85 under the block mask of :210-229), head_dim 64, windows of
// at most 128 tokens.  qkv fp32 [N][3d] (q | k | v), out fp32 [N][d].  One workgroup per (window, head): K and V of the
// window sit in LDS (read as broadcasts), a thread owns one query row: max pass, then exp / sum / P.V pass -- the plain
// softmax of the reference, everything fp32 (expf, not the fast intrinsic).
#define AF_MAXWIN 128
__global__ __launch_bounds__(128) void attn_fwd_f32_kernel(const float* __restrict__ qkv, long ld, int d, const int* __restrict__ seg_start,
                                                           const int* __restrict__ seg_len, float scale, float* __restrict__ out, long ldo) {
    __shared__ __attribute__((aligned(16))) float sK[AF_MAXWIN][64];
    __shared__ __attribute__((aligned(16))) float sV[AF_MAXWIN][64];
    const int seg = blockIdx.x, head = blockIdx.y, tid = threadIdx.x;
    const int len = min(seg_len[seg], AF_MAXWIN);
    const long row0 = seg_start[seg];
    for (int i = tid; i < len * 16; i += 128) {
        const int r = i >> 4, c = (i & 15) * 4;
        const float* src = qkv + (row0 + r) * ld + head * 64 + c;
        *reinterpret_cast<f32x4*>(&sK[r][c]) = *reinterpret_cast<const f32x4*>(src + d);
        *reinterpret_cast<f32x4*>(&sV[r][c]) = *reinterpret_cast<const f32x4*>(src + 2 * d);
    }
    __syncthreads();
    if (tid >= len) return;
    float q[64];
    const float* qr = qkv + (row0 + tid) * ld + head * 64;
#pragma unroll
    for (int e = 0; e < 64; e += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qr + e);
        q[e] = t[0]; q[e + 1] = t[1]; q[e + 2] = t[2]; q[e + 3] = t[3];
    }
    float m = -INFINITY;
    for (int j = 0; j < len; ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 64; ++e) s = fmaf(q[e], sK[j][e], s);
        m = fmaxf(m, s * scale);
    }
    float l = 0.f, o[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) o[e] = 0.f;
    for (int j = 0; j < len; ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 64; ++e) s = fmaf(q[e], sK[j][e], s);
        const float p = expf(s * scale - m);
        l += p;
#pragma unroll
        for (int e = 0; e < 64; ++e) o[e] = fmaf(p, sV[j][e], o[e]);
    }
    float* dst = out + (row0 + tid) * ldo + head * 64;
    const float inv = 1.0f / l;
#pragma unroll
    for (int e = 0; e < 64; e += 4) *reinterpret_cast<f32x4*>(dst + e) = f32x4{o[e] * inv, o[e + 1] * inv, o[e + 2] * inv, o[e + 3] * inv};
}

extern "C" int q3asr_attn_fwd_f32(const float* qkv, int64_t ld, int d_model, const int32_t* seg_start, const int32_t* seg_len, int n_segs,
                                  int max_len, int n_heads, float scale, float* out, int64_t ldo, void* stream) {
    Q3_REQUIRE(qkv && seg_start && seg_len && out, "attn_fwd_f32: null pointer");
    Q3_REQUIRE(n_segs > 0 && n_heads > 0 && d_model == 64 * n_heads && ld >= 3L * d_model && ld % 4 == 0 && ldo >= d_model && ldo % 4 == 0,
               "attn_fwd_f32: needs head_dim 64 and qkv rows of 3*d_model (d_model=%d heads=%d)", d_model, n_heads);
    Q3_REQUIRE(max_len > 0 && max_len <= AF_MAXWIN, "attn_fwd_f32: windows of at most %d tokens (got %d)", AF_MAXWIN, max_len);
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(n_segs, n_heads), dim3(128), 0, (hipStream_t)stream, qkv, (long)ld, d_model, seg_start,
                       seg_len, scale, out, (long)ldo);
    return q3_check_launch("attn_fwd_f32");
}
