// LayerNorm (encoder, MLX nn.LayerNorm eps 1e-5: encoder.py:98,102,189) and RMSNorm
// (decoder, MLX nn.RMSNorm: decoder.py:177,179,217).  One wave per row, statistics by
// wavefront shuffles, 16-byte loads.  HBM-bound: LayerNorm moves 4*D in + 2*D out bytes
// per row, RMSNorm 2*D + 2*D.
#include "common.h"

// x fp32 [M][D] -> y bf16 [M][D] (the next GEMM's A operand).  D % 256 == 0, D <= 2048.
template <int VPL>   // float4 vectors per lane: D = 256 * VPL
__global__ __launch_bounds__(256) void layernorm_f32_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, bf16* __restrict__ y,
                                                                 int M, float eps) {
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
    bf16x4* yr = reinterpret_cast<bf16x4*>(y + (long)row * D);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const f32x4 wv = reinterpret_cast<const f32x4*>(w)[lane + 64 * i];
        const f32x4 bv = reinterpret_cast<const f32x4*>(b)[lane + 64 * i];
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf((v[i][e] - mean) * rstd * wv[e] + bv[e]);
        yr[lane + 64 * i] = o;
    }
}

extern "C" int q3asr_layernorm_f32_bf16(const float* x, const float* w, const float* b, void* y, int M, int D, float eps,
                                        void* stream) {
    Q3_REQUIRE(x && w && b && y, "layernorm: null pointer");
    Q3_REQUIRE(M > 0 && (D == 256 || D == 512 || D == 1024 || D == 2048), "layernorm: M=%d D=%d unsupported (D in {256,512,1024,2048})", M, D);
    dim3 grid((M + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    switch (D) {
        case 256: hipLaunchKernelGGL(layernorm_f32_bf16_kernel<1>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
        case 512: hipLaunchKernelGGL(layernorm_f32_bf16_kernel<2>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
        case 1024: hipLaunchKernelGGL(layernorm_f32_bf16_kernel<4>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
        default: hipLaunchKernelGGL(layernorm_f32_bf16_kernel<8>, grid, blk, 0, s, x, w, b, (bf16*)y, M, eps); break;
    }
    return q3_check_launch("layernorm");
}

// x bf16 [M][D] -> y bf16 = bf16(x * rsqrt(mean(x^2)+eps) * w).  D % 512 == 0, D <= 4096.
// Optional row gather: row_idx[m] selects the source row (prefill's "last position only" head, K18).
template <int VPL>   // 8-element vectors per lane: D = 512 * VPL
__global__ __launch_bounds__(256) void rmsnorm_bf16_kernel(const bf16* __restrict__ x, long ldx, const bf16* __restrict__ w,
                                                           bf16* __restrict__ y, long ldy, const int* __restrict__ row_idx,
                                                           int M, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    constexpr int D = 512 * VPL;
    const long src = row_idx ? row_idx[row] : row;
    const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + src * ldx);
    bf16x8 v[VPL];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = xr[lane + 64 * i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = bf2f(v[i][e]);
            q = fmaf(f, f, q);
        }
    }
    const float r = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    bf16x8* yr = reinterpret_cast<bf16x8*>(y + (long)row * ldy);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bf16x8 wv = reinterpret_cast<const bf16x8*>(w)[lane + 64 * i];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(v[i][e]) * r * bf2f(wv[e]));
        yr[lane + 64 * i] = o;
    }
}

extern "C" int q3asr_rmsnorm_bf16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, const int32_t* row_idx,
                                  int M, int D, float eps, void* stream) {
    Q3_REQUIRE(x && w && y, "rmsnorm: null pointer");
    Q3_REQUIRE(M > 0 && D % 512 == 0 && D >= 512 && D <= 4096 && (D / 512 == 1 || D / 512 == 2 || D / 512 == 4 || D / 512 == 8),
               "rmsnorm: M=%d D=%d unsupported (D in {512,1024,2048,4096})", M, D);
    Q3_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= D && ldy >= D, "rmsnorm: bad leading dimension");
    dim3 grid((M + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16* xx = (const bf16*)x; const bf16* ww = (const bf16*)w; bf16* yy = (bf16*)y;
    switch (D / 512) {
        case 1: hipLaunchKernelGGL(rmsnorm_bf16_kernel<1>, grid, blk, 0, s, xx, (long)ldx, ww, yy, (long)ldy, row_idx, M, eps); break;
        case 2: hipLaunchKernelGGL(rmsnorm_bf16_kernel<2>, grid, blk, 0, s, xx, (long)ldx, ww, yy, (long)ldy, row_idx, M, eps); break;
        case 4: hipLaunchKernelGGL(rmsnorm_bf16_kernel<4>, grid, blk, 0, s, xx, (long)ldx, ww, yy, (long)ldy, row_idx, M, eps); break;
        default: hipLaunchKernelGGL(rmsnorm_bf16_kernel<8>, grid, blk, 0, s, xx, (long)ldx, ww, yy, (long)ldy, row_idx, M, eps); break;
    }
    return q3_check_launch("rmsnorm");
}


// ---------------------------------------------------------------------------
// Per-token fp8 quantisation of a GEMM's activation operand (BASELINE.json configs[4], q3asr_gemm_fp8).
// y8[m][k] = e4m3(v[m][k] / scale[m]),  scale[m] = max_k |v[m][k]| / 448  (1 for an all-zero row),  v = the bf16 values the
// bf16 path would feed the GEMM (RMSNorm output, decoder.py:189,194 / attention output / SwiGLU output).
// One wave per row; the row is read twice (the second time from L1 / L2).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

template <bool NORM>
__global__ __launch_bounds__(256) void rows_to_fp8_kernel(const bf16* __restrict__ x, long ldx, const bf16* __restrict__ w, float eps,
                                                          unsigned char* __restrict__ y8, long ldy, float* __restrict__ scale, int M, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + (long)row * ldx);
    const int nv = D >> 3;
    float r = 1.0f;
    if (NORM) {
        float q = 0.f;
        for (int i = lane; i < nv; i += 64) {
            const bf16x8 v = xr[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = bf2f(v[e]); q = fmaf(f, f, q); }
        }
        r = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    }
    float amax = 0.f;
    for (int i = lane; i < nv; i += 64) {
        const bf16x8 v = xr[i];
        bf16x8 wv;
        if (NORM) wv = reinterpret_cast<const bf16x8*>(w)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = NORM ? round_bf(bf2f(v[e]) * r * bf2f(wv[e])) : bf2f(v[e]);
            amax = fmaxf(amax, fabsf(f));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2* yr = reinterpret_cast<u32x2*>(y8 + (long)row * ldy);
    for (int i = lane; i < nv; i += 64) {
        const bf16x8 v = xr[i];
        bf16x8 wv;
        if (NORM) wv = reinterpret_cast<const bf16x8*>(w)[i];
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (NORM ? round_bf(bf2f(v[e]) * r * bf2f(wv[e])) : bf2f(v[e])) * inv;
        yr[i] = u32x2{pack4_fp8(f[0], f[1], f[2], f[3]), pack4_fp8(f[4], f[5], f[6], f[7])};
    }
}

static int rows_to_fp8_check(const char* who, const void* x, int64_t ldx, void* y8, int64_t ldy, float* scale, int M, int D) {
    Q3_REQUIRE(x && y8 && scale, "%s: null pointer", who);
    Q3_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && ldx >= D && ldx % 8 == 0 && ldy >= D && ldy % 8 == 0, "%s: M=%d D=%d ldx=%ld ldy=%ld", who, M, D, (long)ldx, (long)ldy);
    return Q3_OK;
}

extern "C" int q3asr_rmsnorm_bf16_fp8(const void* x, int64_t ldx, const void* w, void* y8, int64_t ldy, float* scale, int M, int D, float eps,
                                      void* stream) {
    if (int rc = rows_to_fp8_check("rmsnorm_bf16_fp8", x, ldx, y8, ldy, scale, M, D)) return rc;
    Q3_REQUIRE(w, "rmsnorm_bf16_fp8: null weight");
    hipLaunchKernelGGL(rows_to_fp8_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (long)ldx, (const bf16*)w, eps,
                       (unsigned char*)y8, (long)ldy, scale, M, D);
    return q3_check_launch("rmsnorm_bf16_fp8");
}

extern "C" int q3asr_quantize_rows_fp8(const void* x, int64_t ldx, void* y8, int64_t ldy, float* scale, int M, int D, void* stream) {
    if (int rc = rows_to_fp8_check("quantize_rows_fp8", x, ldx, y8, ldy, scale, M, D)) return rc;
    hipLaunchKernelGGL(rows_to_fp8_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (long)ldx, (const bf16*)nullptr, 0.f,
                       (unsigned char*)y8, (long)ldy, scale, M, D);
    return q3_check_launch("quantize_rows_fp8");
}
