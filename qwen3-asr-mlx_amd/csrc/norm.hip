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
