// Shared device/host helpers for libq3asr_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define Q3_WAVE 64

// ---- status codes (include/q3asr.h)
#define Q3_OK 0
#define Q3_EBADARG (-1)
#define Q3_ELAUNCH (-2)

extern "C" void q3asr_set_error(const char* fmt, ...);
int q3_check_launch(const char* what);

#define Q3_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            q3asr_set_error(__VA_ARGS__);     \
            return Q3_EBADARG;                \
        }                                     \
    } while (0)

// ---- bf16 helpers -------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }          // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
__device__ __forceinline__ float round_bf(float x) { return (float)(bf16)x; }

// exact-erf GELU (MLX nn.gelu; encoder.py:118,273-275,320): x * Phi(x) with Phi from the complementary error function in
// Chebyshev-fitted form erfc(z) = t exp(-z^2 + P(t)), t = 1 / (1 + z/2), z >= 0 -- fractional error < 1.2e-7 everywhere
// (Numerical Recipes erfcc), so the negative tail keeps relative accuracy, which 1 + erf(.) would lose; branch-free, one
// reciprocal and one exponential (the library erff costs about three times as many instructions and the GELU epilogues of
// conv1 / conv2 / conv3 / fc1 are VALU-bound).  Outputs are rounded to bf16 by every caller.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.5f * z);            // v_rcp_f32 (1 ulp): inside the fit's 1.2e-7
    float p = 0.17087277f;
    p = fmaf(p, t, -0.82215223f);
    p = fmaf(p, t, 1.48851587f);
    p = fmaf(p, t, -1.13520398f);
    p = fmaf(p, t, 0.27886807f);
    p = fmaf(p, t, -0.18628806f);
    p = fmaf(p, t, 0.09678418f);
    p = fmaf(p, t, 0.37409196f);
    p = fmaf(p, t, 1.00002368f);
    p = fmaf(p, t, -1.26551223f);
    const float half_erfc = 0.5f * t * __expf(fmaf(-z, z, p));           // Phi(-|x|)
    return x * (x < 0.f ? half_erfc : 1.0f - half_erfc);
}

// ---- wave reductions (wave = 64 lanes) -----------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD
// (observed round-robin dispatch); give each XCD a contiguous chunk of tiles so
// neighbouring tiles hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, x = bid & 7u;
    const unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
