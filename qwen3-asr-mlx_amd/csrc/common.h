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

// exact-erf GELU (MLX nn.gelu; encoder.py:118,273-275,320): x * Phi(x), Phi(-|x|) = erfc(|x| / sqrt 2) / 2 with erfc from
// Abramowitz & Stegun 7.1.26, erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z): |error| <= 1.5e-7 ABSOLUTE on
// erfc (about one fp32 ulp of an O(1) output).  The bound is not relative: deep in the negative tail, where erfc itself is
// small, the polynomial's relative error grows (about 0.3 % of the output at x = -4, where the output is ~1.3e-4); every
// caller rounds to bf16 (2^-8 = 0.4 % relative), so the tail stays inside one bf16 ulp and the absolute error is what the
// parity tests see.  The product form x * Phi(-|x|) avoids the cancellation 1 + erf(.) would add on top.  15 VALU instructions, two of them transcendental (one
// reciprocal, one exp2 with the scale folded into the argument): the epilogues of conv1 / conv2 / conv3 / fc1 are
// VALU-bound on this function (the Numerical Recipes erfcc form used before cost 21: fc1 149 -> see profiles/README.md).
__device__ __forceinline__ float gelu_erf(float x) {
    constexpr float C = 0.84932180028801904f;                 // sqrt(log2(e) / 2): u^2 = z^2 log2(e), z = |x| / sqrt 2
    constexpr float P = 0.3275911f / 1.2011224087864498f;     // p / sqrt(log2 e): 1 + p z = 1 + P u
    const float u = fabsf(x) * C;
    const float t = __builtin_amdgcn_rcpf(fmaf(P, u, 1.0f));   // v_rcp_f32 (1 ulp)
    float r = 0.5f * 1.061405429f;
    r = fmaf(r, t, 0.5f * -1.453152027f);
    r = fmaf(r, t, 0.5f * 1.421413741f);
    r = fmaf(r, t, 0.5f * -0.284496736f);
    r = fmaf(r, t, 0.5f * 0.254829592f);
    const float half_erfc = r * t * __builtin_amdgcn_exp2f(-(u * u));   // Phi(-|x|)
    const float xh = x * half_erfc;
    return x < 0.f ? xh : x - xh;
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
