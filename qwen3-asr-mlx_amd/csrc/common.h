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
// A/B switches of launch forms that do not change results (q3asr_set_option, include/q3asr.h); read by the launchers only
extern int g_q3_option[8];
#define Q3ASR_OPT_HEAD_XCD_MAP 1      // the ids of include/q3asr.h
#define Q3ASR_OPT_HEAD_STREAM 2
#define Q3ASR_OPT_ATTN_XCD_ORDER 3
#define Q3ASR_OPT_SAMPLER_RESIDENT 4
#define Q3ASR_OPT_ATTN_HALF_WG 5
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

// SwiGLU at the reference's rounding points.  decoder.py:97 is nn.silu(gate_proj(x)) * up_proj(x) on bf16 arrays, and under
// MLX's per-op dtype rule that is three bf16 results: sigmoid(g) -> bf16, g * sigmoid(g) -> bf16 (nn.silu = x * mx.sigmoid(x)),
// then the product with up -> bf16.  Rounds 1-4 rounded once at the end (a sub-ulp deviation the round-4 review asked to
// remove); every SwiGLU epilogue of the library (8-phase / 128^2 / 64^2 GEMM, skinny3, skinny4; bf16 and fp8) calls this.
// Arguments are the bf16-rounded projection outputs.  v_rcp_f32 (1 ulp) instead of an IEEE division: the quotient is rounded
// to 8 significant bits next, and the division sequence cost ~8 VALU per element of a VALU-bound epilogue.
__device__ __forceinline__ bf16 swiglu_bf(float gate, float up) {
    const float s = round_bf(__builtin_amdgcn_rcpf(1.0f + __expf(-gate)));
    const float t = round_bf(gate * s);
    return f2bf(t * up);
}

// ---- wave reductions (wave = 64 lanes) -----------------------------------
// The butterfly v = op(v, v[lane ^ o]), o = 32, 16, 8, 4, 2, 1, of rounds 1-3 -- the same partners in the same order, so the
// same bits for the commutative ops used here -- without the LDS crossbar: hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS
// instruction and an s_waitcnt per step: ~600 cycles for a 6-step reduction on the dependent chain of every norm and
// attention prologue).  gfx950 has v_permlane32_swap / v_permlane16_swap (row exchanges: with both operands the same
// value, result 0 (op) result 1 = op(v, v[lane ^ 32]) resp. v[lane ^ 16] in every lane) and DPP covers the steps inside a
// row of 16: row_ror:8 = lane ^ 8, row_half_mirror then quad_perm [3,2,1,0] = lane ^ 7 ^ 3 = lane ^ 4, quad_perm [2,3,0,1]
// = lane ^ 2, [1,0,3,2] = lane ^ 1.  Round 4.
// CONTRACT: all 64 lanes active (full wave, convergent control flow).  The DPP steps run with bound_ctrl = false and old = 0,
// and the permlane swaps exchange whole rows: an inactive lane would contribute 0 (wave_max would return max(v, 0)) or stale
// register contents, silently.  Every call site is wave-uniform (block sizes are multiples of 64, no lane-dependent branch
// around a reduction); a kernel that needs a partial-wave reduction must mask its VALUES (0 / -inf), not its lanes.
#define Q3_DPP(x, ctrl) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), (ctrl), 0xf, 0xf, false))
__device__ __forceinline__ float lane_xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane_xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = lane_xor32_sum(v);
    v = lane_xor16_sum(v);
    v += Q3_DPP(v, 0x128);                                  // row_ror:8
    v += Q3_DPP(Q3_DPP(v, 0x141), 0x1b);                    // row_half_mirror, quad_perm [3,2,1,0]
    v += Q3_DPP(v, 0x4e);                                   // quad_perm [2,3,0,1]
    v += Q3_DPP(v, 0xb1);                                   // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = lane_xor32_max(v);
    v = lane_xor16_max(v);
    v = fmaxf(v, Q3_DPP(v, 0x128));
    v = fmaxf(v, Q3_DPP(Q3_DPP(v, 0x141), 0x1b));
    v = fmaxf(v, Q3_DPP(v, 0x4e));
    v = fmaxf(v, Q3_DPP(v, 0xb1));
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
