// Aggressor kernels for tools/lane_race4.py: each exercises one resource so that the one that disturbs a victim kernel on
// another stream can be named.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libaggr.so tools/micro/aggr.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void aggr_loads(const uint4* p, unsigned long long n, int iters) {
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it)
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) acc ^= p[i].x;
    asm volatile("" ::"v"(acc));
}

__global__ __launch_bounds__(256) void aggr_mfma(float* sink, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void aggr_valu(float* sink, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void aggr_lds(float* sink, int iters) {
    __shared__ float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = (float)i;
    __syncthreads();
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < 8192; i += 256) s += buf[(i * 17 + it) & 8191];
        __syncthreads();
        buf[threadIdx.x] = s;
        __syncthreads();
    }
    if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void aggr_stores(uint4* p, unsigned long long n, unsigned v) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) p[i] = uint4{v, v, v, v};
}

extern "C" int aggr_launch(int kind, void* buf, unsigned long long bytes, int grid, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
        case 0: hipLaunchKernelGGL(aggr_loads, dim3(grid), dim3(256), 0, s, (const uint4*)buf, bytes / 16, iters); break;
        case 1: hipLaunchKernelGGL(aggr_mfma, dim3(grid), dim3(256), 0, s, (float*)buf, iters); break;
        case 2: hipLaunchKernelGGL(aggr_valu, dim3(grid), dim3(256), 0, s, (float*)buf, iters); break;
        case 3: hipLaunchKernelGGL(aggr_lds, dim3(grid), dim3(256), 0, s, (float*)buf, iters); break;
        case 4: hipLaunchKernelGGL(aggr_stores, dim3(grid), dim3(256), 0, s, (uint4*)buf, bytes / 16, (unsigned)iters); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ---- victims: one VALU instruction kind each, results depend only on the thread index --------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void victim_kernel(float* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a = 0.999f - 1e-6f * (t & 1023), b = 0.001f * ((t & 63) + 1);
    f32x2 pa = {a, a - 1e-4f}, pb = {b, b * 0.5f};
    f32x2 p[4];
    float s[8];
    for (int i = 0; i < 4; ++i) p[i] = f32x2{0.1f * i, 0.2f * i};
    for (int i = 0; i < 8; ++i) s[i] = 0.1f * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pa), "v"(pb));
            if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
            if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));   // src1.lo for both halves
            if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p[i]) : "v"(pa), "v"(pb));      // src1.hi for both halves
            if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(p[i]) : "v"(pa), "v"(pb));      // src0.hi for the low result
            if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));     // src2.hi for the low result
            if (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
            if (KIND == 12) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2 op_sel:[0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));
            if (KIND == 13) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));   // src1 halves swapped
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
            if (KIND == 3) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(s[i]) : "v"(a));       // + 0.001
            if (KIND == 4) asm volatile("v_exp_f32 %0, %0\n\ts_nop 1\n\tv_sub_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
            if (KIND == 5) asm volatile("v_rcp_f32 %0, %0\n\ts_nop 1\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(a));
            if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(s[i]) : "v"(a), "v"(b));
        }
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += p[i][0] + p[i][1];
    for (int i = 0; i < 8; ++i) r += s[i];
    out[t] = r;
}

// Same question inside ONE kernel: waves 0-1 of every workgroup run the packed FMA with op_sel, waves 2-3 issue MFMAs
// (mfma != 0) or idle (mfma == 0).
__global__ __launch_bounds__(256) void same_kernel_mix(float* out, float* sink, int iters, int mfma) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if ((threadIdx.x >> 6) < 2) {
        float a = 0.999f - 1e-6f * (t & 1023), b = 0.001f * ((t & 63) + 1);
        f32x2 pa = {a, a - 1e-4f}, pb = {b, b * 0.5f};
        f32x2 p[4];
        for (int i = 0; i < 4; ++i) p[i] = f32x2{0.1f * i, 0.2f * i};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p[i]) : "v"(pa), "v"(pb));
        float r = 0.f;
        for (int i = 0; i < 4; ++i) r += p[i][0] + p[i][1];
        out[t] = r;
    } else {
        out[t] = 0.f;
        if (!mfma) return;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        float s2 = 0.f;
        for (int i = 0; i < 8; ++i) s2 += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s2 == 12345.678f) sink[0] = s2;
    }
}

extern "C" int same_kernel_launch(float* out, float* sink, int grid, int iters, int mfma, void* stream) {
    hipLaunchKernelGGL(same_kernel_mix, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, sink, iters, mfma);
    return (int)hipGetLastError();
}

extern "C" int victim_launch(int kind, float* out, int grid, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
        case 0: hipLaunchKernelGGL(victim_kernel<0>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 1: hipLaunchKernelGGL(victim_kernel<1>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 2: hipLaunchKernelGGL(victim_kernel<2>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 3: hipLaunchKernelGGL(victim_kernel<3>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 4: hipLaunchKernelGGL(victim_kernel<4>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 5: hipLaunchKernelGGL(victim_kernel<5>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 6: hipLaunchKernelGGL(victim_kernel<6>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 7: hipLaunchKernelGGL(victim_kernel<7>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 8: hipLaunchKernelGGL(victim_kernel<8>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 13: hipLaunchKernelGGL(victim_kernel<13>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 12: hipLaunchKernelGGL(victim_kernel<12>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 11: hipLaunchKernelGGL(victim_kernel<11>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 10: hipLaunchKernelGGL(victim_kernel<10>, dim3(grid), dim3(256), 0, s, out, iters); break;
        case 9: hipLaunchKernelGGL(victim_kernel<9>, dim3(grid), dim3(256), 0, s, out, iters); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ---- conv1 variants as victims (copy of the kernel under test in csrc/elementwise.hip, with one part replaced each) -------------
#include "../../qwen3-asr-mlx_amd/csrc/common.h"

template <int VAR>
__global__ __launch_bounds__(256) void conv1_variant(const float* __restrict__ mel, int Fmax, const int* __restrict__ chunk_b,
                                                     const int* __restrict__ chunk_f0, const float* __restrict__ w9,
                                                     const float* __restrict__ bias, bf16* __restrict__ out, int C1) {
    __shared__ float in_s[3 * 130];
    const int ci = blockIdx.y, ot = blockIdx.x;
    const int b = chunk_b[ci], f0 = chunk_f0[ci];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * 130; i += 256) {
        const int kt = i / 130, fp = i - kt * 130;
        const int t = 2 * ot - 1 + kt, f = fp - 1;
        float v = 0.f;
        if (t >= 0 && t < 100 && f >= 0 && f < 128 && f0 + t < Fmax) v = mel[((long)b * 128 + f) * Fmax + f0 + t];
        in_s[i] = v;
    }
    const int octs = C1 / 8, groups = 256 / octs;
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
    bf16* obase = out + ((long)(ci * 50 + ot) * 64) * C1 + oc * 8;
    for (int of = fg; of < 64; of += groups) {
        float x[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) x[kt * 3 + kf] = in_s[kt * 130 + 2 * of + kf];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = bb[e];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (VAR == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(w[e][tap]), "v"(x[tap]));      // no packed math
                else a = fmaf(w[e][tap], x[tap], a);
            }
            if (VAR == 1 || VAR == 3) o[e] = f2bf(a);                    // no GELU
            else if (VAR == 2) o[e] = f2bf(0.5f * a * (1.0f + erff(a * 0.70710678f)));
            else o[e] = f2bf(gelu_erf(a));
        }
        *reinterpret_cast<bf16x8*>(obase + (long)of * C1) = o;
    }
}

extern "C" int conv1_variant_launch(int var, const float* mel, int Fmax, const int* chunk_b, const int* chunk_f0, int n_chunks,
                                    const float* w9, const float* bias, void* out, int C1, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 g(50, n_chunks), bk(256);
    switch (var) {
        case 0: hipLaunchKernelGGL(conv1_variant<0>, g, bk, 0, s, mel, Fmax, chunk_b, chunk_f0, w9, bias, (bf16*)out, C1); break;
        case 1: hipLaunchKernelGGL(conv1_variant<1>, g, bk, 0, s, mel, Fmax, chunk_b, chunk_f0, w9, bias, (bf16*)out, C1); break;
        case 2: hipLaunchKernelGGL(conv1_variant<2>, g, bk, 0, s, mel, Fmax, chunk_b, chunk_f0, w9, bias, (bf16*)out, C1); break;
        case 3: hipLaunchKernelGGL(conv1_variant<3>, g, bk, 0, s, mel, Fmax, chunk_b, chunk_f0, w9, bias, (bf16*)out, C1); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
