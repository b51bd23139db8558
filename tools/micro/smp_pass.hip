// Micro-benchmark: one streaming pass of a 1024-thread workgroup over a 151936-logit bf16 row (40 rows, one WG each)
// with different per-element work, to see what bounds the sampler's passes.
// Build: hipcc --offload-arch=gfx950 -O3 smp_pass.hip -o smp_pass
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned key16(float x) {
    const unsigned u = __float_as_uint(x) >> 16;
    return (u & 0x8000u) ? (~u & 0xFFFFu) : (u | 0x8000u);
}
__device__ __forceinline__ float key16_to_float(unsigned k) {
    const unsigned u = (k & 0x8000u) ? (k & 0x7FFFu) : (~k & 0xFFFFu);
    return __uint_as_float(u << 16);
}

// MODE 0: load + max; 1: + key + u32 atomics 4096 bins; 2: u64 atomics with exp mass; 3: u32 atomics 65536/.. (window compare)
// 4: 4 private u32 hists   5: MODE 2 without the atomic (exp etc. only)
template <int MODE>
__global__ __launch_bounds__(1024) void pass_kernel(const bf16* logits, int V, float rt, float* out) {
    extern __shared__ unsigned char dyn[];
    unsigned long long* h64 = (unsigned long long*)dyn;
    unsigned* h32 = (unsigned*)dyn;
    const int tid = threadIdx.x, nchunks = V / 8;
    const bf16* row = logits + (long)blockIdx.x * V;
    for (int i = tid; i < 4 * 4096; i += 1024) h32[i] = 0;
    __syncthreads();
    float acc = 0.f;
    for (int c0 = tid; c0 < nchunks; c0 += 4096) {
        bf16x8 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = *(const bf16x8*)(row + 8 * min(c0 + u * 1024, nchunks - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u * 1024 >= nchunks) break;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (float)raw[u][e];
                if (MODE == 0) { acc = fmaxf(acc, v); continue; }
                v = (float)(bf16)(v * rt);
                const unsigned k = key16(v);
                if (MODE == 1) atomicAdd(&h32[k >> 4], 1u);
                if (MODE == 2 || MODE == 5) {
                    const float top = key16_to_float(k | 15u);
                    const float rel = v == top ? 1.0f : __expf(v - top);
                    const unsigned fx = (unsigned)(rel * 16777216.0f + 0.5f);
                    if (MODE == 2) atomicAdd(&h64[k >> 4], (1ull << 44) | fx); else acc += (float)fx;
                }
                if (MODE == 3) { const unsigned rel = k - 0x8000u; if (rel < 2048u) atomicAdd(&h32[rel], 1u); }
                if (MODE == 4) atomicAdd(&h32[(tid >> 8) * 4096 + (k >> 4)], 1u);
            }
        }
    }
    __syncthreads();
    if (MODE != 0 && MODE != 5) acc = (float)h32[tid];
    out[blockIdx.x * 1024 + tid] = acc;
}

template <int MODE>
static void run(const char* name, const bf16* d, int B, int V, float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)pass_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 4);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pass_kernel<MODE>, dim3(B), dim3(1024), 4 * 4096 * 4, 0, d, V, 1.0f / 0.7f, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pass_kernel<MODE>, dim3(B), dim3(1024), 4 * 4096 * 4, 0, d, V, 1.0f / 0.7f, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-40s %8.1f us per launch\n", name, ms * 1000 / 20);
}

int main() {
    const int B = 40, V = 151936;
    unsigned short* h = (unsigned short*)malloc((size_t)B * V * 2);
    srand(1);
    for (size_t i = 0; i < (size_t)B * V; ++i) {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
        float g = 0.9f * sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2);
        unsigned x; memcpy(&x, &g, 4);
        h[i] = (unsigned short)((x + 0x8000u) >> 16);
    }
    bf16* d; float* out;
    hipMalloc(&d, (size_t)B * V * 2); hipMalloc(&out, B * 1024 * 4);
    hipMemcpy(d, h, (size_t)B * V * 2, hipMemcpyHostToDevice);
    run<0>("load + max", d, B, V, out);
    run<5>("scale, key, exp mass (no atomic)", d, B, V, out);
    run<1>("u32 atomics, 4096 bins", d, B, V, out);
    run<2>("u64 atomics (count|mass), 4096 bins", d, B, V, out);
    run<3>("u32 atomics, 2048-key window", d, B, V, out);
    run<4>("u32 atomics, 4 private copies of 4096 bins", d, B, V, out);
    return 0;
}
