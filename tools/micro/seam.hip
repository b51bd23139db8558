// Micro-benchmark: the residual+RMSNorm -> GEMM seam of the decode step kept INSIDE the consuming launch, on the skinny4 body.
//
// A decode layer without its attention launch, batch 32, 28 layers of distinct weights (2.8 GB, nothing cached):
//   arm A (shipped): o GEMM -> resnorm -> gate/up GEMM (SwiGLU) -> down GEMM -> resnorm -> qkv GEMM        = 6 launches
//   arm B (fused)  : o GEMM -> [producers | gate/up GEMM]       -> down GEMM -> [producers | qkv GEMM]     = 4 launches
// In arm B the first NP workgroups of the consuming launch are PRODUCERS: four waves per row run the arithmetic of
// slab_resnorm_x16_kernel thread for thread (same bits), store the X16 image write-through (sc1), drain, and add to a
// counter; every GEMM workgroup puts ALL of its weight loads in flight first (the skinny4 rule), then one lane polls the
// counter, the workgroup barriers, and the X fragments are read with sc1 loads (L2-served, the L1 is bypassed).  The
// producers are workgroups of their own so that no wave has weight loads queued in front of its publish (vmcnt retires in
// order: a producer that also streamed weights could not drain its stores before its weights had landed).
// Round 2 measured this seam on skinny3 as a loss; what differs here: every weight byte of the launch
// is in flight before the poll, the producers delay no weight stream, and the row arithmetic needs one load round trip.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/seam.hip -o tools/micro/seam
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
__device__ __forceinline__ float round_bf(float x) { return (float)(bf16)x; }
__device__ __host__ __forceinline__ long x16_index(int m, int k, int MT) {
    return ((long)(k >> 5) * MT + (m >> 4)) * 512 + ((((k & 31) >> 3) << 4) + (m & 15)) * 8 + (k & 7);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void fill_kernel(bf16* p, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = f2bf(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
}

constexpr int HID = 2048;
constexpr unsigned GRAN_TAG = 0x5EA30001u;                 // granule buffers are zeroed before every replay
__device__ __forceinline__ auto make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, 0x7fffffff, 0x00020000);
}

// ---- the row arithmetic of slab_resnorm_x16_kernel<8, SS> (csrc/decode.hip), for the 256 threads t of one row m.
// XS: 0 plain X16 store, 1 write-through (sc1) store.
template <int SS, int XS>
__device__ __forceinline__ void resnorm_row(int m, int t, float* red, const float* __restrict__ slab, long slab_stride, const bf16* h_in,
                                            bf16* h_out, const bf16* __restrict__ w, bf16* x16, int MT, float eps, bool sync_all) {
    const long c0 = (long)m * HID + 8 * t;
    f32x4 p0[SS], p1[SS];
#pragma unroll
    for (int q = 0; q < SS; ++q) {
        p0[q] = *reinterpret_cast<const f32x4*>(slab + q * slab_stride + c0);
        p1[q] = *reinterpret_cast<const f32x4*>(slab + q * slab_stride + c0 + 4);
    }
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + 8 * t);
    const bf16x8 hi = *reinterpret_cast<const bf16x8*>(h_in + c0);
    f32x4 a0 = p0[0], a1 = p1[0];
#pragma unroll
    for (int q = 1; q < SS; ++q) { a0 += p0[q]; a1 += p1[q]; }
    float hv[8], q2 = 0.f;
    bf16x8 ho;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = round_bf(round_bf(e < 4 ? a0[e] : a1[e - 4]) + bf2f(hi[e]));
        hv[e] = v;
        q2 = fmaf(v, v, q2);
        ho[e] = f2bf(v);
    }
    *reinterpret_cast<bf16x8*>(h_out + c0) = ho;
    q2 = wave_sum(q2);
    if ((t & 63) == 0) red[t >> 6] = q2;
    __syncthreads();
    const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) * (1.0f / HID) + eps);
    bf16x8 xo;
#pragma unroll
    for (int e = 0; e < 8; ++e) xo[e] = f2bf(hv[e] * r * bf2f(wv[e]));
    bf16* dst = x16 + x16_index(m, 8 * t, MT);
    if constexpr (XS == 0) {
        *reinterpret_cast<bf16x8*>(dst) = xo;
    } else if constexpr (XS == 1) {
        (void)dst;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xo), make_rsrc(x16), (int)(x16_index(m, 8 * t, MT) * 2), 0, 16);   // aux 16 = sc1
    } else {
        // granules: 8 bytes = {tag, two bf16}; the fragment's four granules are 32 contiguous bytes, written by two 16-byte
        // write-through stores (each 8-byte half lands whole); the data is its own flag
        (void)dst;
        const u32x4 v = __builtin_bit_cast(u32x4, xo);
        const auto r = make_rsrc(x16);
        const int off = (int)(x16_index(m, 8 * t, MT) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{GRAN_TAG, v[0], GRAN_TAG, v[1]}, r, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{GRAN_TAG, v[2], GRAN_TAG, v[3]}, r, off + 16, 0, 16);
    }
}

template <int SS>
__global__ __launch_bounds__(256) void resnorm_kernel(const float* __restrict__ slab, long slab_stride, const bf16* h_in, bf16* h_out,
                                                      const bf16* __restrict__ w, bf16* x16, int MT, float eps) {
    __shared__ float red[4];
    resnorm_row<SS, 0>(blockIdx.x, threadIdx.x, red, slab, slab_stride, h_in, h_out, w, x16, MT, eps, true);
}

// ---- skinny4 (bf16, MT = 2 m-tiles, one pass), the shipped body; FUSED: producer workgroups + counter hand-off
struct Seam {
    const float* slab; long slab_stride; const bf16* h_in; bf16* h_out; const bf16* w; float eps;
    unsigned* counter;        // zeroed before the launch; producers add their row count
    unsigned* fail;           // set when a bounded spin gives up
    int rows;                 // M
    int delay;                // consumers sleep 64 * delay cycles before their weight loads (0: none)
    unsigned long long* stamps;   // diagnostic: [workgroup][8] wall_clock64() values (100 MHz), or nullptr
};
#define STAMP(k) do { if (sm.stamps && threadIdx.x == 0) sm.stamps[(long)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)

template <int TPW, int NKS, int EPI, int WAVES, int SS, int FUSED>
__global__ __launch_bounds__(64 * WAVES) void sk4_kernel(const bf16* X16, const bf16* __restrict__ Wt, int M, int N, int K, float* __restrict__ slab,
                                                         bf16* __restrict__ out, int gx, const Seam sm) {
    constexpr int MT = 2;
    extern __shared__ f32x4 red4[];                          // [WAVES][TPW][MT][64]
    constexpr int RPW = WAVES / 4;                           // rows per producer workgroup
    int bid = blockIdx.x;
    if constexpr (FUSED) {
        const int NP = (sm.rows + RPW - 1) / RPW;
        if (bid < NP) {                                      // ---- producer: RPW rows, four waves each
            float* red = reinterpret_cast<float*>(red4);
            const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
            const int m = min(bid * RPW + grp, sm.rows - 1);   // a surplus group repeats the last row (same bytes twice)
            STAMP(0);
            if constexpr (FUSED == 2) {
                resnorm_row<SS, 2>(m, t, red + 4 * grp, sm.slab, sm.slab_stride, sm.h_in, sm.h_out, sm.w, const_cast<bf16*>(X16), MT, sm.eps, true);
                STAMP(1);
                return;                                        // no drain, no counter: the granules carry their own tags
            }
            resnorm_row<SS, 1>(m, t, red + 4 * grp, sm.slab, sm.slab_stride, sm.h_in, sm.h_out, sm.w, const_cast<bf16*>(X16), MT, sm.eps, true);
            STAMP(1);                                          // row arithmetic done, stores issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
            __syncthreads();
            STAMP(2);
            if (threadIdx.x == 0) __hip_atomic_fetch_add(sm.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        bid -= NP;
    }
    if constexpr (FUSED) STAMP(0);
    const int bx = bid % gx, by = bid / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = bx * TPW;
    const int ks0 = (by * WAVES + wave) * NKS;
    const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
    bf16x8 w16[NKS][TPW];
    bf16x8 xf[NKS][MT];
    if constexpr (!FUSED) {
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int mc = 0; mc < MT; ++mc) xf[i][mc] = *reinterpret_cast<const bf16x8*>(xp + ((long)i * MT + mc) * 512);
    }
    if constexpr (FUSED) {
        for (int d = 0; d < sm.delay; ++d) __builtin_amdgcn_s_sleep(1);
    }
    const bf16* wp = Wt + ((long)t0 * (K >> 5) + ks0) * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t) w16[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ((long)t * (K >> 5) + i) * 512));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSED == 2) {
        // every wave sweeps its own fragments' granules until all tags are this launch's (bounded)
        const auto xr = make_rsrc(X16);
        u32x4 ga[NKS][MT], gb[NKS][MT];
        unsigned spins = 0;
        if (sm.stamps && threadIdx.x == 0) { (void)__hip_atomic_load(sm.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(1); }
        while (true) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < NKS; ++i)
#pragma unroll
                for (int mc = 0; mc < MT; ++mc) {
                    const int off = (int)((((long)(ks0 + i) * MT + mc) * 512 + lane * 8) * 4);
                    ga[i][mc] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 16);
                    gb[i][mc] = __builtin_amdgcn_raw_buffer_load_b128(xr, off + 16, 0, 16);
                }
#pragma unroll
            for (int i = 0; i < NKS; ++i)
#pragma unroll
                for (int mc = 0; mc < MT; ++mc)
                    ok = ok && ga[i][mc][0] == GRAN_TAG && ga[i][mc][2] == GRAN_TAG && gb[i][mc][0] == GRAN_TAG && gb[i][mc][2] == GRAN_TAG;
            // rows >= M are never published: their lanes' granules stay zero; only rows < M count
            const int li_ = lane & 15;
            bool need = true;
            (void)li_;
            if (__all(ok || !need)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000u) { if (lane == 0) __hip_atomic_store(sm.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        STAMP(2);
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int mc = 0; mc < MT; ++mc) xf[i][mc] = __builtin_bit_cast(bf16x8, u32x4{ga[i][mc][1], ga[i][mc][3], gb[i][mc][1], gb[i][mc][3]});
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
    }
    if constexpr (FUSED == 1) {
        const int NP = (sm.rows + RPW - 1) / RPW;
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            if (sm.stamps) { (void)__hip_atomic_load(sm.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(1); }   // this wave's weights have landed
            while (__hip_atomic_load(sm.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NP) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { __hip_atomic_store(sm.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        STAMP(2);                                            // counter complete
        __syncthreads();
        const auto xr = make_rsrc(X16);
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int mc = 0; mc < MT; ++mc) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)((((long)(ks0 + i) * MT + mc) * 512 + lane * 8) * 2), 0, 16);   // sc1
                xf[i][mc] = __builtin_bit_cast(bf16x8, v);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (sm.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(3); }   // X fragments landed
    }
    f32x4 acc[TPW][MT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int mc = 0; mc < MT; ++mc) acc[t][mc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mc = 0; mc < MT; ++mc) acc[t][mc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w16[i][t], xf[i][mc], acc[t][mc], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int mc = 0; mc < MT; ++mc) red4[((wave * TPW + t) * MT + mc) * 64 + lane] = acc[t][mc];
    __syncthreads();
    if constexpr (FUSED) STAMP(4);                           // MFMAs done, partial sums in LDS
    const int g = lane >> 4, li = lane & 15;
    auto total = [&](int t, int mt) {
        f32x4 a = red4[(t * MT + mt) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < WAVES; ++ww) a += red4[((ww * TPW + t) * MT + mt) * 64 + lane];
        return a;
    };
    if (EPI == 2) {
        for (int c = wave; c < (TPW / 2) * MT; c += WAVES) {
            const int p = c / MT, mt = c - p * MT;
            const f32x4 a = total(2 * p, mt), b = total(2 * p + 1, mt);
            const int m = 16 * mt + li;
            if (m >= M) continue;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float av = round_bf(a[r]), bv = round_bf(b[r]);
                o[r] = f2bf(av / (1.0f + __expf(-av)) * bv);
            }
            *reinterpret_cast<bf16x4*>(out + x16_index(m, ((t0 + 2 * p) >> 1) * 16 + 4 * g, MT)) = o;
        }
        return;
    }
    for (int c = wave; c < TPW * MT; c += WAVES) {
        const int t = c / MT, mt = c - t * MT;
        const f32x4 a = total(t, mt);
        const int m = 16 * mt + li;
        if (m >= M) continue;
        *reinterpret_cast<f32x4*>(slab + ((long)by * M + m) * N + 16 * (t0 + t) + 4 * g) = a;
    }
}

// ------------------------------------------------------------------ host
static hipStream_t st;
static int M_ = 32;

template <int TPW, int NKS, int EPI, int WAVES, int SS, int FUSED>
static void launch_sk4(const bf16* X, const bf16* W, int N, int K, float* slab, bf16* out, const Seam& sm) {
    const int gx = N / 16 / TPW, S = K / (32 * NKS * WAVES);
    const int NP = FUSED != 0 ? (M_ + WAVES / 4 - 1) / (WAVES / 4) : 0;
    const size_t lds = (size_t)WAVES * TPW * 2 * 64 * 16;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)sk4_kernel<TPW, NKS, EPI, WAVES, SS, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((sk4_kernel<TPW, NKS, EPI, WAVES, SS, FUSED>), dim3(gx * S + NP), dim3(64 * WAVES), lds, st, X, W, M_, N, K, slab, out, gx, sm);
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 28;
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    M_ = argc > 3 ? atoi(argv[3]) : 32;
    const int MT = 2, I = 6144, NQKV = 4096;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto alloc_fill = [&](size_t n, unsigned seed, float scale) {
        bf16* p; CK(hipMalloc(&p, n * 2));
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, (long)n, seed, scale);
        return p;
    };
    std::vector<bf16*> Wo(L), Wgu(L), Wdn(L), Wqkv(L);
    for (int l = 0; l < L; ++l) {
        Wo[l] = alloc_fill((size_t)HID * HID, 100u + l, 0.02f);
        Wgu[l] = alloc_fill((size_t)2 * I * HID, 200u + l, 0.02f);
        Wdn[l] = alloc_fill((size_t)HID * I, 300u + l, 0.02f);
        Wqkv[l] = alloc_fill((size_t)NQKV * HID, 400u + l, 0.02f);
    }
    bf16* att16 = alloc_fill((size_t)MT * 16 * HID, 7u, 1.0f);
    bf16* w1 = alloc_fill(HID, 8u, 1.0f);
    bf16* w2 = alloc_fill(HID, 9u, 1.0f);
    bf16* h0 = alloc_fill((size_t)32 * HID, 10u, 1.0f);
    bf16 *h[3], *xa[3], *xb[3], *act[3];
    float *slabA[3], *slabB[3], *slabQ[3];
    unsigned *cnt, *fail;
    // arm C: one granule image (twice the X16 bytes) per seam of every layer, zeroed before every replay
    const size_t gran_bytes = (size_t)MT * 16 * HID * 4;
    char* gran;
    CK(hipMalloc(&gran, gran_bytes * 2 * L));
    for (int a = 0; a < 3; ++a) {
        CK(hipMalloc(&h[a], (size_t)32 * HID * 2));
        CK(hipMalloc(&xa[a], (size_t)MT * 16 * HID * 2)); CK(hipMalloc(&xb[a], (size_t)MT * 16 * HID * 2));
        CK(hipMalloc(&act[a], (size_t)MT * 16 * I * 2));
        CK(hipMalloc(&slabA[a], (size_t)4 * 32 * HID * 4)); CK(hipMalloc(&slabB[a], (size_t)4 * 32 * HID * 4));
        CK(hipMalloc(&slabQ[a], (size_t)2 * 32 * NQKV * 4));
        CK(hipMemset(xa[a], 0, (size_t)MT * 16 * HID * 2)); CK(hipMemset(xb[a], 0, (size_t)MT * 16 * HID * 2));
    }
    const size_t cnt_bytes = (size_t)((2 * L * 4 + 15) / 16) * 16;
    CK(hipMalloc(&cnt, cnt_bytes)); CK(hipMalloc(&fail, 16));
    CK(hipMemset(fail, 0, 16));
    CK(hipStreamSynchronize(st));

    int stamp_layer = -1;
    const int delay_gu = argc > 5 ? atoi(argv[5]) : 0, delay_qkv = argc > 6 ? atoi(argv[6]) : 0;
    unsigned long long *stampA, *stampB;
    CK(hipMalloc(&stampA, 512 * 8 * 8)); CK(hipMalloc(&stampB, 512 * 8 * 8));
    auto layer = [&](int arm, int l) {
        Seam none{};
        // o: att16 -> slabA (4 slabs)
        launch_sk4<2, 2, 0, 8, 4, 0>(att16, Wo[l], HID, HID, slabA[arm], nullptr, none);
        if (arm == 0) {
            hipLaunchKernelGGL(resnorm_kernel<4>, dim3(M_), dim3(256), 0, st, slabA[arm], (long)32 * HID, h[arm], h[arm], w1, xa[arm], MT, 1e-6f);
            launch_sk4<4, 4, 2, 16, 4, 0>(xa[arm], Wgu[l], 2 * I, HID, nullptr, act[arm], none);
        } else {
            Seam s{slabA[arm], (long)32 * HID, h[arm], h[arm], w1, 1e-6f, cnt + 2 * l, fail, M_, delay_gu, stamp_layer == l ? stampA : nullptr};
            if (arm == 1) launch_sk4<4, 4, 2, 16, 4, 1>(xa[arm], Wgu[l], 2 * I, HID, nullptr, act[arm], s);
            else launch_sk4<4, 4, 2, 16, 4, 2>((const bf16*)(gran + gran_bytes * (2 * l)), Wgu[l], 2 * I, HID, nullptr, act[arm], s);
        }
        launch_sk4<2, 3, 0, 16, 4, 0>(act[arm], Wdn[l], HID, I, slabB[arm], nullptr, none);
        if (arm == 0) {
            hipLaunchKernelGGL(resnorm_kernel<4>, dim3(M_), dim3(256), 0, st, slabB[arm], (long)32 * HID, h[arm], h[arm], w2, xb[arm], MT, 1e-6f);
            launch_sk4<2, 4, 0, 8, 4, 0>(xb[arm], Wqkv[l], NQKV, HID, slabQ[arm], nullptr, none);
        } else {
            Seam s{slabB[arm], (long)32 * HID, h[arm], h[arm], w2, 1e-6f, cnt + 2 * l + 1, fail, M_, delay_qkv, stamp_layer == l ? stampB : nullptr};
            if (arm == 1) launch_sk4<2, 4, 0, 8, 4, 1>(xb[arm], Wqkv[l], NQKV, HID, slabQ[arm], nullptr, s);
            else launch_sk4<2, 4, 0, 8, 4, 2>((const bf16*)(gran + gran_bytes * (2 * l + 1)), Wqkv[l], NQKV, HID, slabQ[arm], nullptr, s);
        }
    };
    hipGraphExec_t ge[3];
    for (int arm = 0; arm < 3; ++arm) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        CK(hipMemcpyAsync(h[arm], h0, (size_t)32 * HID * 2, hipMemcpyDeviceToDevice, st));
        if (arm == 1) CK(hipMemsetAsync(cnt, 0, cnt_bytes, st));
        if (arm == 2) CK(hipMemsetAsync(gran, 0, gran_bytes * 2 * L, st));
        for (int l = 0; l < L; ++l) layer(arm, l);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge[arm], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    std::vector<float> ts[3];
    for (int r = 0; r < rounds; ++r)
        for (int arm = 0; arm < 3; ++arm) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge[arm], st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) ts[arm].push_back(ms * 1000.f / L);
        }
    // equality of the two arms: residual stream, last X images, last qkv slabs
    auto fetch = [&](const void* p, size_t n) { std::vector<unsigned char> v(n); CK(hipMemcpy(v.data(), p, n, hipMemcpyDeviceToHost)); return v; };
    size_t diff = 0;
    for (int other = 1; other < 3; ++other) {
        auto a = fetch(h[0], (size_t)M_ * HID * 2), b = fetch(h[other], (size_t)M_ * HID * 2);
        for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
        auto c = fetch(slabQ[0], (size_t)2 * 32 * NQKV * 4), d = fetch(slabQ[other], (size_t)2 * 32 * NQKV * 4);
        // rows >= M are never written: compare the written rows of both slabs
        for (int s = 0; s < 2; ++s)
            for (size_t i = 0; i < (size_t)M_ * NQKV * 4; ++i) diff += c[(size_t)s * M_ * NQKV * 4 + i] != d[(size_t)s * M_ * NQKV * 4 + i];
        auto e = fetch(act[0], (size_t)MT * 16 * I * 2), f = fetch(act[other], (size_t)MT * 16 * I * 2);
        if (M_ == 32) for (size_t i = 0; i < e.size(); ++i) diff += e[i] != f[i];
    }
    unsigned failed = 0;
    CK(hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost));
    const char* arm_name[3] = {"A separate launches (6 per layer)", "B flag hand-off (4 per layer)", "C granule hand-off (4 per layer)"};
    for (int arm = 0; arm < 3; ++arm) {
        std::sort(ts[arm].begin(), ts[arm].end());
        printf("arm %-34s median %.2f us per layer, min %.2f, max %.2f\n", arm_name[arm], ts[arm][ts[arm].size() / 2], ts[arm].front(), ts[arm].back());
    }
    printf("consumer delays: gate/up %d, qkv %d (x 64 cycles)\n", delay_gu, delay_qkv);
    printf("bytes differing between the arms: %zu; spin timeouts: %u\n", diff, failed);
    if (argc > 4 && !strncmp(argv[4], "stamp", 5)) {         // stamp mode: one eager pass of arm B ("stamp") or C ("stampc"), layer L/2 instrumented
        const int sarm = !strcmp(argv[4], "stampc") ? 2 : 1;
        stamp_layer = L / 2;
        CK(hipMemset(stampA, 0, 512 * 64)); CK(hipMemset(stampB, 0, 512 * 64));
        CK(hipMemcpyAsync(h[sarm], h0, (size_t)32 * HID * 2, hipMemcpyDeviceToDevice, st));
        CK(hipMemsetAsync(cnt, 0, cnt_bytes, st));
        CK(hipMemsetAsync(gran, 0, gran_bytes * 2 * L, st));
        for (int l = 0; l < L; ++l) layer(sarm, l);
        CK(hipStreamSynchronize(st));
        const char* names[2] = {"gate/up'", "qkv'"};
        unsigned long long* bufs[2] = {stampA, stampB};
        const int np[2] = {M_ / 4, M_ / 2}, nwg[2] = {192 + M_ / 4, 256 + M_ / 2};
        for (int k = 0; k < 2; ++k) {
            std::vector<unsigned long long> v(512 * 8);
            CK(hipMemcpy(v.data(), bufs[k], 512 * 64, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < nwg[k]; ++b) t0 = std::min(t0, v[b * 8]);
            auto stat = [&](int lo, int hi, int idx, const char* what) {
                std::vector<double> x;
                for (int b = lo; b < hi; ++b) x.push_back((double)(v[b * 8 + idx] - t0) * 0.01);
                std::sort(x.begin(), x.end());
                printf("  %-34s min %6.2f  median %6.2f  max %6.2f us\n", what, x.front(), x[x.size() / 2], x.back());
            };
            printf("%s (times from the first workgroup's entry):\n", names[k]);
            stat(0, np[k], 0, "producer entry");
            stat(0, np[k], 1, "producer row done, stores issued");
            stat(0, np[k], 2, "producer stores drained");
            stat(np[k], nwg[k], 0, "consumer entry");
            stat(np[k], nwg[k], 1, "consumer wave 0 weights landed");
            stat(np[k], nwg[k], 2, "consumer counter complete");
            stat(np[k], nwg[k], 3, "consumer X landed");
            stat(np[k], nwg[k], 4, "consumer MFMAs done");
        }
    }
    return 0;
}
