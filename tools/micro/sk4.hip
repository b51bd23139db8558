// Micro-benchmark: decode weight-streaming GEMM geometries on the W16 / X16 layouts of csrc/decode.hip.
//   sk3  the shipped skinny3_kernel body (bf16 path): waves own RT tiles, K split over the waves of a workgroup and over
//        grid.y slabs, weights through a two-group register double buffer
//   sk4  one fat workgroup (8 waves) per TPW consecutive tiles: every wave issues ALL its loads (NKS k-steps of every
//        tile + its X fragments) before the first MFMA, partial sums meet in LDS, summed in fixed wave order
//   empty  the same grids with an immediate return (launch + dispatch + drain floor)
// Each case: 28 distinct weight matrices (cold: > 256 MiB of Infinity Cache for the larger shapes), 28 dependent launches
// on one stream between two events, median of 7 rounds -> us per launch INCLUDING the kernel boundary.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/sk4.hip -o tools/micro/sk4
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
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
__device__ __forceinline__ float round_bf(float x) { return (float)(bf16)x; }
__device__ __forceinline__ long x16_index(int m, int k, int MT) {
    return ((long)(k >> 5) * MT + (m >> 4)) * 512 + ((((k & 31) >> 3) << 4) + (m & 15)) * 8 + (k & 7);
}

__global__ void fill_kernel(bf16* p, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = f2bf(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
}

__global__ __launch_bounds__(1024) void empty_kernel(int* sink) {
    if (sink && threadIdx.x == 9999) *sink = 1;
}

// ------------------------------------------------------------------ sk3 (shipped body)
template <int MT, int EPI, int RT>
__global__ __launch_bounds__(256) void sk3_kernel(const bf16* __restrict__ X16, const bf16* __restrict__ W16, int M, int N, int K, int kslice,
                                                  int wg_ksplit, float* __restrict__ slab, bf16* __restrict__ out, long ldo) {
    extern __shared__ f32x4 red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int rt0 = (wg_ksplit ? blockIdx.x : blockIdx.x * waves + wave) * RT;
    const int n_tiles = N >> 4;
    if (rt0 >= n_tiles) return;
    const int nks = wg_ksplit ? (kslice >> 5) / waves : (kslice >> 5);
    const int ks0 = blockIdx.y * (kslice >> 5) + (wg_ksplit ? wave * nks : 0);
    f32x4 acc[RT][MT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
    const long tile_stride = (long)(K >> 5) * 512;
    const bf16* wp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) wp[t] = W16 + (long)min(rt0 + t, n_tiles - 1) * tile_stride + (long)ks0 * 512 + lane * 8;
    constexpr int G = RT == 1 ? 8 : 4;
    bf16x8 wA[G][RT], wB[G][RT];
    auto loadw = [&](bf16x8 (&w)[G][RT], int ks) {
#pragma unroll
        for (int i = 0; i < G; ++i)
            if (ks + i < nks) {
#pragma unroll
                for (int t = 0; t < RT; ++t) w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + (long)(ks + i) * 512));
            }
    };
    auto compute = [&](const bf16x8 (&w)[G][RT], int ks) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (ks + i < nks) {
                bf16x8 xf[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(xp + ((long)(ks + i) * MT + mt) * 512);
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][t], xf[mt], acc[t][mt], 0, 0, 0);
            }
        }
    };
    loadw(wA, 0);
    for (int ks = 0; ks < nks; ks += 2 * G) {
        if (ks + G < nks) loadw(wB, ks + G);
        compute(wA, ks);
        if (ks + 2 * G < nks) loadw(wA, ks + 2 * G);
        if (ks + G < nks) compute(wB, ks + G);
    }
    if (wg_ksplit) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) red[((wave * RT + t) * MT + mt) * 64 + lane] = acc[t][mt];
        __syncthreads();
        if (wave != 0) return;
        for (int w = 1; w < waves; ++w)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[t][mt] += red[((w * RT + t) * MT + mt) * 64 + lane];
    }
    const int g = lane >> 4, li = lane & 15;
    if (EPI == 2) {
#pragma unroll
        for (int t = 0; t + 1 < RT; t += 2) {
            const int f0 = ((rt0 + t) >> 1) * 16 + 4 * g;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = 16 * mt + li;
                if (m >= M) continue;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = round_bf(acc[t][mt][r]), b = round_bf(acc[t + 1 < RT ? t + 1 : t][mt][r]);
                    o[r] = f2bf(a / (1.0f + __expf(-a)) * b);
                }
                *reinterpret_cast<bf16x4*>(out + x16_index(m, f0, MT)) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (rt0 + t >= n_tiles) break;
        const int n = 16 * (rt0 + t) + 4 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 16 * mt + li;
            if (m >= M) continue;
            if (EPI == 1) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[t][mt][r]);
                *reinterpret_cast<bf16x4*>(out + (long)m * ldo + n) = o;
            } else {
                *reinterpret_cast<f32x4*>(slab + ((long)blockIdx.y * M + m) * N + n) = acc[t][mt];
            }
        }
    }
}

// ------------------------------------------------------------------ sk4
// grid (n_tiles / TPW, S); 8 waves; wave w of slab s owns k-steps [(8 s + w) NKS, +NKS) of each of the TPW tiles.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(unsigned lo, unsigned hi) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    bf16x8 r;
    r[0] = f2bf(a[0]); r[1] = f2bf(a[1]); r[2] = f2bf(b[0]); r[3] = f2bf(b[1]);
    r[4] = f2bf(c[0]); r[5] = f2bf(c[1]); r[6] = f2bf(d[0]); r[7] = f2bf(d[1]);
    return r;
}

template <int MT, int TPW, int NKS, int EPI, int WAVES, bool F8 = false>
__global__ __launch_bounds__(64 * WAVES) void sk4_kernel(const bf16* __restrict__ X16, const bf16* __restrict__ W16, int M, int N, int K,
                                                         float* __restrict__ slab, bf16* __restrict__ out, long ldo = 0) {
    extern __shared__ f32x4 red[];                          // [WAVES][TPW][MT][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = N >> 4;
    const int t0 = blockIdx.x * TPW;
    const int ks0 = (blockIdx.y * WAVES + wave) * NKS;
    const long tile_stride = (long)(K >> 5) * 512;
    bf16x8 xf[NKS][MT];
    const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[i][mt] = *reinterpret_cast<const bf16x8*>(xp + ((long)i * MT + mt) * 512);
    f32x4 acc[TPW][MT];
    if constexpr (!F8) {
        bf16x8 w[NKS][TPW];
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(W16 + (long)min(t0 + t, n_tiles - 1) * tile_stride + (long)(ks0 + i) * 512 + lane * 8));
        __builtin_amdgcn_sched_barrier(0);                      // every load above is issued before the first MFMA below
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][t], xf[i][mt], acc[t][mt], 0, 0, 0);
    } else {
        static_assert(!F8 || NKS % 2 == 0, "fp8 tiles come in pairs of k steps");
        constexpr int NKP = NKS / 2;
        const char* W8 = reinterpret_cast<const char*>(W16);
        const long tstride8 = (long)(K >> 6) * 1024;
        u32x4 w[NKP > 0 ? NKP : 1][TPW];
#pragma unroll
        for (int i = 0; i < NKP; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W8 + (long)min(t0 + t, n_tiles - 1) * tstride8 + (long)((ks0 >> 1) + i) * 1024 + lane * 16));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NKP; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const bf16x8 wf = h == 0 ? fp8x8_to_bf16(w[i][t][0], w[i][t][1]) : fp8x8_to_bf16(w[i][t][2], w[i][t][3]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[2 * i + h][mt], acc[t][mt], 0, 0, 0);
                }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) red[((wave * TPW + t) * MT + mt) * 64 + lane] = acc[t][mt];
    __syncthreads();
    const int g = lane >> 4, li = lane & 15;
    if (EPI == 2) {
        for (int c = wave; c < (TPW / 2) * MT; c += WAVES) {
            const int p = c / MT, mt = c % MT;
            f32x4 a = red[((0 * TPW + 2 * p) * MT + mt) * 64 + lane], b = red[((0 * TPW + 2 * p + 1) * MT + mt) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) {
                a += red[((ww * TPW + 2 * p) * MT + mt) * 64 + lane];
                b += red[((ww * TPW + 2 * p + 1) * MT + mt) * 64 + lane];
            }
            const int m = 16 * mt + li;
            if (m >= M) continue;
            const int f0 = ((t0 + 2 * p) >> 1) * 16 + 4 * g;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float av = round_bf(a[r]), bv = round_bf(b[r]);
                o[r] = f2bf(av / (1.0f + __expf(-av)) * bv);
            }
            *reinterpret_cast<bf16x4*>(out + x16_index(m, f0, MT)) = o;
        }
        return;
    }
    for (int c = wave; c < TPW * MT; c += WAVES) {
        const int t = c / MT, mt = c % MT;
        if (t0 + t >= n_tiles) continue;
        f32x4 a = red[((0 * TPW + t) * MT + mt) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < WAVES; ++ww) a += red[((ww * TPW + t) * MT + mt) * 64 + lane];
        const int m = 16 * mt + li;
        if (m >= M) continue;
        if (EPI == 1) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(a[r]);
            *reinterpret_cast<bf16x4*>(out + (long)m * ldo + 16 * (t0 + t) + 4 * g) = o;
        } else {
            *reinterpret_cast<f32x4*>(slab + ((long)blockIdx.y * M + m) * N + 16 * (t0 + t) + 4 * g) = a;
        }
    }
}

// sk5: sk4 (bf16, slab epilogue) with the m-tiles taken in passes of MC over the same weight registers (X per pass)
template <int MT, int TPW, int NKS, int WAVES, int MC>
__global__ __launch_bounds__(64 * WAVES) void sk5_kernel(const bf16* __restrict__ X16, const bf16* __restrict__ W16, int M, int N, int K,
                                                         float* __restrict__ slab) {
    extern __shared__ f32x4 red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = blockIdx.x * TPW;
    const int ks0 = (blockIdx.y * WAVES + wave) * NKS;
    const long tile_stride = (long)(K >> 5) * 512;
    bf16x8 xf[NKS][MC], w[NKS][TPW];
    const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) xf[i][mc] = *reinterpret_cast<const bf16x8*>(xp + ((long)i * MT + mc) * 512);
#pragma unroll
    for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
            w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(W16 + (long)(t0 + t) * tile_stride + (long)(ks0 + i) * 512 + lane * 8));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m0 = 0; m0 < MT; m0 += MC) {
        if (m0 > 0) {
#pragma unroll
            for (int i = 0; i < NKS; ++i)
#pragma unroll
                for (int mc = 0; mc < MC; ++mc) xf[i][mc] = *reinterpret_cast<const bf16x8*>(xp + ((long)i * MT + m0 + mc) * 512);
        }
        f32x4 acc[TPW][MC];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mc = 0; mc < MC; ++mc) acc[t][mc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int mc = 0; mc < MC; ++mc) acc[t][mc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][t], xf[i][mc], acc[t][mc], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mc = 0; mc < MC; ++mc) red[((wave * TPW + t) * MT + m0 + mc) * 64 + lane] = acc[t][mc];
    }
    __syncthreads();
    const int g = lane >> 4, li = lane & 15;
    for (int c = wave; c < TPW * MT; c += WAVES) {
        const int t = c / MT, mt = c % MT;
        f32x4 a = red[((0 * TPW + t) * MT + mt) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < WAVES; ++ww) a += red[((ww * TPW + t) * MT + mt) * 64 + lane];
        const int m = 16 * mt + li;
        if (m >= M) continue;
        *reinterpret_cast<f32x4*>(slab + ((long)blockIdx.y * M + m) * N + 16 * (t0 + t) + 4 * g) = a;
    }
}

// sk6: persistent sk4 for the tied head (bf16, EPI 1): grid.x workgroups loop over tile groups t = blockIdx.x, + gridDim.x, ...;
// a wave's X fragments (its k-steps never change) stay in registers for the whole launch; the next group's weights are
// requested before this group's partial sums are reduced.
template <int MT, int TPW, int NKS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void sk6_kernel(const bf16* __restrict__ X16, const bf16* __restrict__ W16, int M, int N, int K,
                                                         bf16* __restrict__ out, long ldo) {
    extern __shared__ f32x4 red[];                          // [2][WAVES][TPW][MT][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_groups = (N >> 4) / TPW;
    const int ks0 = wave * NKS;
    const long tile_stride = (long)(K >> 5) * 512;
    bf16x8 xf[NKS][MT], wA[NKS][TPW], wB[NKS][TPW];
    const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < NKS; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[i][mt] = *reinterpret_cast<const bf16x8*>(xp + ((long)i * MT + mt) * 512);
    auto loadw = [&](bf16x8 (&w)[NKS][TPW], int grp) {
        const bf16* wp = W16 + ((long)grp * TPW * (K >> 5) + ks0) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t) w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ((long)t * (K >> 5) + i) * 512));
    };
    const int g = lane >> 4, li = lane & 15;
    auto body = [&](const bf16x8 (&w)[NKS][TPW], int grp, int buf) {
        f32x4 acc[TPW][MT];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][t], xf[i][mt], acc[t][mt], 0, 0, 0);
        f32x4* r = red + buf * WAVES * TPW * MT * 64;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) r[((wave * TPW + t) * MT + mt) * 64 + lane] = acc[t][mt];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int c = wave; c < TPW * MT; c += WAVES) {
            const int t = c / MT, mt = c % MT;
            f32x4 a = r[(t * MT + mt) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) a += r[((ww * TPW + t) * MT + mt) * 64 + lane];
            const int m = 16 * mt + li;
            if (m >= M) continue;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(a[e]);
            *reinterpret_cast<bf16x4*>(out + (long)m * ldo + 16 * (grp * TPW + t) + 4 * g) = o;
        }
    };
    int grp = blockIdx.x;
    if (grp < n_groups) loadw(wA, grp);
    for (; grp < n_groups; grp += 2 * gridDim.x) {
        if (grp + (int)gridDim.x < n_groups) loadw(wB, grp + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        body(wA, grp, 0);
        if (grp + 2 * (int)gridDim.x < n_groups) loadw(wA, grp + 2 * gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        if (grp + (int)gridDim.x < n_groups) body(wB, grp + gridDim.x, 1);
    }
}

__global__ void ref8_kernel(const bf16* X16, const unsigned char* W8, int M, int N, int K, int MT, float* C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const int q = k >> 6, h = (k >> 5) & 1, g = (k & 31) >> 3, j = k & 7;
        const long wi = ((long)(n >> 4) * (K >> 6) + q) * 1024 + (g * 16 + (n & 15)) * 16 + 8 * h + j;
        const unsigned b = W8[wi];
        const f32x2 v = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, false);
        s += v[0] * bf2f(X16[x16_index(m, k, MT)]);
    }
    C[idx] = s;
}
__global__ void fill8_kernel(unsigned char* p, long n, unsigned seed) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15;
    unsigned b = h & 0xff;
    if ((b & 0x7f) >= 0x40) b &= 0xbf;             // keep magnitudes < 2 (and away from the NaN codes)
    p[i] = (unsigned char)b;
}

// reference: sum of slabs / act against a naive fp32 product (one thread per output)
__global__ void ref_kernel(const bf16* X16, const bf16* W16, int M, int N, int K, int MT, float* C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const long wi = ((long)(n >> 4) * (K >> 5) + (k >> 5)) * 512 + ((((k & 31) >> 3) << 4) + (n & 15)) * 8 + (k & 7);
        s += bf2f(W16[wi]) * bf2f(X16[x16_index(m, k, MT)]);
    }
    C[idx] = s;
}

struct Shape { const char* name; int N, K, epi; };
static int M_ = 32, MT_ = 2;
static int L = 28;
static bool F8_ = false;
static hipStream_t st;
static std::vector<bf16*> Ws;
static bf16 *X = nullptr, *OUT = nullptr;
static float *SLAB = nullptr, *REF = nullptr;
static hipEvent_t e0, e1;

template <class F>
static float time_case(F launch) {                // us per launch, median of 7 replays of a graph of L launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; ++l) launch(Ws[l]);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<float> ts;
    for (int r = 0; r < 8; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) ts.push_back(ms * 1000.f / L);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

static double check(const Shape& s, int S) {      // max error against the naive product (first matrix), relative to the largest output
    std::vector<float> ref((size_t)M_ * s.N), got;
    CK(hipMemcpy(ref.data(), REF, ref.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (float v : ref) scale = std::max(scale, (double)fabsf(v));
    if (s.epi == 2) {
        const int F = s.N / 2;
        std::vector<bf16> act((size_t)MT_ * 16 * F);
        CK(hipMemcpy(act.data(), OUT, act.size() * 2, hipMemcpyDeviceToHost));
        for (int m = 0; m < M_; ++m)
            for (int f = 0; f < F; ++f) {
                const int tile = f / 16, r = f % 16;
                const float a = (float)(bf16)ref[(size_t)m * s.N + 32 * tile + r], b = (float)(bf16)ref[(size_t)m * s.N + 32 * tile + 16 + r];
                const float want = a / (1.0f + expf(-a)) * b;
                const long xi = ((long)(f >> 5) * MT_ + (m >> 4)) * 512 + ((((f & 31) >> 3) << 4) + (m & 15)) * 8 + (f & 7);
                worst = std::max(worst, (double)fabsf((float)act[xi] - want));
            }
        return worst / std::max(scale * scale, 1e-9);
    }
    if (s.epi == 1) {
        std::vector<bf16> o((size_t)M_ * s.N);
        CK(hipMemcpy(o.data(), OUT, o.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ref.size(); ++i) worst = std::max(worst, (double)fabsf((float)o[i] - ref[i]));
        return worst / std::max(scale, 1e-9);
    }
    got.resize((size_t)S * M_ * s.N);
    CK(hipMemcpy(got.data(), SLAB, got.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ref.size(); ++i) {
        float v = 0;
        for (int q = 0; q < S; ++q) v += got[(size_t)q * M_ * s.N + i];
        worst = std::max(worst, (double)fabsf(v - ref[i]));
    }
    return worst / std::max(scale, 1e-9);
}

static double wbytes(const Shape& s) { return (F8_ ? 1.0 : 2.0) * s.N * s.K; }

template <int MT, int EPI, int RT>
static void run_sk3(const Shape& s, int S, int waves, int ksplit) {
    if (F8_) return;
    const int n_tiles = s.N / 16, per_wg = (ksplit ? 1 : waves) * RT;
    dim3 grid((n_tiles + per_wg - 1) / per_wg, S), blk(64 * waves);
    const size_t lds = ksplit ? (size_t)waves * RT * MT * 64 * 16 : 0;
    const int kslice = s.K / S;
    auto go = [&](bf16* W) { hipLaunchKernelGGL((sk3_kernel<MT, EPI, RT>), grid, blk, lds, st, X, W, M_, s.N, s.K, kslice, ksplit, SLAB, OUT, (long)s.N); };
    go(Ws[0]); CK(hipStreamSynchronize(st));
    const double err = check(s, S);
    const float us = time_case(go);
    printf("%-5s sk3 rt=%d waves=%d S=%d ksplit=%d grid=%5dx%-2d  %7.2f us  %5.2f TB/s  err %.1e\n", s.name, RT, waves, S, ksplit,
           grid.x, grid.y, us, wbytes(s) / us * 1e-6, err);
    fflush(stdout);
}

template <int MT, int TPW, int NKS, int EPI, int WAVES, bool F8>
static void run_sk4_(const Shape& s) {
    const int n_tiles = s.N / 16;
    const int S = s.K / 32 / (WAVES * NKS);
    if (S < 1 || S * WAVES * NKS * 32 != s.K || n_tiles % TPW || (EPI != 0 && S != 1)) return;
    dim3 grid(n_tiles / TPW, S), blk(64 * WAVES);
    const size_t lds = (size_t)WAVES * TPW * MT * 64 * 16;
    if (lds > 160 * 1024) return;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)sk4_kernel<MT, TPW, NKS, EPI, WAVES, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    auto go = [&](bf16* W) { hipLaunchKernelGGL((sk4_kernel<MT, TPW, NKS, EPI, WAVES, F8>), grid, blk, lds, st, X, W, M_, s.N, s.K, SLAB, OUT, (long)s.N); };
    go(Ws[0]); CK(hipStreamSynchronize(st));
    const double err = check(s, S);
    const float us = time_case(go);
    printf("%-5s sk4 TPW=%d NKS=%d WAVES=%2d S=%d grid=%5dx%-2d  %7.2f us  %5.2f TB/s  err %.1e\n", s.name, TPW, NKS, WAVES, S,
           grid.x, grid.y, us, wbytes(s) / us * 1e-6, err);
    fflush(stdout);
}
template <int MT, int TPW, int NKS, int EPI, int WAVES>
static void run_sk4(const Shape& s) {
    if (F8_) { if constexpr (NKS % 2 == 0) run_sk4_<MT, TPW, NKS, EPI, WAVES, true>(s); }
    else run_sk4_<MT, TPW, NKS, EPI, WAVES, false>(s);
}

template <int MT, int TPW, int NKS, int WAVES, int MC>
static void run_sk5(const Shape& s) {
    if (F8_) return;
    const int n_tiles = s.N / 16;
    const int S = s.K / 32 / (WAVES * NKS);
    if (S < 1 || S * WAVES * NKS * 32 != s.K || n_tiles % TPW || s.epi != 0) return;
    dim3 grid(n_tiles / TPW, S), blk(64 * WAVES);
    const size_t lds = (size_t)WAVES * TPW * MT * 64 * 16;
    if (lds > 160 * 1024) return;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)sk5_kernel<MT, TPW, NKS, WAVES, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    auto go = [&](bf16* W) { hipLaunchKernelGGL((sk5_kernel<MT, TPW, NKS, WAVES, MC>), grid, blk, lds, st, X, W, M_, s.N, s.K, SLAB); };
    go(Ws[0]); CK(hipStreamSynchronize(st));
    const double err = check(s, S);
    const float us = time_case(go);
    printf("%-5s sk5 TPW=%d NKS=%d WAVES=%2d MC=%d S=%d grid=%5dx%-2d  %7.2f us  %5.2f TB/s  err %.1e\n", s.name, TPW, NKS, WAVES, MC, S,
           grid.x, grid.y, us, wbytes(s) / us * 1e-6, err);
    fflush(stdout);
}

template <int MT, int TPW, int NKS, int WAVES>
static void run_sk6(const Shape& s, int nwg) {
    if (F8_ || s.epi != 1 || WAVES * NKS * 32 != s.K || (s.N / 16) % TPW) return;
    dim3 grid(nwg), blk(64 * WAVES);
    const size_t lds = (size_t)2 * WAVES * TPW * MT * 64 * 16;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)sk6_kernel<MT, TPW, NKS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    auto go = [&](bf16* W) { hipLaunchKernelGGL((sk6_kernel<MT, TPW, NKS, WAVES>), grid, blk, lds, st, X, W, M_, s.N, s.K, OUT, (long)s.N); };
    CK(hipMemsetAsync(OUT, 0, (size_t)M_ * s.N * 2, st));
    go(Ws[0]); CK(hipStreamSynchronize(st));
    const double err = check(s, 1);
    const float us = time_case(go);
    printf("%-5s sk6 TPW=%d NKS=%d WAVES=%2d persistent grid=%5d  %7.2f us  %5.2f TB/s  err %.1e\n", s.name, TPW, NKS, WAVES, nwg, us, wbytes(s) / us * 1e-6, err);
    fflush(stdout);
}

template <int MT>
static void sweep(const char* only) {
    const Shape shapes[] = {{"qkv", 4096, 2048, 0}, {"o", 2048, 2048, 0}, {"gu", 12288, 2048, 2}, {"down", 2048, 6144, 0}, {"head", 151936, 2048, 1}};
    for (const Shape& s : shapes) {
        if (only && strcmp(only, s.name)) continue;
        const size_t wn = (size_t)s.N * s.K;
        L = s.epi == 1 ? 4 : 28;
        for (int l = 0; l < L; ++l) {
            bf16* p; CK(hipMalloc(&p, wn * (F8_ ? 1 : 2)));
            if (F8_) hipLaunchKernelGGL(fill8_kernel, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, st, (unsigned char*)p, (long)wn, 1234u + l);
            else hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, st, p, (long)wn, 1234u + l, 0.05f);
            Ws.push_back(p);
        }
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((size_t)MT * 16 * s.K + 255) / 256)), dim3(256), 0, st, X, (long)MT * 16 * s.K, 77u, 1.0f);
        if (F8_) hipLaunchKernelGGL(ref8_kernel, dim3((unsigned)(((long)M_ * s.N + 255) / 256)), dim3(256), 0, st, X, (const unsigned char*)Ws[0], M_, s.N, s.K, MT, REF);
        else hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((long)M_ * s.N + 255) / 256)), dim3(256), 0, st, X, Ws[0], M_, s.N, s.K, MT, REF);
        CK(hipStreamSynchronize(st));
        if (s.epi == 2) {
            run_sk3<MT, 2, 2>(s, 1, 4, 1);
            run_sk4<MT, 2, 8, 2, 8>(s);
            run_sk4<MT, 4, 8, 2, 8>(s);
            run_sk4<MT, 4, 4, 2, 16>(s);
            run_sk4<MT, 2, 4, 2, 16>(s);
            run_sk4<MT, 6, 8, 2, 8>(s);
        } else if (s.epi == 1) {
            run_sk3<MT, 1, 2>(s, 1, 4, 1);
            run_sk4<MT, 2, 8, 1, 8>(s);
            run_sk4<MT, 4, 8, 1, 8>(s);
            run_sk4<MT, 2, 4, 1, 16>(s);
            run_sk4<MT, 4, 4, 1, 16>(s);
            run_sk4<MT, 8, 4, 1, 16>(s);
            run_sk4<MT, 8, 8, 1, 8>(s);
            for (int nwg : {256, 512, 768, 1024}) {
                run_sk6<MT, 2, 8, 8>(s, nwg);
                run_sk6<MT, 2, 4, 16>(s, nwg);
            }
            run_sk6<MT, 4, 4, 16>(s, 256);
            run_sk6<MT, 4, 4, 16>(s, 512);
        } else if (s.K == 6144) {
            run_sk3<MT, 0, 1>(s, 8, 4, 1);
            run_sk4<MT, 2, 8, 0, 8>(s);      // S=3
            run_sk4<MT, 2, 6, 0, 8>(s);      // S=4
            run_sk4<MT, 2, 4, 0, 8>(s);      // S=6
            run_sk4<MT, 2, 3, 0, 8>(s);      // S=8
            run_sk4<MT, 2, 2, 0, 8>(s);      // S=12
            run_sk4<MT, 4, 2, 0, 8>(s);      // S=12
            run_sk4<MT, 2, 6, 0, 16>(s);     // S=2
            run_sk4<MT, 2, 3, 0, 16>(s);     // S=4
            run_sk4<MT, 2, 2, 0, 16>(s);     // S=6
            run_sk4<MT, 4, 2, 0, 16>(s);     // S=6
            run_sk4<MT, 1, 2, 0, 16>(s);     // S=6
            if constexpr (MT == 2) {
                run_sk5<MT, 1, 24, 8, 1>(s);     // complete sums: 128 workgroups of 8 waves
                run_sk5<MT, 1, 12, 16, 1>(s);    // complete sums: 128 workgroups of 16 waves
                run_sk5<MT, 1, 12, 16, 2>(s);
                run_sk5<MT, 1, 12, 8, 1>(s);     // S=2
                run_sk5<MT, 1, 12, 8, 2>(s);     // S=2
            } else {
                run_sk5<MT, 1, 24, 8, 1>(s);
                run_sk5<MT, 1, 12, 16, 1>(s);
            }
        } else {
            run_sk3<MT, 0, 2>(s, 4, 4, 1);
            run_sk3<MT, 0, 1>(s, 8, 2, 0);
            run_sk4<MT, 2, 4, 0, 8>(s);      // S=2
            run_sk4<MT, 2, 2, 0, 8>(s);      // S=4
            run_sk4<MT, 1, 2, 0, 8>(s);      // S=4
            run_sk4<MT, 4, 2, 0, 8>(s);      // S=4
            run_sk4<MT, 2, 2, 0, 16>(s);     // S=2
            run_sk4<MT, 2, 4, 0, 16>(s);     // S=1
            run_sk4<MT, 4, 2, 0, 16>(s);     // S=2
            run_sk4<MT, 2, 1, 0, 16>(s);     // S=4 (bf16 only)
            run_sk4<MT, 4, 1, 0, 16>(s);     // S=4 (bf16 only)
            run_sk4<MT, 2, 1, 0, 8>(s);      // S=8 (bf16 only)
            run_sk4<MT, 1, 8, 0, 8>(s);      // complete sums
            run_sk4<MT, 1, 4, 0, 16>(s);     // complete sums
            run_sk4<MT, 2, 8, 0, 8>(s);      // complete sums
        }
        for (bf16* p : Ws) CK(hipFree(p));
        Ws.clear();
    }
}

int main(int argc, char** argv) {
    M_ = argc > 1 ? atoi(argv[1]) : 32;
    F8_ = argc > 2 && !strcmp(argv[2], "fp8");
    const char* only = argc > 3 ? argv[3] : nullptr;
    MT_ = M_ <= 16 ? 1 : 2;
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMalloc(&X, (size_t)2 * 16 * 6144 * 2));
    CK(hipMalloc(&OUT, (size_t)32 * 151936 * 2));
    CK(hipMalloc(&SLAB, (size_t)16 * 32 * 12288 * 4));
    CK(hipMalloc(&REF, (size_t)32 * 151936 * 4));
    printf("M = %d (MT = %d) %s, us per launch incl. boundary (graph of dependent launches over distinct matrices)\n", M_, MT_, F8_ ? "fp8 weights" : "bf16 weights");
    if (MT_ == 1) sweep<1>(only); else sweep<2>(only);
    return 0;
}
