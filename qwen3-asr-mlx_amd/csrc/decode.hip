// Decode step, HBM-bound kernels in their streaming layouts (reference generate.py:289-313 ->
// decoder.py:219-253 at T = 1, batch M <= 64).
//
// Layouts (static weights are re-tiled once at load; activations are written in this form by their
// producer kernels, so every wave-level load below is ONE contiguous, fully used 1 KiB):
//   W16  [N/16][K/32][64 lanes][8]   lane (g = lane>>4, li = lane&15) holds W[16 rt + li][32 ks + 8 g + j]
//                                    -> a row tile's K range is one contiguous stream (K/32 KiB);
//   X16  [K/32][MT][64 lanes][8]     lane holds X[16 mt + li][32 ks + 8 g + j], MT = ceil(M/16).
// These are exactly the A / B operand fragments of v_mfma_f32_16x16x32_bf16, so the GEMM is
// global_load_dwordx4 -> MFMA with no LDS and no barrier; X16 (128 KB at K = 2048) stays in L1/L2.
#include "common.h"
#include <atomic>

// m-tiles of 16 rows in every X16 image of a step: 1, 2, 3 or 4 (ops.mt_of holds the same rule).  Round 5: three m-tiles for 33-48
// rows (rounds 1-4 went from 2 straight to 4: the 40 segments of BASELINE configs[4] paid the MFMAs, the X fragments and the
// partial-sum images of 64 rows).
static inline int q3_mt_of(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 48 ? 3 : 4)); }
__device__ __forceinline__ long x16_index(int m, int k, int MT) {      // element offset of X[m][k]
    return ((long)(k >> 5) * MT + (m >> 4)) * 512 + ((((k & 31) >> 3) << 4) + (m & 15)) * 8 + (k & 7);
}

// ---------------------------------------------------------------------------
// skinny3: C[M, N] = X . W^T.  One wave per RT consecutive 16-row weight tiles and one K slice: the
// X16 fragments it loads (L1/L2) are reused for RT weight tiles, so activation loads are 1/RT of the
// weight loads; RT*G KiB of weights are in flight per wave (nontemporal: each byte is used once).
// A workgroup is a bundle of waves that either own different tiles of the same K slice
// (grid (ceil(N/16/RT/waves), S)) or -- wg_ksplit -- split the K slice of ONE tile group among
// themselves and add their partial sums through LDS in fixed wave order (grid (N/16/RT, S)); the
// latter gives complete sums per workgroup, which the SwiGLU epilogue needs, with the same
// parallelism as a split-K launch.
//   EPI 0: fp32 slab [S][M][N] (consumer adds slabs in fixed order)
//   EPI 1: bf16 row-major out [M][ldo]                      (tied lm_head -> logits)
//   EPI 2: SwiGLU of the (gate tile, up tile) pairs a wave owns -> bf16 act in X16 layout (K' = N/2), S = 1
// Algorithmic bytes per launch: 2*N*K.
// ---------------------------------------------------------------------------
// fp8 (OCP e4m3) weights: W16F8 [N/16][K/64][64 lanes][16 B] -- a lane's 16 bytes are its fragments of two
// consecutive 32-wide k steps (bytes 0-7: step 2q, bytes 8-15: step 2q+1) -- plus one fp32 scale per weight
// row.  The bytes are widened to bf16 in registers (exact) and go through the same bf16 MFMA; the row scale
// multiplies the accumulator in the epilogue.  Half the HBM bytes of the bf16 stream.
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(unsigned lo, unsigned hi) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    bf16x8 r;
    r[0] = f2bf(a[0]); r[1] = f2bf(a[1]); r[2] = f2bf(b[0]); r[3] = f2bf(b[1]);
    r[4] = f2bf(c[0]); r[5] = f2bf(c[1]); r[6] = f2bf(d[0]); r[7] = f2bf(d[1]);
    return r;
}

// FUSE (M <= 2, 256 threads): the activation operand is not read from X16 but built by every workgroup for itself from
// the previous GEMM's split-K slabs -- h = bf16(h + bf16(sum slabs)), x = RMSNorm(h) * w, the arithmetic of
// slab_resnorm_x16_kernel thread for thread (bit-identical) -- into LDS rows, while the first weight tiles are
// already in flight.  Saves the residual+norm launch of the dependent chain (decoder.py:189-196 at batch 1); workgroup
// (0, 0) also writes h to a buffer other than h_in (its neighbours are still reading that).
struct FuseNorm {
    const float* slab; int S; long stride;
    const bf16* h_in; bf16* h_out; const bf16* w; float eps;
};

__device__ __forceinline__ void fuse_norm_prologue(const FuseNorm& f, int M, int K, bf16* xl, float* hl, float* fred, bool writer) {
    // the first 256 threads work (the thread -> column mapping of slab_resnorm_x16_kernel); a larger workgroup's other
    // threads only meet them at the barriers
    const int NV = K >> 3, tid = threadIdx.x;
    const int v0 = tid < 256 ? tid : NV;
    for (int m = 0; m < M; ++m) {
        float q2 = 0.f;
        for (int v = v0; v < NV; v += 256) {
            const long c0 = (long)m * K + 8 * v;
            f32x4 a0 = *reinterpret_cast<const f32x4*>(f.slab + c0), a1 = *reinterpret_cast<const f32x4*>(f.slab + c0 + 4);
            int s = 1;
            for (; s + 1 < f.S; s += 2) {                  // two slabs per trip: independent loads in flight
                const float* p = f.slab + s * f.stride + c0;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p), b1 = *reinterpret_cast<const f32x4*>(p + 4);
                const f32x4 c0v = *reinterpret_cast<const f32x4*>(p + f.stride), c1v = *reinterpret_cast<const f32x4*>(p + f.stride + 4);
                a0 += b0; a1 += b1; a0 += c0v; a1 += c1v;
            }
            if (s < f.S) {
                const float* p = f.slab + s * f.stride + c0;
                a0 += *reinterpret_cast<const f32x4*>(p);
                a1 += *reinterpret_cast<const f32x4*>(p + 4);
            }
            const bf16x8 hi = *reinterpret_cast<const bf16x8*>(f.h_in + c0);
            bf16x8 ho;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = round_bf(round_bf(e < 4 ? a0[e] : a1[e - 4]) + bf2f(hi[e]));
                hl[8 * v + e] = t;
                q2 = fmaf(t, t, q2);
                ho[e] = f2bf(t);
            }
            if (writer) *reinterpret_cast<bf16x8*>(f.h_out + c0) = ho;
        }
        q2 = wave_sum(q2);
        if ((tid & 63) == 0 && tid < 256) fred[tid >> 6] = q2;
        __syncthreads();
        const float r = rsqrtf((fred[0] + fred[1] + fred[2] + fred[3]) * (1.0f / K) + f.eps);
        for (int v = v0; v < NV; v += 256) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(f.w + 8 * v);
            bf16x8 xo;
#pragma unroll
            for (int e = 0; e < 8; ++e) xo[e] = f2bf(hl[8 * v + e] * r * bf2f(wv[e]));
            *reinterpret_cast<bf16x8*>(xl + (long)m * K + 8 * v) = xo;
        }
        __syncthreads();
    }
}

template <int MT, int EPI, int RT, bool F8, bool FUSE = false>
__global__ __launch_bounds__(256) void skinny3_kernel(const bf16* __restrict__ X16, const void* __restrict__ Wt, int M,
                                                      int N, int K, int kslice, int wg_ksplit, const float* __restrict__ w_scale,
                                                      float* __restrict__ slab, bf16* __restrict__ out, long ldo, const FuseNorm fz) {
    extern __shared__ f32x4 red_raw[];                      // [FUSE: x rows [2][K] bf16 | K floats] wg_ksplit: [waves][RT][MT][64]
    __shared__ float fred[4];
    bf16* xl = reinterpret_cast<bf16*>(red_raw);            // row-major: lanes of rows >= M re-read the last row (results unused)
    float* hl = reinterpret_cast<float*>(reinterpret_cast<char*>(red_raw) + (size_t)K * 4);
    f32x4* red = FUSE ? reinterpret_cast<f32x4*>(reinterpret_cast<char*>(red_raw) + (size_t)K * 8) : red_raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int rt0 = (wg_ksplit ? blockIdx.x : blockIdx.x * waves + wave) * RT;
    const int n_tiles = N >> 4;
    if (!FUSE && rt0 >= n_tiles) return;                    // whole workgroup when wg_ksplit; else no barrier below
    const int nks = wg_ksplit ? (kslice >> 5) / waves : (kslice >> 5);
    const int ks0 = blockIdx.y * (kslice >> 5) + (wg_ksplit ? wave * nks : 0);
    f32x4 acc[RT][MT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
    const int xrow = min(lane & 15, M - 1) * K + 8 * (lane >> 4);      // FUSE: this lane's fragment inside the LDS rows
    if (!F8) {
        const bf16* W16 = (const bf16*)Wt;
        const long tile_stride = (long)(K >> 5) * 512;
        const bf16* wp[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) wp[t] = W16 + (long)min(rt0 + t, n_tiles - 1) * tile_stride + (long)ks0 * 512 + lane * 8;
        constexpr int G = RT == 1 ? 8 : 4;                  // ks steps per group
        bf16x8 wA[G][RT], wB[G][RT];
        auto loadw = [&](bf16x8 (&w)[G][RT], int ks) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (ks + i < nks) {
#pragma unroll
                    for (int t = 0; t < RT; ++t)
                        w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + (long)(ks + i) * 512));
                }
        };
        auto compute = [&](const bf16x8 (&w)[G][RT], int ks) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                if (ks + i < nks) {
                    bf16x8 xf[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        xf[mt] = FUSE ? *reinterpret_cast<const bf16x8*>(xl + xrow + (ks0 + ks + i) * 32)
                                      : *reinterpret_cast<const bf16x8*>(xp + ((long)(ks + i) * MT + mt) * 512);
#pragma unroll
                    for (int t = 0; t < RT; ++t)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][t], xf[mt], acc[t][mt], 0, 0, 0);
                }
            }
        };
        loadw(wA, 0);
        if constexpr (FUSE) {
            fuse_norm_prologue(fz, M, K, xl, hl, fred, blockIdx.x == 0 && blockIdx.y == 0);
            if (rt0 >= n_tiles) return;                     // after the prologue's barriers (wg_ksplit: whole workgroup)
        }
        for (int ks = 0; ks < nks; ks += 2 * G) {
            if (ks + G < nks) loadw(wB, ks + G);
            compute(wA, ks);
            if (ks + 2 * G < nks) loadw(wA, ks + 2 * G);
            if (ks + G < nks) compute(wB, ks + G);
        }
    } else {
        const char* W8 = (const char*)Wt;
        const long tile_stride = (long)(K >> 6) * 1024;     // bytes per 16-row tile
        const int nkp = nks >> 1, kp0 = ks0 >> 1;           // pairs of k steps
        const char* wp[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) wp[t] = W8 + (long)min(rt0 + t, n_tiles - 1) * tile_stride + (long)kp0 * 1024 + lane * 16;
        constexpr int G = RT == 1 ? 8 : 4;                  // pairs per group: the same bytes in flight as the bf16 path
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 wA[G][RT], wB[G][RT];
        auto loadw = [&](u32x4 (&w)[G][RT], int kp) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (kp + i < nkp) {
#pragma unroll
                    for (int t = 0; t < RT; ++t)
                        w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[t] + (long)(kp + i) * 1024));
                }
        };
        auto compute = [&](const u32x4 (&w)[G][RT], int kp) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                if (kp + i < nkp) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bf16x8 xf[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            xf[mt] = FUSE ? *reinterpret_cast<const bf16x8*>(xl + xrow + (ks0 + 2 * (kp + i) + h) * 32)
                                          : *reinterpret_cast<const bf16x8*>(xp + ((long)(2 * (kp + i) + h) * MT + mt) * 512);
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            const bf16x8 wf = h == 0 ? fp8x8_to_bf16(w[i][t][0], w[i][t][1]) : fp8x8_to_bf16(w[i][t][2], w[i][t][3]);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[mt], acc[t][mt], 0, 0, 0);
                        }
                    }
                }
            }
        };
        loadw(wA, 0);
        if constexpr (FUSE) {
            fuse_norm_prologue(fz, M, K, xl, hl, fred, blockIdx.x == 0 && blockIdx.y == 0);
            if (rt0 >= n_tiles) return;
        }
        for (int kp = 0; kp < nkp; kp += 2 * G) {
            if (kp + G < nkp) loadw(wB, kp + G);
            compute(wA, kp);
            if (kp + 2 * G < nkp) loadw(wA, kp + 2 * G);
            if (kp + G < nkp) compute(wB, kp + G);
        }
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
    // lane (g, li) holds C[n = 16 (rt0+t) + 4 g + r][m = 16 mt + li]
    const int g = lane >> 4, li = lane & 15;
    if (F8) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(w_scale + 16 * min(rt0 + t, n_tiles - 1) + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] *= sc;
        }
    }
    if (EPI == 2) {
#pragma unroll
        for (int t = 0; t + 1 < RT; t += 2) {               // tile rt0+t = gate rows, rt0+t+1 = up rows of 16 features
            const int f0 = ((rt0 + t) >> 1) * 16 + 4 * g;   // feature index = k of the down projection
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = 16 * mt + li;
                if (m >= M) continue;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = round_bf(acc[t][mt][r]), b = round_bf(acc[t + 1 < RT ? t + 1 : t][mt][r]);   // bf16 gate/up (decoder.py:97)
                    o[r] = swiglu_bf(a, b);
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
            if (EPI == 0) {
                *reinterpret_cast<f32x4*>(slab + ((long)blockIdx.y * M + m) * N + n) = acc[t][mt];
            } else {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[t][mt][r]);
                *reinterpret_cast<bf16x4*>(out + (long)m * ldo + n) = o;
            }
        }
    }
}

static int skinny3_launch(const char* who, const void* X16, const FuseNorm* fuse, int M, const void* W16, const float* w_scale, int N, int K,
                          int n_splits, int epi, int waves_per_wg, int tiles_per_wave, int wg_ksplit, float* slab, void* out,
                          int64_t ldo, void* stream) {
    Q3_REQUIRE((X16 || fuse) && W16, "%s: null pointer", who);
    Q3_REQUIRE(M > 0 && M <= 64 && N > 0 && N % 16 == 0, "%s: M=%d must be 1..64, N=%d a multiple of 16", who, M, N);
    Q3_REQUIRE(epi >= 0 && epi <= 2 && (waves_per_wg == 1 || waves_per_wg == 2 || waves_per_wg == 4), "%s: bad epi/waves", who);
    Q3_REQUIRE(tiles_per_wave == 1 || tiles_per_wave == 2 || tiles_per_wave == 4, "%s: tiles_per_wave=%d", who, tiles_per_wave);
    const int kq = w_scale ? 64 : 32;                       // fp8 tiles are addressed in pairs of k steps
    Q3_REQUIRE(n_splits >= 1 && K % (n_splits * kq) == 0, "%s: K=%d must be a multiple of %d*n_splits (%d)", who, K, kq, n_splits);
    if (epi == 0) Q3_REQUIRE(slab, "%s: slab epilogue without workspace", who);
    else Q3_REQUIRE(out && n_splits == 1, "%s: direct epilogues need out and n_splits == 1", who);
    if (epi == 1) Q3_REQUIRE(ldo % 4 == 0 && ldo >= N, "%s: bad ldo", who);
    if (epi == 2) Q3_REQUIRE(tiles_per_wave >= 2 && (N / 16) % tiles_per_wave == 0, "%s: SwiGLU needs whole (gate,up) tile pairs per wave", who);
    const int MT = q3_mt_of(M);
    const int RT = tiles_per_wave;
    const int n_tiles = N / 16;
    const int per_wg = (wg_ksplit ? 1 : waves_per_wg) * RT;
    if (wg_ksplit) Q3_REQUIRE((K / n_splits) % (kq * waves_per_wg) == 0, "%s: K slice not divisible among the waves", who);
    dim3 grid((n_tiles + per_wg - 1) / per_wg, n_splits), blk(64 * waves_per_wg);
    size_t lds = wg_ksplit ? (size_t)waves_per_wg * RT * MT * 64 * sizeof(f32x4) : 0;
    hipStream_t s = (hipStream_t)stream;
    const bf16* x = (const bf16*)X16; bf16* o = (bf16*)out;
    const int kslice = K / n_splits;
    FuseNorm fz{};
    if (fuse) {
        fz = *fuse;
        Q3_REQUIRE(M <= 2 && waves_per_wg == 4 && epi != 1 && K % 32 == 0, "%s: fused norm needs M <= 2, 4 waves per workgroup, epi 0 or 2", who);
        Q3_REQUIRE(fz.slab && fz.S >= 1 && fz.h_in && fz.h_out && fz.w && fz.h_in != fz.h_out && fz.slab != slab,
                   "%s: fused norm needs its own input slabs and distinct h_in / h_out", who);
        lds += (size_t)K * 8;                               // two bf16 rows + K floats
        Q3_REQUIRE(lds <= 144 * 1024, "%s: K=%d too large for the fused activation image", who, K);
    }
#define Q3_L(MT_, EPI_, RT_, F8_) hipLaunchKernelGGL((skinny3_kernel<MT_, EPI_, RT_, F8_>), grid, blk, lds, s, x, W16, M, N, K, kslice, wg_ksplit, w_scale, slab, o, (long)ldo, fz)
#define Q3_F(MT_, EPI_, RT_) do { if (w_scale) Q3_L(MT_, EPI_, RT_, true); else Q3_L(MT_, EPI_, RT_, false); } while (0)
#define Q3_R(MT_, EPI_) do { if (RT == 1) Q3_F(MT_, EPI_, 1); else if (RT == 2) Q3_F(MT_, EPI_, 2); else Q3_F(MT_, EPI_, 4); } while (0)
#define Q3_M(EPI_) do { if (MT == 1) Q3_R(1, EPI_); else if (MT == 2) Q3_R(2, EPI_); else if (MT == 3) Q3_R(3, EPI_); else Q3_R(4, EPI_); } while (0)
#define Q3_LZ(EPI_, RT_, F8_) do {                                                                                                   \
        static std::atomic<bool> attr{false};                                                                                                    \
        if (!attr) { (void)hipFuncSetAttribute((const void*)skinny3_kernel<1, EPI_, RT_, F8_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); attr = true; } \
        hipLaunchKernelGGL((skinny3_kernel<1, EPI_, RT_, F8_, true>), grid, blk, lds, s, x, W16, M, N, K, kslice, wg_ksplit, w_scale, slab, o, (long)ldo, fz); \
    } while (0)
#define Q3_FZ(EPI_, RT_) do { if (w_scale) Q3_LZ(EPI_, RT_, true); else Q3_LZ(EPI_, RT_, false); } while (0)
#define Q3_RZ(EPI_) do { if (RT == 1) Q3_FZ(EPI_, 1); else if (RT == 2) Q3_FZ(EPI_, 2); else Q3_FZ(EPI_, 4); } while (0)
    if (fuse) { if (epi == 0) Q3_RZ(0); else Q3_RZ(2); }
    else if (epi == 0) Q3_M(0); else if (epi == 1) Q3_M(1); else Q3_M(2);
#undef Q3_RZ
#undef Q3_FZ
#undef Q3_LZ
#undef Q3_M
#undef Q3_R
#undef Q3_F
#undef Q3_L
    return q3_check_launch(who);
}

extern "C" int q3asr_skinny3_gemm(const void* X16, int M, const void* W16, const float* w_scale, int N, int K, int n_splits,
                                  int epi, int waves_per_wg, int tiles_per_wave, int wg_ksplit, float* slab, void* out,
                                  int64_t ldo, void* stream) {
    return skinny3_launch("skinny3_gemm", X16, nullptr, M, W16, w_scale, N, K, n_splits, epi, waves_per_wg, tiles_per_wave, wg_ksplit,
                          slab, out, ldo, stream);
}

extern "C" int q3asr_skinny3_gemm_norm(const float* in_slab, int in_n_slabs, int64_t in_slab_stride, const void* h_in, void* h_out,
                                       const void* norm_w, float eps, int M, const void* W16, const float* w_scale, int N, int K,
                                       int n_splits, int epi, int waves_per_wg, int tiles_per_wave, int wg_ksplit, float* slab,
                                       void* out, int64_t ldo, void* stream) {
    FuseNorm f{in_slab, in_n_slabs, (long)in_slab_stride, (const bf16*)h_in, (bf16*)h_out, (const bf16*)norm_w, eps};
    return skinny3_launch("skinny3_gemm_norm", nullptr, &f, M, W16, w_scale, N, K, n_splits, epi, waves_per_wg, tiles_per_wave,
                          wg_ksplit, slab, out, ldo, stream);
}

// ---------------------------------------------------------------------------
// skinny4: the same product on the same layouts, one fat workgroup per TPW consecutive weight tiles.
// grid (N/16/TPW, S), WAVES waves; wave w of slab s owns the k-steps [(s WAVES + w) NKS, +NKS) of every tile of the
// workgroup, so K is cut into S*WAVES contiguous segments of NKS steps.  A wave issues ALL of its loads -- its X fragments
// (L2) and its NKS*TPW weight fragments (HBM, non-temporal) -- before its first MFMA: one memory round trip per launch
// instead of skinny3's two-group pipeline, and the whole matrix is in flight from the first microsecond.  The waves'
// partial sums meet in LDS and are added in wave order by whichever wave owns the (tile, m-tile) pair, so every output is
//     slab_s[m][n] = ((p_{s,0} + p_{s,1}) + ...) + p_{s,WAVES-1},  p = one k-ascending MFMA chain over NKS steps,
// a function of (NKS, WAVES, S) only: the batch size (MT) and TPW never change a bit.  M > 32 runs the m-tiles in two
// passes over the same weight registers.  tools/micro/sk4.hip (graph of 28 dependent launches over distinct matrices, batch
// 32, us per launch incl. the boundary, skinny3 -> skinny4): qkv 6.7 -> 5.4 (two slabs instead of four: 5.7, and faster
// inside the step because the attention prologue adds fewer), o 5.1 -> 3.9, gate/up 13.4 -> 11.8-12.2, down 8.4 -> 6.9, tied
// head 114.7 -> 109.5; batch 1: 5.6 -> 4.7, 4.5 -> 3.5, 12.3 -> 10.8, 7.2 -> 6.0, 100.3 -> 92.7; an empty dependent launch
// costs 1.76 (tools/micro/empty.hip).
// ---------------------------------------------------------------------------
// FUSE (M <= 2): the X fragments come from LDS rows that every workgroup builds for itself from the previous GEMM's slabs
// (fuse_norm_prologue, the arithmetic of slab_resnorm_x16_kernel) while its weights are in flight.
template <int MT, int TPW, int NKS, int EPI, int WAVES, bool F8, bool FUSE = false>
__global__ __launch_bounds__(64 * WAVES) void skinny4_kernel(const bf16* __restrict__ X16, const void* __restrict__ Wt, int M, int N, int K,
                                                             const float* __restrict__ w_scale, float* __restrict__ slab,
                                                             bf16* __restrict__ out, long ldo, const FuseNorm fz, int xcd_map) {
    static_assert(!F8 || NKS % 2 == 0, "fp8 fragments come in pairs of k steps");
    static_assert(EPI != 2 || TPW % 2 == 0, "SwiGLU needs whole (gate, up) tile pairs per workgroup");
    static_assert(!FUSE || (MT == 1 && WAVES >= 4), "fused norm: M <= 2, at least 256 threads");
    extern __shared__ f32x4 red4[];                          // [WAVES][TPW][MT][64] | FUSE: x rows [2][K] bf16 | K floats
    __shared__ float fred4[4];
    // m-tiles per pass.  The tied head at 17-32 rows takes its two m-tiles one after the other: 113 instead of 137 VGPRs, so two
    // workgroups share a CU and one's loads fly while the other reduces and stores (106.2 -> 101.6 us at 32 rows; same sums).
    // Three m-tiles (33-48 rows): one pass with all three (the weight fragments are converted / consumed once), except for the head.
    constexpr int MC = (MT < 2 || (EPI == 1 && MT <= 3)) ? 1 : (MT == 3 ? 3 : 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bx = blockIdx.x;
    if (EPI == 1 && xcd_map) {
        // The tied head writes 32 * TPW bytes per row and workgroup: half a 128-byte line at TPW = 2.  Consecutive workgroups
        // go to consecutive XCDs (each with an L2 of its own), so the two halves of a line would be written back separately,
        // as partial lines (0.33 us per logits row measured, 0.12 with whole lines).  Workgroup b therefore takes tile group
        // (b % 8) * (grid / 8) + b / 8: neighbours in n are neighbours in time on ONE XCD and their halves meet in its L2.
        // The launcher rounds the grid up to a multiple of 8; the surplus workgroups leave here, before any barrier.
        bx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (bx * TPW >= (N >> 4)) return;
    }
    const int t0 = bx * TPW;
    const int ks0 = (blockIdx.y * WAVES + wave) * NKS;
    const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NW = F8 ? NKS / 2 : NKS;                   // weight loads per tile
    bf16x8 w16[F8 ? 1 : NW][TPW];
    u32x4 w8[F8 ? NW : 1][TPW];
    bf16x8 xf[NKS][MC];
    if constexpr (!FUSE) {
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int mc = 0; mc < MC; ++mc) xf[i][mc] = *reinterpret_cast<const bf16x8*>(xp + ((long)i * MT + mc) * 512);
    }
    if constexpr (!F8) {
        const bf16* wp = (const bf16*)Wt + ((long)t0 * (K >> 5) + ks0) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                w16[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ((long)t * (K >> 5) + i) * 512));
    } else {
        const char* wp = (const char*)Wt + ((long)t0 * (K >> 6) + (ks0 >> 1)) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                w8[i][t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + ((long)t * (K >> 6) + i) * 1024));
    }
    __builtin_amdgcn_sched_barrier(0);                       // every load above is issued before the first MFMA below
    if constexpr (FUSE) {
        bf16* xl = reinterpret_cast<bf16*>(red4 + WAVES * TPW * MT * 64);
        float* hl = reinterpret_cast<float*>(reinterpret_cast<char*>(xl) + (size_t)K * 4);
        fuse_norm_prologue(fz, M, K, xl, hl, fred4, blockIdx.x == 0 && blockIdx.y == 0);
        const int xrow = min(lane & 15, M - 1) * K + 8 * (lane >> 4);      // lanes of rows >= M re-read the last row (results unused)
#pragma unroll
        for (int i = 0; i < NKS; ++i) xf[i][0] = *reinterpret_cast<const bf16x8*>(xl + xrow + (ks0 + i) * 32);
    }
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
        if constexpr (!F8) {
#pragma unroll
            for (int i = 0; i < NKS; ++i)
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int mc = 0; mc < MC; ++mc) acc[t][mc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w16[i][t], xf[i][mc], acc[t][mc], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const bf16x8 wf = h == 0 ? fp8x8_to_bf16(w8[i][t][0], w8[i][t][1]) : fp8x8_to_bf16(w8[i][t][2], w8[i][t][3]);
#pragma unroll
                        for (int mc = 0; mc < MC; ++mc) acc[t][mc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[2 * i + h][mc], acc[t][mc], 0, 0, 0);
                    }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mc = 0; mc < MC; ++mc) red4[((wave * TPW + t) * MT + m0 + mc) * 64 + lane] = acc[t][mc];
    }
    __syncthreads();
    // lane (g, li) of the (tile t, m-tile mt) image holds C[n = 16 (t0 + t) + 4 g + r][m = 16 mt + li]
    const int g = lane >> 4, li = lane & 15;
    auto total = [&](int t, int mt) {
        f32x4 a = red4[(t * MT + mt) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < WAVES; ++ww) a += red4[((ww * TPW + t) * MT + mt) * 64 + lane];
        if (F8) a *= *reinterpret_cast<const f32x4*>(w_scale + 16 * (t0 + t) + 4 * g);
        return a;
    };
    if (EPI == 2) {
        for (int c = wave; c < (TPW / 2) * MT; c += WAVES) {
            const int p = c / MT, mt = c - p * MT;           // tile t0 + 2p = gate rows, t0 + 2p + 1 = up rows of 16 features
            const f32x4 a = total(2 * p, mt), b = total(2 * p + 1, mt);
            const int m = 16 * mt + li;
            if (m >= M) continue;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float av = round_bf(a[r]), bv = round_bf(b[r]);                   // bf16 gate/up (decoder.py:97)
                o[r] = swiglu_bf(av, bv);
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
        const int n = 16 * (t0 + t) + 4 * g;
        if (EPI == 0) {
            *reinterpret_cast<f32x4*>(slab + ((long)blockIdx.y * M + m) * N + n) = a;
        } else {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(a[r]);
            *reinterpret_cast<bf16x4*>(out + (long)m * ldo + n) = o;
        }
    }
}

// ---------------------------------------------------------------------------
// The tied head as one workgroup per CU that walks its tile groups (bf16 weights, K = 2,048, the (2, 8, 8) geometry of
// skinny4_kernel: 2 weight tiles per group, 8 waves x 8 k-steps).  A slot of skinny4_kernel runs ~9 workgroups per launch of
// the head, each re-reading its X fragments from L2 (128 KB at 17-32 rows, as much as its weights) and paying a workgroup's
// start and drain; here the X fragments stay in registers, the next group's weights are requested before the current
// group's MFMAs (two register buffers), and the partial-sum image in LDS is double-buffered so that one barrier per group
// suffices.  Same sums in the same order as skinny4_kernel<MT, 2, 8, 1, 8>: bit-identical logits.
// Workgroup b serves XCD b % 8's contiguous share of the groups (see the XCD note in skinny4_kernel), taking groups
// b / 8, b / 8 + gridDim.x / 8, ... of it, so neighbouring groups are written at about the same time through one L2.
// ---------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(512) void head_stream_kernel(const bf16* __restrict__ X16, const bf16* __restrict__ W16, int M, int N,
                                                          bf16* __restrict__ out, long ldo) {
    constexpr int TPW = 2, NKS = 8, WAVES = 8, KS = 64;     // KS: k-steps of 32 in K = 2,048
    extern __shared__ f32x4 redp[];                          // [2][WAVES][TPW][MT][64]
    constexpr int IMG = WAVES * TPW * MT * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int ks0 = wave * NKS;
    bf16x8 xf[NKS][MT];
    {
        const bf16* xp = X16 + (long)ks0 * MT * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int mc = 0; mc < MT; ++mc) xf[i][mc] = *reinterpret_cast<const bf16x8*>(xp + ((long)i * MT + mc) * 512);
    }
    const int P = (N >> 4) / TPW, per = (P + 7) >> 3;
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
    int j = blockIdx.x >> 3;                                 // index inside the XCD's share
    bf16x8 wA[NKS][TPW], wB[NKS][TPW];
    auto load = [&](bf16x8 (&w)[NKS][TPW], int jj) {
        const bf16* wp = W16 + ((long)(xcd * per + jj) * TPW * KS + ks0) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < NKS; ++i)
#pragma unroll
            for (int t = 0; t < TPW; ++t) w[i][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ((long)t * KS + i) * 512));
    };
    auto finish = [&](const bf16x8 (&w)[NKS][TPW], int jj, int buf) {
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
                for (int mc = 0; mc < MT; ++mc) acc[t][mc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][t], xf[i][mc], acc[t][mc], 0, 0, 0);
        f32x4* red = redp + buf * IMG;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int mc = 0; mc < MT; ++mc) red[((wave * TPW + t) * MT + mc) * 64 + lane] = acc[t][mc];
        __syncthreads();                                     // the only barrier of a group: the other image is being filled meanwhile
        const int t0 = (xcd * per + jj) * TPW;
        for (int c = wave; c < TPW * MT; c += WAVES) {
            const int t = c / MT, mt = c - t * MT;
            f32x4 a = red[(t * MT + mt) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) a += red[((ww * TPW + t) * MT + mt) * 64 + lane];
            const int m = 16 * mt + li;
            if (m >= M) continue;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(a[r]);
            *reinterpret_cast<bf16x4*>(out + (long)m * ldo + 16 * (t0 + t) + 4 * g) = o;
        }
    };
    // Loads are unconditional (past the end they re-read the share's last group and are dropped): the compiler then counts
    // vmcnt exactly, and the MFMAs of a group wait for that group's 16 loads only, with the next group's 16 in flight.
    const int n_mine = min(per, P - xcd * per);              // groups in this XCD's share
    const int cnt = n_mine > j ? (n_mine - j + nslot - 1) / nslot : 0;      // ... of which this workgroup's
    if (cnt == 0) return;                                    // (small N)
    const int last = n_mine - 1;
    load(wA, j);
    for (int k = 0; k < cnt; k += 2) {
        load(wB, min(j + nslot, last));
        __builtin_amdgcn_sched_barrier(0);                   // the next group's loads are issued before this group's MFMAs
        finish(wA, j, 0);
        load(wA, min(j + 2 * nslot, last));
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < cnt) finish(wB, j + nslot, 1);
        j += 2 * nslot;
    }
}

template <int MT, int TPW, int NKS, int EPI, int WAVES, bool F8, bool FUSE>
static void skinny4_go(dim3 grid, hipStream_t s, const void* X16, const FuseNorm& fz, const void* W, int M, int N, int K, const float* w_scale,
                       float* slab, void* out, long ldo, int xcd_map) {
    constexpr size_t image = (size_t)WAVES * TPW * MT * 64 * sizeof(f32x4);
    static_assert(image <= 160 * 1024, "skinny4: partial-sum image exceeds LDS");
    const size_t lds = image + (FUSE ? (size_t)K * 8 : 0);   // + two bf16 rows and K floats
    static std::atomic<size_t> granted{64 * 1024};   // host threads of several lanes launch concurrently: the attribute call is idempotent,
    if (lds > granted.load(std::memory_order_relaxed)) {                                      // the flag only saves repeating it
        (void)hipFuncSetAttribute((const void*)skinny4_kernel<MT, TPW, NKS, EPI, WAVES, F8, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        granted.store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((skinny4_kernel<MT, TPW, NKS, EPI, WAVES, F8, FUSE>), grid, dim3(64 * WAVES), lds, s, (const bf16*)X16, W, M, N, K, w_scale,
                       slab, (bf16*)out, ldo, fz, xcd_map);
}

// The instantiated geometries (tiles per workgroup, k-steps per wave, waves), by epilogue; ops.gemm_geometry4 holds the same
// list.  M > 32 takes the two-tile form of the 16-wave SwiGLU geometry (256 KiB of partial sums would not fit): same
// (NKS, WAVES), same bits.

static int skinny4_launch(const char* who, const void* X16, const FuseNorm* fuse, int M, const void* W16, const float* w_scale, int N, int K, int epi,
                          int tiles_per_wg, int ksteps_per_wave, int waves, float* slab, void* out, int64_t ldo, void* stream) {
    Q3_REQUIRE((X16 || fuse) && W16, "%s: null pointer", who);
    Q3_REQUIRE(M > 0 && M <= 64 && N > 0 && N % 16 == 0 && K > 0, "%s: M=%d must be 1..64, N=%d a multiple of 16", who, M, N);
    Q3_REQUIRE(epi >= 0 && epi <= 2, "%s: bad epi %d", who, epi);
    const int seg = 32 * ksteps_per_wave * waves;
    Q3_REQUIRE(seg > 0 && tiles_per_wg > 0 && K % seg == 0 && (N / 16) % tiles_per_wg == 0, "%s: K=%d not a multiple of %d or N/16=%d not of %d", who, K,
               seg, N / 16, tiles_per_wg);
    const int S = K / seg;
    if (epi == 0) Q3_REQUIRE(slab, "%s: slab epilogue without workspace", who);
    else Q3_REQUIRE(out && S == 1, "%s: direct epilogues need out and K = 32 * k-steps * waves (one slab)", who);
    if (epi == 1) Q3_REQUIRE(ldo % 4 == 0 && ldo >= N, "%s: bad ldo", who);
    if (w_scale) Q3_REQUIRE(ksteps_per_wave % 2 == 0, "%s: fp8 weights need an even number of k-steps per wave", who);
    FuseNorm fz{};
    if (fuse) {
        fz = *fuse;
        Q3_REQUIRE(M <= 2 && epi != 1, "%s: fused norm needs M <= 2, epi 0 or 2", who);
        Q3_REQUIRE(fz.slab && fz.S >= 1 && fz.h_in && fz.h_out && fz.w && fz.h_in != fz.h_out && fz.slab != slab,
                   "%s: fused norm needs its own input slabs and distinct h_in / h_out", who);
        Q3_REQUIRE(K % 8 == 0 && (size_t)waves * tiles_per_wg * 1024 + (size_t)K * 8 <= 160 * 1024, "%s: K=%d too large for the fused activation image", who, K);
    }
    const int MT = q3_mt_of(M);
    const int xcd_map = epi == 1 && g_q3_option[Q3ASR_OPT_HEAD_XCD_MAP];
    dim3 grid(xcd_map ? (N / 16 / tiles_per_wg + 7) / 8 * 8 : N / 16 / tiles_per_wg, S);   // see the XCD note in the kernel
    hipStream_t s = (hipStream_t)stream;
    bool done = false;
    if (epi == 1 && !fuse && !w_scale && K == 2048 && tiles_per_wg == 2 && ksteps_per_wave == 8 && waves == 8 && MT == 2 &&
        g_q3_option[Q3ASR_OPT_HEAD_STREAM]) {
        // the tied head at two m-tiles: persistent form of this geometry (head_stream_kernel), bit-identical; at one m-tile the
        // per-group launch is as fast (two workgroups per CU fit there: 95.2 vs 94.7 us)
        const int groups = N / 32;
        const int wgs = groups >= 256 ? 256 : (groups + 7) / 8 * 8;                  // one workgroup per CU, a multiple of 8
        const size_t lds = (size_t)2 * 8 * 2 * MT * 64 * sizeof(f32x4);
        hipLaunchKernelGGL(head_stream_kernel<2>, dim3(wgs), dim3(512), lds, s, (const bf16*)X16, (const bf16*)W16, M, N, (bf16*)out, (long)ldo);
        return q3_check_launch(who);
    }
#define Q3_TRY(MT_, TPW_, NKS_, EPI_, WAVES_, F8_, FUSE_)                                                                              \
    if (!done && MT == MT_ && tiles_per_wg == TPW_ && ksteps_per_wave == NKS_ && epi == EPI_ && waves == WAVES_ && (w_scale != nullptr) == F8_ && \
        (fuse != nullptr) == FUSE_) {                                                                                                  \
        skinny4_go<MT_, TPW_, NKS_, EPI_, WAVES_, F8_, FUSE_>(grid, s, X16, fz, W16, M, N, K, w_scale, slab, out, (long)ldo, xcd_map); \
        done = true;                                                                                                                   \
    }
#define Q3_TRY_F8(MT_, TPW_, NKS_, EPI_, WAVES_, FUSE_)                                 \
    Q3_TRY(MT_, TPW_, NKS_, EPI_, WAVES_, false, FUSE_)                                 \
    if constexpr ((NKS_) % 2 == 0) { Q3_TRY(MT_, TPW_, NKS_, EPI_, WAVES_, true, FUSE_) }
#define Q3_TRY_MT(TPW_, NKS_, WAVES_, EPI_)                                             \
    Q3_TRY_F8(1, TPW_, NKS_, EPI_, WAVES_, false)                                       \
    if constexpr ((EPI_) != 1) { Q3_TRY_F8(1, TPW_, NKS_, EPI_, WAVES_, true) }         \
    Q3_TRY_F8(2, TPW_, NKS_, EPI_, WAVES_, false)                                       \
    if constexpr ((WAVES_) * (TPW_) * 3 <= 160) { Q3_TRY_F8(3, TPW_, NKS_, EPI_, WAVES_, false) } \
    if constexpr ((WAVES_) * (TPW_) * 4 <= 160) { Q3_TRY_F8(4, TPW_, NKS_, EPI_, WAVES_, false) }
    // epilogue 0 (fp32 slabs): the projections; 1 (bf16 rows): the tied head; 2 (SwiGLU -> X16): gate/up
    Q3_TRY_MT(2, 2, 8, 0) Q3_TRY_MT(2, 4, 8, 0) Q3_TRY_MT(2, 6, 8, 0) Q3_TRY_MT(2, 3, 16, 0)
    Q3_TRY_MT(2, 2, 8, 1) Q3_TRY_MT(2, 8, 8, 1) Q3_TRY_MT(4, 8, 8, 1)
    Q3_TRY_MT(2, 2, 8, 2) Q3_TRY_MT(4, 4, 16, 2) Q3_TRY_MT(2, 4, 16, 2)
#undef Q3_TRY_MT
#undef Q3_TRY_F8
#undef Q3_TRY
    Q3_REQUIRE(done, "%s: geometry (tiles %d, k-steps %d, waves %d, epi %d, M %d%s) is not instantiated", who, tiles_per_wg, ksteps_per_wave, waves, epi,
               M, w_scale ? ", fp8" : "");
    return q3_check_launch(who);
}

extern "C" int q3asr_skinny4_gemm(const void* X16, int M, const void* W16, const float* w_scale, int N, int K, int epi, int tiles_per_wg,
                                  int ksteps_per_wave, int waves, float* slab, void* out, int64_t ldo, void* stream) {
    return skinny4_launch("skinny4_gemm", X16, nullptr, M, W16, w_scale, N, K, epi, tiles_per_wg, ksteps_per_wave, waves, slab, out, ldo, stream);
}

extern "C" int q3asr_skinny4_gemm_norm(const float* in_slab, int in_n_slabs, int64_t in_slab_stride, const void* h_in, void* h_out,
                                       const void* norm_w, float eps, int M, const void* W16, const float* w_scale, int N, int K, int epi,
                                       int tiles_per_wg, int ksteps_per_wave, int waves, float* slab, void* out, int64_t ldo, void* stream) {
    FuseNorm f{in_slab, in_n_slabs, (long)in_slab_stride, (const bf16*)h_in, (bf16*)h_out, (const bf16*)norm_w, eps};
    return skinny4_launch("skinny4_gemm_norm", nullptr, &f, M, W16, w_scale, N, K, epi, tiles_per_wg, ksteps_per_wave, waves, slab, out, ldo, stream);
}

// ---------------------------------------------------------------------------
// Slab consumer: h = bf16(h + bf16(sum_s slab_s)); x = RMSNorm(h) * w written in X16 layout
// (decoder.py:191,194 / :196,189 / :251).  One workgroup per row; a thread owns 8 consecutive
// columns = one X16 fragment (one 16-byte store).
// ---------------------------------------------------------------------------
// SS > 0: the slab count is known at compile time -- every load of the thread (SS slab pieces, its residual and norm-weight
// vectors) is issued before the first add, one memory round trip instead of one per pair of slabs plus two more; the sum
// runs in the same order (slab 0, 1, 2, ...), so results are bit-identical to the SS = 0 loop.
template <int EPT, int SS = 0>   // N = 256 * EPT, EPT in {2,4,8,16} (EPT < 8: N = 512/1024 handled with 8-wide threads, fewer active)
__global__ __launch_bounds__(256) void slab_resnorm_x16_kernel(const float* __restrict__ slab, int S, long slab_stride,
                                                               const bf16* __restrict__ h_in, bf16* __restrict__ h_out,
                                                               const bf16* __restrict__ w, bf16* __restrict__ x16, int MT,
                                                               float eps) {
    __shared__ float red[4];
    constexpr int N = 256 * EPT;
    constexpr int NV = N / 8;                              // 8-wide vectors per row
    const int m = blockIdx.x;
    float q2 = 0.f;
    float hv[(NV + 255) / 256][8];
    bf16x8 wpre[(NV + 255) / 256];
    bf16x8 hpre;
#pragma unroll
    for (int it = 0; it < (NV + 255) / 256; ++it) {
        const int v = threadIdx.x + 256 * it;
        if (v < NV) {
            const long c0 = (long)m * N + 8 * v;
            f32x4 a0, a1;
            if constexpr (SS > 0) {
                f32x4 p0[SS], p1[SS];
#pragma unroll
                for (int q = 0; q < SS; ++q) {
                    p0[q] = *reinterpret_cast<const f32x4*>(slab + q * slab_stride + c0);
                    p1[q] = *reinterpret_cast<const f32x4*>(slab + q * slab_stride + c0 + 4);
                }
                wpre[it] = *reinterpret_cast<const bf16x8*>(w + 8 * v);
                hpre = *reinterpret_cast<const bf16x8*>(h_in + c0);
                a0 = p0[0]; a1 = p1[0];
#pragma unroll
                for (int q = 1; q < SS; ++q) { a0 += p0[q]; a1 += p1[q]; }
            } else {
            a0 = *reinterpret_cast<const f32x4*>(slab + c0); a1 = *reinterpret_cast<const f32x4*>(slab + c0 + 4);
            int s = 1;
            for (; s + 1 < S; s += 2) {                    // two slabs per trip: independent loads in flight
                const float* p = slab + s * slab_stride + c0;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p), b1 = *reinterpret_cast<const f32x4*>(p + 4);
                const f32x4 c0v = *reinterpret_cast<const f32x4*>(p + slab_stride), c1v = *reinterpret_cast<const f32x4*>(p + slab_stride + 4);
                a0 += b0; a1 += b1; a0 += c0v; a1 += c1v;
            }
            if (s < S) {
                const float* p = slab + s * slab_stride + c0;
                a0 += *reinterpret_cast<const f32x4*>(p);
                a1 += *reinterpret_cast<const f32x4*>(p + 4);
            }
            hpre = *reinterpret_cast<const bf16x8*>(h_in + c0);
            }
            const bf16x8 hi = hpre;
            bf16x8 ho;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = round_bf(round_bf(e < 4 ? a0[e] : a1[e - 4]) + bf2f(hi[e]));
                hv[it][e] = t;
                q2 = fmaf(t, t, q2);
                ho[e] = f2bf(t);
            }
            *reinterpret_cast<bf16x8*>(h_out + c0) = ho;
        }
    }
    q2 = wave_sum(q2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q2;
    __syncthreads();
    const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) * (1.0f / N) + eps);
#pragma unroll
    for (int it = 0; it < (NV + 255) / 256; ++it) {
        const int v = threadIdx.x + 256 * it;
        if (v < NV) {
            bf16x8 wv;
            if constexpr (SS > 0) wv = wpre[it];
            else wv = *reinterpret_cast<const bf16x8*>(w + 8 * v);
            bf16x8 xo;
#pragma unroll
            for (int e = 0; e < 8; ++e) xo[e] = f2bf(hv[it][e] * r * bf2f(wv[e]));
            *reinterpret_cast<bf16x8*>(x16 + x16_index(m, 8 * v, MT)) = xo;
        }
    }
}

extern "C" int q3asr_slab_resnorm_x16(const float* slab, int n_slabs, int64_t slab_stride, const void* h_in, void* h_out,
                                      const void* w, void* x16, int M, int N, float eps, void* stream) {
    Q3_REQUIRE(slab && h_in && h_out && w && x16, "slab_resnorm_x16: null pointer");
    Q3_REQUIRE(M > 0 && M <= 64 && n_slabs > 0 && (N == 512 || N == 1024 || N == 2048 || N == 4096), "slab_resnorm_x16: M=%d N=%d unsupported", M, N);
    const int MT = q3_mt_of(M);
    dim3 grid(M), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16* hi = (const bf16*)h_in; bf16* ho = (bf16*)h_out; const bf16* ww = (const bf16*)w; bf16* xo = (bf16*)x16;
    switch (N / 256) {
        case 2: hipLaunchKernelGGL(slab_resnorm_x16_kernel<2>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, MT, eps); break;
        case 4: hipLaunchKernelGGL(slab_resnorm_x16_kernel<4>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, MT, eps); break;
        case 8:                                                       // hidden size 2048: the decode step's shape
            if (n_slabs == 4) hipLaunchKernelGGL((slab_resnorm_x16_kernel<8, 4>), grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, MT, eps);
            else if (n_slabs == 8) hipLaunchKernelGGL((slab_resnorm_x16_kernel<8, 8>), grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, MT, eps);
            else if (n_slabs == 2) hipLaunchKernelGGL((slab_resnorm_x16_kernel<8, 2>), grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, MT, eps);
            else hipLaunchKernelGGL(slab_resnorm_x16_kernel<8>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, MT, eps);
            break;
        default: hipLaunchKernelGGL(slab_resnorm_x16_kernel<16>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, MT, eps); break;
    }
    return q3_check_launch("slab_resnorm_x16");
}

// h = E[tok[m]] (or a given bf16 row when tok == nullptr); x16 = RMSNorm(h) * w   (generate.py:294 + decoder.py:189)
// Row ids come from device memory the C-ABI cannot validate: an id outside [0, V) gathers nothing (h = 0, x = 0) instead of
// reading out of bounds -- the policy of the cache-row writes (fault records).
template <int EPT>
__global__ __launch_bounds__(256) void embed_norm_x16_kernel(const int* __restrict__ tok, const bf16* __restrict__ E,
                                                             bf16* __restrict__ h_out, const bf16* __restrict__ w,
                                                             bf16* __restrict__ x16, int MT, float eps, int V) {
    __shared__ float red[4];
    constexpr int N = 256 * EPT;
    constexpr int NV = N / 8;
    const int m = blockIdx.x;
    const int id = tok ? tok[m] : m;
    const bool ok = (unsigned)id < (unsigned)V;                        // uniform per workgroup
    const bf16* src = E + (long)(ok ? id : 0) * N;
    float q2 = 0.f;
    float hv[(NV + 255) / 256][8];
#pragma unroll
    for (int it = 0; it < (NV + 255) / 256; ++it) {
        const int v = threadIdx.x + 256 * it;
        if (v < NV) {
            bf16x8 hi = bf16x8{};
            if (ok) hi = *reinterpret_cast<const bf16x8*>(src + 8 * v);
            *reinterpret_cast<bf16x8*>(h_out + (long)m * N + 8 * v) = hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) { hv[it][e] = bf2f(hi[e]); q2 = fmaf(hv[it][e], hv[it][e], q2); }
        }
    }
    q2 = wave_sum(q2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q2;
    __syncthreads();
    const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) * (1.0f / N) + eps);
#pragma unroll
    for (int it = 0; it < (NV + 255) / 256; ++it) {
        const int v = threadIdx.x + 256 * it;
        if (v < NV) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + 8 * v);
            bf16x8 xo;
#pragma unroll
            for (int e = 0; e < 8; ++e) xo[e] = f2bf(hv[it][e] * r * bf2f(wv[e]));
            *reinterpret_cast<bf16x8*>(x16 + x16_index(m, 8 * v, MT)) = xo;
        }
    }
}

extern "C" int q3asr_embed_norm_x16(const int32_t* tok, const void* E, int V, void* h_out, const void* w, void* x16, int M, int N,
                                    float eps, void* stream) {
    Q3_REQUIRE(E && h_out && w && x16, "embed_norm_x16: null pointer");
    Q3_REQUIRE(M > 0 && M <= 64 && (N == 512 || N == 1024 || N == 2048 || N == 4096), "embed_norm_x16: M=%d N=%d unsupported", M, N);
    Q3_REQUIRE(V > 0 && (tok || V >= M), "embed_norm_x16: E has V=%d rows (M=%d rows are read when tok is NULL)", V, M);
    const int MT = q3_mt_of(M);
    dim3 grid(M), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16* e = (const bf16*)E; bf16* ho = (bf16*)h_out; const bf16* ww = (const bf16*)w; bf16* xo = (bf16*)x16;
    switch (N / 256) {
        case 2: hipLaunchKernelGGL(embed_norm_x16_kernel<2>, grid, blk, 0, s, tok, e, ho, ww, xo, MT, eps, V); break;
        case 4: hipLaunchKernelGGL(embed_norm_x16_kernel<4>, grid, blk, 0, s, tok, e, ho, ww, xo, MT, eps, V); break;
        case 8: hipLaunchKernelGGL(embed_norm_x16_kernel<8>, grid, blk, 0, s, tok, e, ho, ww, xo, MT, eps, V); break;
        default: hipLaunchKernelGGL(embed_norm_x16_kernel<16>, grid, blk, 0, s, tok, e, ho, ww, xo, MT, eps, V); break;
    }
    return q3_check_launch("embed_norm_x16");
}

// ---------------------------------------------------------------------------
// Decode attention with the q/k RMSNorm + RoPE + KV append fused into its prologue
// (decoder.py:140-163 at T = 1).  The qkv projection arrives as S fp32 split-K slabs [S][B][(nh+2nkv)*128];
// their sum is rounded to bf16 (the projection's output dtype).  The attention output (the o
// projection's operand) is written in X16 layout.
//
// Workgroup (kv head, sequence, group of 1024-token chunks of the cache).  Waves 0,1 normalise/rotate the two query
// heads; in the chunk that holds the new token waves 2,3 do the same for k and pass v through, and the new
// k/v go to the cache AND to LDS (the token attends to itself without a global round trip).  A chunk is
// swept in 32-key MFMA tiles, wave w owning tiles w, w + 8, ... (the first one issued before the prologue); the
// waves' online-softmax partials are merged through LDS.
// Sequences longer than one chunk: every chunk writes (max, sum, numerator[128]) per head to the workspace,
// and the LAST workgroup to arrive (device-scope counter) merges the chunks in chunk order, so the result
// depends only on the sequence, not on the batch, the cache capacity or the arrival order.
// Chunks past the sequence's length exit at once: the grid is sized for the cache capacity because the
// launch is captured in a graph while positions advance.
// ---------------------------------------------------------------------------
// Chunk = the unit of the reduction tree (a function of the sequence only, never of the batch or the launch geometry, so
// results are bit-identical for any batch / cache capacity / chunks-per-workgroup choice).  1,024 tokens.  A workgroup
// that owns all chunks of its sequence keeps the per-chunk partials in LDS (same merge, same order, no workspace trip).
// Tried (round 2): 256-token chunks, one per workgroup at small batch (24 workgroups instead of 8 for a batch-1 step at
// a 500-token context): batch-1 latency 180.5 vs 180.0 ms -- the dependent chain (slab sum -> QK-norm -> RoPE -> one tile
// per wave -> merge) is what a batch-1 launch costs, and the cross-workgroup merge adds a ticket round trip -- and the
// batch-32 step lost 3.7 % to the extra per-chunk barriers.
#define ADF_CHUNK 1024
#define ADF_LOCAL_MAX 8               // chunks whose partials fit the LDS array below
#define ADF_PART (2 * 130)            // floats per (sequence, kv head, chunk): per head 128 numerators, max, sum

// 256-byte rows of a V tile in LDS: 16-byte chunk ch of row r sits at 256 r + 16 (ch ^ f(r)); conflict-free for the row
// writes and for the transposing reads (same image as the prefill kernel, attn.hip)
__device__ __forceinline__ int adm_off(int row, int ch) { return 256 * row + 16 * (ch ^ ((row & 7) << 1)); }   // round 4: conflict-free (attn.hip, TileImg<128>)

#ifdef ADF_TIMING                     // tools/micro/adf_time.hip: wall_clock64 stamps of wave 0 of workgroup (0, 0, 0)
__device__ long long g_adf_t[16];
#define ADF_T(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_adf_t[i] = wall_clock64(); } while (0)
#else
#define ADF_T(i)
#endif
// NT: the cached keys / values are read with non-temporal loads (a cache that is streamed once per step).
// W4: a workgroup of FOUR waves computes the eight per-wave partials of a chunk -- wave w those of the "logical" waves w and w + 4,
// one after the other, each with its own online-softmax state, written to the same s_m / s_l / s_acc slots and merged by the
// same code: bit-identical to the eight-wave form.  Round 5, for launches with more workgroups than CUs (33-64 rows: 264-512
// (sequence, kv head) pairs): at 220 VGPRs an eight-wave workgroup owns its CU, so 320 of them ran as a full round and a
// quarter-filled one (23.0 us per launch at 40 rows against 13.6 at 32); two four-wave workgroups share a CU.
template <bool NT, bool W4>
// Parameter order: the first 16 dwords are handed over in SGPRs at dispatch (kernarg preload, build.py); they are what the loads of
// the prologue and of the first key tile need (slab addresses; the position and the cache rows), the rest is fetched beside them.
__global__ __launch_bounds__(W4 ? 256 : 512, 2) void attn_decode_fused_kernel(const float* __restrict__ qkv, long ld, long slab_stride,
                                                                const int* __restrict__ pos_arr,
                                                                bf16* __restrict__ k_cache, bf16* __restrict__ v_cache,
                                                                int n_slabs, int nh, int nkv, int Lmax, int G, int nc, int MT,
                                                                float scale, float eps,
                                                                const bf16* __restrict__ qn_w, const bf16* __restrict__ kn_w,
                                                                const float* __restrict__ inv_freq, bf16* __restrict__ out16,
                                                                float* part, int* counters) {
    constexpr int NP = 8;                                             // partials merged per chunk: one per wave
    __shared__ __attribute__((aligned(16))) bf16 s_qb[2][128];
    __shared__ __attribute__((aligned(16))) bf16 s_k[128], s_v[128];
    __shared__ float s_m[NP][2], s_l[NP][2];
    __shared__ __attribute__((aligned(16))) float s_acc[NP][2][128];
    __shared__ int s_ticket;
    __shared__ float s_part[ADF_LOCAL_MAX][ADF_PART];                  // per-chunk partials of a workgroup that owns the whole sequence
    extern __shared__ __attribute__((aligned(16))) char adm_tiles[];   // one 32-key V tile (8 KiB) per wave
    const int kvh = blockIdx.x, b = blockIdx.y;
    ADF_T(0);
    int pos = pos_arr[b];                                             // the head of the longest chain (position -> key tile -> scores): requested first,
    asm volatile("" : "+s"(pos) :: "memory");                         // consumed below, after the slab loads have been issued
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    // Round 5: the prologue's own loads are requested BEFORE the position is known.  The position decides which workgroups
    // have work and which one owns the new token, and every later statement used to sit behind that test: a scalar load from a
    // cold L2, then the slab loads, two dependent memory round trips in front of the norm (stamped, tools/micro/adf_time.hip:
    // slabs summed 3.2 us after entry at batch 1, 5.3 us at batch 32).  The slab values of all four prologue waves do not
    // depend on the position: waves 2 and 3 request their k / v columns whether or not this workgroup turns out to own the new
    // token, and a workgroup without work leaves with 16 loads per lane outstanding (64 KiB per workgroup at most).  A wave's
    // loads return in order, so they also stay ahead of the first 64 KiB of keys and values requested below.
    const int col = wave < 2 ? (2 * kvh + wave) * 128 : (wave == 2 ? (nh + kvh) * 128 : (nh + nkv + kvh) * 128);
    const float* src = qkv + (long)b * ld + col;
    const bf16* nw = wave < 2 ? qn_w : kn_w;
    float p1[8], p2[8];
    if (wave < 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {                          // 8 slabs: 16 independent loads in flight
            const bool in = u < n_slabs;
            p1[u] = in ? src[u * slab_stride + lane] : 0.f;
            p2[u] = in ? src[u * slab_stride + lane + 64] : 0.f;
        }
    }
    ADF_T(9);                                                  // the slab loads are issued
    bf16 nw1 = nw[lane], nw2 = nw[lane + 64];
    const float freq = inv_freq[lane];
    ADF_T(10);
    const int L = pos + 1, n_active = (L + ADF_CHUNK - 1) / ADF_CHUNK;
    const int c_begin = blockIdx.z * G, c_end = min(c_begin + G, n_active);   // this workgroup's chunks
    if (c_begin >= n_active) return;                                  // whole workgroup: no barrier is skipped
    const int n_groups = (n_active + G - 1) / G;                      // workgroups active for this (sequence, kv head)
    const bool local = n_groups == 1 && n_active <= ADF_LOCAL_MAX;    // uniform: partials stay in LDS, no workspace round trip
    const bool owner = c_end == n_active;                             // the last chunk holds the new token
    bf16* Kb = k_cache + ((long)b * nkv + kvh) * Lmax * 128;
    bf16* Vb = v_cache + ((long)b * nkv + kvh) * Lmax * 128;
    const int t0 = c_begin * ADF_CHUNK;
    const int t_last = pos > 0 ? pos - 1 : 0;                         // tokens >= pos are never read from the cache
    const bool pro = wave < 2 || (wave < 4 && owner);
    // One register set for a 32-key tile (K as A fragments, V rows on their way to LDS); the next tile's loads
    // are issued as soon as this tile's K fed the MFMAs and its V sits in LDS.  The first tile is issued here, ahead of
    // the prologue's arithmetic.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int li = i;
    bf16x8 kR[2][4];
    u32x4 vR[8];
#define ADM_LD(P) (NT ? __builtin_nontemporal_load(P) : *(P))
#define ADM_LOAD(KR, VR, TB)                                                                                    \
    _Pragma("unroll") for (int k4 = 0; k4 < 2; ++k4) {                                                      \
        const long kr_ = (long)min((TB) + 16 * k4 + li, t_last) * 128 + 8 * g;                              \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) KR[k4][ks] = ADM_LD(reinterpret_cast<const bf16x8*>(Kb + kr_ + 32 * ks)); \
    }                                                                                                       \
    _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                      \
        const int c_ = lane + 64 * it;                                                                      \
        VR[it] = ADM_LD(reinterpret_cast<const u32x4*>(Vb + (long)min((TB) + (c_ >> 4), t_last) * 128 + 8 * (c_ & 15))); \
        }
    if (t0 + 32 * wave < min(L, t0 + ADF_CHUNK)) { ADM_LOAD(kR, vR, t0 + 32 * wave) }
    ADF_T(11);                                                 // the position has arrived and the first key tile is requested
    if (pro) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += p1[u]; s2 += p2[u]; }          // fixed order s = 0, 1, ...
        for (int s0 = 8; s0 < n_slabs; s0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = s0 + u < n_slabs;
                p1[u] = in ? src[(s0 + u) * slab_stride + lane] : 0.f;
                p2[u] = in ? src[(s0 + u) * slab_stride + lane + 64] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += p1[u]; s2 += p2[u]; }
        }
        const float x1 = round_bf(s1), x2 = round_bf(s2);
        ADF_T(1);                                                  // the slab values have arrived and are summed
        if (wave == 3) {
            const bf16 v1 = f2bf(x1), v2 = f2bf(x2);
            s_v[lane] = v1; s_v[lane + 64] = v2;
            Vb[(long)pos * 128 + lane] = v1; Vb[(long)pos * 128 + lane + 64] = v2;
        } else {
            const float ss = wave_sum(x1 * x1 + x2 * x2) * (1.0f / 128.0f);
            const float r = rsqrtf(ss + eps);
            const float y1 = round_bf(x1 * r * bf2f(nw1)), y2 = round_bf(x2 * r * bf2f(nw2));
            float sn, cs;
            sincosf((float)pos * freq, &sn, &cs);
            const float o1 = round_bf(y1 * cs - y2 * sn), o2 = round_bf(y1 * sn + y2 * cs);
            if (wave < 2) {
                s_qb[wave][lane] = f2bf(o1); s_qb[wave][lane + 64] = f2bf(o2);
            }
            else {
                s_k[lane] = f2bf(o1); s_k[lane + 64] = f2bf(o2);
                Kb[(long)pos * 128 + lane] = f2bf(o1); Kb[(long)pos * 128 + lane + 64] = f2bf(o2);
            }
        }
    }
    // The first batch of cached keys/values does not depend on the prologue: it was issued above and is
    // in flight across this barrier.
    ADF_T(2);                                                      // norm + RoPE done (wave 0)
    __syncthreads();
    ADF_T(3);
    {
        // ---- S^T = K Q^T and O^T += V^T P^T on v_mfma_f32_16x16x32_bf16, the two heads of this kv head being two
        // of the 16 query columns (prefill kernel's formulation, attn.hip).  Wave w owns the 32-key tiles w, w+8, ... of a
        // chunk: K goes global -> A fragment (lane (li, g): key li, dims 32 ks + 8 g), V through a wave-private LDS tile
        // and ds_read_b64_tr_b16; the next tile's K/V loads are issued between this tile's QK and its softmax / PV.
        const int q4 = li >> 2, p4 = li & 3;
        char* sVw = adm_tiles + wave * 8192;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = bf16x8{};
            if (li < 2) qf[ks] = *reinterpret_cast<const bf16x8*>(&s_qb[li][32 * ks + 8 * g]);
        }
#define ADM_STAGE_QK(KR, VR, TB)                                                                                \
        {                                                                                                       \
            if ((TB) + 32 > pos) { /* the tile holding the new token: its K/V come from LDS, not from HBM */    \
                _Pragma("unroll") for (int k4 = 0; k4 < 2; ++k4)                                                \
                    if ((TB) + 16 * k4 + li == pos)                                                             \
                        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) KR[k4][ks] = *reinterpret_cast<const bf16x8*>(s_k + 32 * ks + 8 * g); \
                _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                              \
                    const int c_ = lane + 64 * it;                                                              \
                    if ((TB) + (c_ >> 4) == pos) VR[it] = *reinterpret_cast<const u32x4*>(s_v + 8 * (c_ & 15)); \
                    else if ((TB) + (c_ >> 4) > pos) VR[it] = u32x4{0u, 0u, 0u, 0u};   /* keys beyond the position weigh 0, but the rows fetched in \
                        their place (row t_last: at position 0 a row that was never written) may hold NaN, and 0 x NaN is not 0 */ \
                }                                                                                               \
            }                                                                                                   \
            _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                  \
                const int c_ = lane + 64 * it;                                                                  \
                *reinterpret_cast<u32x4*>(sVw + adm_off(c_ >> 4, c_ & 15)) = VR[it];                            \
            }                                                                                                   \
            _Pragma("unroll") for (int k4 = 0; k4 < 2; ++k4) {                                                  \
                sc_[k4] = f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                \
                    sc_[k4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KR[k4][ks], qf[ks], sc_[k4], 0, 0, 0);    \
            }                                                                                                   \
        }
#define ADM_SOFTMAX_PV(TB)                                                                                      \
        {                                                                                                       \
            float tmax_ = -INFINITY;                                                                            \
            _Pragma("unroll") for (int k4 = 0; k4 < 2; ++k4)                                                    \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                 \
                    const float v_ = (TB) + 16 * k4 + 4 * g + r <= pos ? sc_[k4][r] * scale : -INFINITY;        \
                    sc_[k4][r] = v_;                                                                            \
                    tmax_ = fmaxf(tmax_, v_);                                                                   \
                }                                                                                               \
            tmax_ = lane_xor16_max(tmax_);                                                                      \
            tmax_ = lane_xor32_max(tmax_);                                                                      \
            const float m_new_ = fmaxf(m_run, tmax_);          /* finite: key TB <= pos is in every tile */      \
            const float alpha_ = __expf(m_run - m_new_);                                                        \
            float psum_ = 0.f;                                                                                  \
            bf16x8 pf_;                                                                                         \
            _Pragma("unroll") for (int k4 = 0; k4 < 2; ++k4)                                                    \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                 \
                    const float pv_ = __expf(sc_[k4][r] - m_new_);                                              \
                    psum_ += pv_;                                                                               \
                    pf_[k4 * 4 + r] = f2bf(pv_);                                                                \
                }                                                                                               \
            l_run = l_run * alpha_ + psum_;                                                                     \
            m_run = m_new_;                                                                                     \
            /* round 4: the transposing V reads of four d tiles as one group in front of their four MFMAs (independent    \
               accumulators) -- read, wait, MFMA one d tile at a time exposed eight LDS round trips per key tile (attn.hip) */ \
            _Pragma("unroll") for (int d0_ = 0; d0_ < 8; d0_ += 4) {                                            \
                union { struct { s16x4 a, b; } h; bf16x8 v; } u_[4];                                            \
                _Pragma("unroll") for (int dd_ = 0; dd_ < 4; ++dd_) {                                           \
                    const int vb_ = adm_off(4 * g + q4, 2 * (d0_ + dd_) + (p4 >> 1)) + 8 * (p4 & 1);            \
                    u_[dd_].h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sVw + vb_));          \
                    u_[dd_].h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sVw + vb_ + 16 * 256)); \
                }                                                                                               \
                _Pragma("unroll") for (int dd_ = 0; dd_ < 4; ++dd_) o_[d0_ + dd_] *= alpha_;                    \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                _Pragma("unroll") for (int dd_ = 0; dd_ < 4; ++dd_)                                             \
                    o_[d0_ + dd_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u_[dd_].v, pf_, o_[d0_ + dd_], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                              \
            }                                                                                                   \
        }
        const int t_endm = min(L, c_end * ADF_CHUNK);
        const int hm = (tid >> 7) & 1, dm = tid & 127;
        float* mym = part + (((long)b * nkv + kvh) * nc) * ADF_PART;
        for (int cb = t0; cb < t_endm; cb += ADF_CHUNK) {                 // chunks of this workgroup
            const int ce = min(t_endm, cb + ADF_CHUNK);
            if (cb != t0 && cb + 32 * wave < ce) { ADM_LOAD(kR, vR, cb + 32 * wave) }   // the first chunk's first tile is already in flight
            // (Round 5, measured and not kept: the keys of a wave's next-but-one tile in flight too -- a second register set, 220 -> 254
            // VGPRs, same bits; tools/ab_step.py in one process: 1480.2 -> 1484.0 us per step at batch 32, 1118.1 -> 1129.8 at batch 1.)
#pragma unroll
            for (int half = 0; half < (W4 ? 2 : 1); ++half) {             // W4: logical waves `wave` and `wave + 4`, in that order
                const int lw = wave + 4 * half;
                f32x4 o_[8];
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) o_[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                float m_run = -INFINITY, l_run = 0.f;
                for (int tb = cb + 32 * lw; tb < ce; tb += 256) {
                    f32x4 sc_[2];
                    ADM_STAGE_QK(kR, vR, tb)
                    // the next tile of this physical wave: the logical wave's own next one, or (W4) the other logical wave's first
                    int nxt = tb + 256;
                    if (W4 && half == 0 && nxt >= ce) nxt = cb + 32 * (lw + 4);
                    if ((!W4 || half == 0 || tb + 256 < ce) && nxt < ce) { ADM_LOAD(kR, vR, nxt) }
                    ADM_SOFTMAX_PV(tb)
                    ADF_T(4 + ((tb - cb) >> 8 & 1));                       // 4, 5: this wave's first / second tile done
                }
                ADF_T(6);
                float l_tot = lane_xor16_sum(l_run);
                l_tot = lane_xor32_sum(l_tot);
                if (li < 2) {
                    if (g == 0) { s_m[lw][li] = m_run; s_l[lw][li] = l_tot; }
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
                        *reinterpret_cast<f32x4*>(&s_acc[lw][li][16 * dt + 4 * g]) = o_[dt];
                }
            }
            __syncthreads();
            ADF_T(7);                                                      // every wave's partial is in LDS
            if (tid < 256) {
                float Mx = -INFINITY, num = 0.f, den = 0.f;
                for (int w = 0; w < 8; ++w) Mx = fmaxf(Mx, s_m[w][hm]);
                for (int w = 0; w < 8; ++w) {
                    const float wgt = __expf(s_m[w][hm] - Mx);             // a wave without tiles holds (-inf, 0, 0): weight 0
                    num = fmaf(s_acc[w][hm][dm], wgt, num);
                    den = fmaf(s_l[w][hm], wgt, den);
                }
                if (n_active == 1) {
                    out16[x16_index(b, (2 * kvh + hm) * 128 + dm, MT)] = f2bf(num / den);
                } else if (local) {
                    float* dst = s_part[cb / ADF_CHUNK] + hm * 130;
                    dst[dm] = num;
                    if (dm == 0) { dst[128] = Mx; dst[129] = den; }
                } else {
                    float* dst = mym + (long)(cb / ADF_CHUNK) * ADF_PART + hm * 130;
                    __hip_atomic_store(dst + dm, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (dm == 0) {
                        __hip_atomic_store(dst + 128, Mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 129, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            ADF_T(8);                                                      // merged and stored
            if (ce < t_endm) __syncthreads();                              // the partials are rewritten by the next chunk
        }
#undef ADM_SOFTMAX_PV
#undef ADM_STAGE_QK
#undef ADM_LOAD
#undef ADM_LD
    }
    if (n_active == 1) return;                                        // uniform
    const int h = (tid >> 7) & 1, d = tid & 127;
    if (local) {                                                      // same merge, same order, operands from LDS
        __syncthreads();
        if (tid < 256) {
            float M = -INFINITY;
            for (int c = 0; c < n_active; ++c) M = fmaxf(M, s_part[c][h * 130 + 128]);
            float N = 0.f, D = 0.f;
            for (int c = 0; c < n_active; ++c) {
                const float* src = s_part[c] + h * 130;
                const float w = __expf(src[128] - M);
                N = fmaf(src[d], w, N);
                D = fmaf(src[129], w, D);
            }
            out16[x16_index(b, (2 * kvh + h) * 128 + d, MT)] = f2bf(N / D);
        }
        return;
    }
    float* my = part + (((long)b * nkv + kvh) * nc) * ADF_PART;
    // Each wave waits for its own partial stores to be acknowledged, the barrier collects the waves, and only
    // then is the ticket taken; the last workgroup to arrive merges every chunk of the sequence.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_groups > 1) {
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(counters + b * nkv + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != n_groups - 1) return;                         // uniform
    }
    if (tid < 256) {
        float M = -INFINITY;
        for (int c = 0; c < n_active; ++c)
            M = fmaxf(M, __hip_atomic_load(my + (long)c * ADF_PART + h * 130 + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float N = 0.f, D = 0.f;
        for (int c = 0; c < n_active; ++c) {                          // chunk order, whoever merges
            const float* src = my + (long)c * ADF_PART + h * 130;
            const float w = __expf(__hip_atomic_load(src + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - M);
            N = fmaf(__hip_atomic_load(src + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w, N);
            D = fmaf(__hip_atomic_load(src + 129, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w, D);
        }
        out16[x16_index(b, (2 * kvh + h) * 128 + d, MT)] = f2bf(N / D);
    }
    if (n_groups > 1 && tid == 0)
        __hip_atomic_store(counters + b * nkv + kvh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

static int g_attn_chunks_per_wg = 0;
// Test hook: 0 = automatic; any other value pins the number of 1,024-token cache chunks a workgroup sweeps (the result does not
// depend on it: per-chunk partials merged in chunk order -- tests/test_kernels_gpu.py).
extern "C" int q3asr_attn_decode_set_chunks_per_wg(int n) { g_attn_chunks_per_wg = n > 0 ? n : 0; return 0; }

// Chunks per workgroup for a launch: minimises rounds x (fixed cost + chunks x chunk time) (+ the cross-workgroup
// merge when a sequence spans several workgroups), one 512-thread workgroup per CU, at the cache's capacity.
// Any value gives the same bits (per-chunk partials, merged in chunk order); q3asr_attn_decode_set_chunks_per_wg overrides.
static int adf_chunks_per_wg(int pairs, int nc) {
    const int forced = g_attn_chunks_per_wg;
    if (forced > 0) return forced < nc ? forced : nc;
    const float fixed_us = 8.f, chunk_us = 24.f, merge_us = 5.f;
    int best = nc;
    float best_cost = 1e30f;
    for (int G = nc; G >= 1; --G) {
        const int groups = (nc + G - 1) / G;
        const int rounds = (pairs * groups + 255) / 256;
        const float cost = rounds * (fixed_us + G * chunk_us) + (groups > 1 ? merge_us : 0.f);
        if (cost < best_cost - 0.5f) { best_cost = cost; best = G; }
    }
    return best;
}

extern "C" int64_t q3asr_attn_decode_fused_workspace(int B, int nkv, int Lmax) {
    if (B <= 0 || nkv <= 0 || Lmax <= 0) return 0;
    const int64_t nc = (Lmax + ADF_CHUNK - 1) / ADF_CHUNK;
    return ((int64_t)B * nkv * nc * ADF_PART + (int64_t)B * nkv) * 4;
}

extern "C" int q3asr_attn_decode_fused(const float* qkv_slab, int64_t ld, int n_slabs, int64_t slab_stride, const void* qn_w, const void* kn_w, const float* inv_freq,
                                       float eps, void* k_cache, void* v_cache, const int32_t* pos, void* out, int B, int nh,
                                       int nkv, int head_dim, int Lmax, float scale, void* workspace, void* stream) {
    Q3_REQUIRE(qkv_slab && n_slabs > 0 && qn_w && kn_w && inv_freq && k_cache && v_cache && pos && out && workspace, "attn_decode_fused: null pointer");
    Q3_REQUIRE(head_dim == 128 && nh == 2 * nkv, "attn_decode_fused: needs head_dim 128 and 2 query heads per kv head");
    Q3_REQUIRE(B > 0 && B <= 64 && Lmax > 0 && ld >= (int64_t)(nh + 2 * nkv) * 128, "attn_decode_fused: bad sizes");
    const int MT = q3_mt_of(B);
    const int nc = (Lmax + ADF_CHUNK - 1) / ADF_CHUNK;
    Q3_REQUIRE(nc <= 65535, "attn_decode_fused: Lmax=%d too long", Lmax);
    float* part = (float*)workspace;
    int* counters = (int*)(part + (int64_t)B * nkv * nc * ADF_PART);
    const int G = adf_chunks_per_wg(B * nkv, nc);
    static std::atomic<bool> attr{false};
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_decode_fused_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192);
        (void)hipFuncSetAttribute((const void*)attn_decode_fused_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192);
        (void)hipFuncSetAttribute((const void*)attn_decode_fused_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 8192);
        (void)hipFuncSetAttribute((const void*)attn_decode_fused_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 8192);
        attr = true;
    }
    // K/V larger than the caches can keep from one step to the next are streamed with non-temporal loads (batch 32, 512-token
    // capacity: 67 MB per layer; 252 -> 250 ms per request); a small cache stays resident in L2 / Infinity Cache between steps
    // and default-policy loads are faster there (batch 1: 162.4 vs 167.2 ms).
    const bool nt = (int64_t)B * nkv * Lmax * 512 > (16LL << 20);
    // More workgroups than the chip has CUs (33-64 rows): four-wave workgroups, two per CU (see W4 at the kernel); same bits.
    const int groups = (nc + G - 1) / G;
    const bool w4 = (long)nkv * B * groups > 256 && g_q3_option[Q3ASR_OPT_ATTN_HALF_WG];
#define ADF_GO(NT_, W4_)                                                                                                                              \
    hipLaunchKernelGGL((attn_decode_fused_kernel<NT_, W4_>), dim3(nkv, B, groups), dim3(W4_ ? 256 : 512), (W4_ ? 4 : 8) * 8192, (hipStream_t)stream, \
                       qkv_slab, (long)ld, (long)slab_stride, pos, (bf16*)k_cache, (bf16*)v_cache, n_slabs, nh, nkv, Lmax, G, nc, MT, scale, eps,      \
                       (const bf16*)qn_w, (const bf16*)kn_w, inv_freq, (bf16*)out, part, counters)
    if (nt) { if (w4) ADF_GO(true, true); else ADF_GO(true, false); }
    else { if (w4) ADF_GO(false, true); else ADF_GO(false, false); }
#undef ADF_GO
    return q3_check_launch("attn_decode_fused");
}
