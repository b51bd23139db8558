// Weight-streaming "skinny" GEMM for the decode step:  C[M<=64, N] = X[M,K] . W[N,K]^T
// (SURVEY.md K11/K16/K17/K18 at M = batch).  HBM-bound on W: every weight byte is read
// exactly once per step, straight into registers (no LDS round trip), 64 contiguous bytes per
// lane and 256 contiguous bytes per weight row per 128-wide k block -- the MFMA's k index is
// re-labelled k = 32g + 8i + j (g = lane>>4, i = MFMA step) on both operands so that a lane's
// four 16-byte loads are consecutive.  X (the small operand) is staged through LDS in 256-wide
// k chunks with the T10(b) XOR swizzle (conflict-free ds_read_b128).
//
// Grid (N/64, S): a workgroup of 4 waves owns 64 weight rows and one K slice of KS columns;
// S > 1 (split-K) fills the 256 CUs when N is small and writes fp32 partial slabs
// [S][M][N] that the consumer kernel sums in fixed order (deterministic, batch-invariant:
// no float atomics).  Algorithmic bytes per launch = 2*N*K (+ 2*M*K re-read from L2).
#include "common.h"

#define SK_KC 256

__device__ __forceinline__ int x_off(int m, int ch) {   // [rows][256 bf16] image, 512 B rows, 32 chunks per row
    const int f = ((m & 3) << 2) | ((m >> 2) & 3);
    return m * 512 + 16 * ((ch & 16) | ((ch & 15) ^ f));
}

template <int MT>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const bf16* __restrict__ X, long ldx, int M,
                                                          const bf16* __restrict__ W, long ldw, int N, int KS,
                                                          float* __restrict__ slab, bf16* __restrict__ out, long ldo) {
    __shared__ __attribute__((aligned(16))) char sx[2][MT * 16 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int n0 = blockIdx.x * 64 + wave * 16;
    const int split = blockIdx.y;
    const int kbeg = split * KS;
    const int nc = KS / SK_KC;
    const int wrow = min(n0 + li, N - 1);
    const bf16* wp = W + (long)wrow * ldw + kbeg + 32 * g;

    // X staging: chunk slots (row = tid>>5 + 8q, ch = tid&31), q < 2*MT
    const int xr0 = tid >> 5, xch = tid & 31;
    uint4 xr[2 * MT];
    auto x_load = [&](int c) {
#pragma unroll
        for (int q = 0; q < 2 * MT; ++q) {
            const int m = xr0 + 8 * q;
            xr[q] = (m < M) ? *reinterpret_cast<const uint4*>(X + (long)m * ldx + kbeg + c * SK_KC + xch * 8)
                            : make_uint4(0, 0, 0, 0);
        }
    };
    auto x_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2 * MT; ++q) *reinterpret_cast<uint4*>(sx[buf] + x_off(xr0 + 8 * q, xch)) = xr[q];
    };
    auto w_load = [&](bf16x8 (&w)[8], int c) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[blk * 4 + i] = *reinterpret_cast<const bf16x8*>(wp + c * SK_KC + blk * 128 + 8 * i);
    };
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const bf16x8 (&w)[8], int buf) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sx[buf] + x_off(16 * mt + li, 16 * blk + 4 * g + i));
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[blk * 4 + i], xf, acc[mt], 0, 0, 0);
                }
    };

    bf16x8 wa[8], wb[8];
    w_load(wa, 0);
    x_load(0);
    x_store(0);
    int c = 0;
    while (true) {
        if (c + 1 < nc) { w_load(wb, c + 1); x_load(c + 1); }
        __syncthreads();
        compute(wa, 0);
        if (c + 1 < nc) x_store(1);
        if (++c >= nc) break;
        if (c + 1 < nc) { w_load(wa, c + 1); x_load(c + 1); }
        __syncthreads();
        compute(wb, 1);
        if (c + 1 < nc) x_store(0);
        if (++c >= nc) break;
    }

    // lane holds C[n = n0 + 4g + r][m = 16 mt + li]
    const int n = n0 + 4 * g;
    if (n >= N) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + li;
        if (m >= M) continue;
        if (slab) {
            *reinterpret_cast<f32x4*>(slab + ((long)split * M + m) * N + n) = acc[mt];
        } else {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[mt][r]);
            *reinterpret_cast<bf16x4*>(out + (long)m * ldo + n) = o;
        }
    }
}

extern "C" int q3asr_skinny_gemm(const void* X, int64_t ldx, int M, const void* W, int64_t ldw, int N, int K, int n_splits,
                                 float* slab, void* out, int64_t ldo, void* stream) {
    Q3_REQUIRE(X && W && (slab || out), "skinny_gemm: null pointer");
    Q3_REQUIRE(M > 0 && M <= 64, "skinny_gemm: M=%d must be in 1..64", M);
    Q3_REQUIRE(N > 0 && N % 4 == 0, "skinny_gemm: N=%d must be a positive multiple of 4", N);
    Q3_REQUIRE(n_splits > 0 && K % (n_splits * SK_KC) == 0, "skinny_gemm: K=%d must be a multiple of n_splits*%d (n_splits=%d)", K, SK_KC, n_splits);
    Q3_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, "skinny_gemm: bad leading dimensions");
    Q3_REQUIRE(slab || n_splits == 1, "skinny_gemm: split-K needs a slab workspace");
    Q3_REQUIRE(slab || (ldo % 4 == 0 && ldo >= N), "skinny_gemm: bad ldo");
    const int KS = K / n_splits;
    dim3 grid((N + 63) / 64, n_splits), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16* x = (const bf16*)X; const bf16* w = (const bf16*)W; bf16* o = (bf16*)out;
    if (M <= 16) hipLaunchKernelGGL(skinny_gemm_kernel<1>, grid, blk, 0, s, x, (long)ldx, M, w, (long)ldw, N, KS, slab, o, (long)ldo);
    else if (M <= 32) hipLaunchKernelGGL(skinny_gemm_kernel<2>, grid, blk, 0, s, x, (long)ldx, M, w, (long)ldw, N, KS, slab, o, (long)ldo);
    else hipLaunchKernelGGL(skinny_gemm_kernel<4>, grid, blk, 0, s, x, (long)ldx, M, w, (long)ldw, N, KS, slab, o, (long)ldo);
    return q3_check_launch("skinny_gemm");
}

// ---------------------------------------------------------------------------
// Consumers of the split-K slabs (fixed summation order s = 0..S-1).
// ---------------------------------------------------------------------------

// h_out = bf16(h_in + bf16(sum slabs)) ; x_out = RMSNorm(h_out) * w   (decoder.py:191,194 / :196,189 / :251)
template <int EPT>   // elements per thread: N = 256 * EPT (EPT = 2: 8-byte accesses, else 16-byte)
__global__ __launch_bounds__(256) void slab_residual_rmsnorm_kernel(const float* __restrict__ slab, int S, long slab_stride,
                                                                    const bf16* __restrict__ h_in, bf16* __restrict__ h_out,
                                                                    const bf16* __restrict__ w, bf16* __restrict__ x_out, float eps) {
    __shared__ float red[4];
    constexpr int N = 256 * EPT;
    constexpr int VW = EPT >= 4 ? 4 : 2;                   // floats per vector load
    typedef float fvec __attribute__((ext_vector_type(VW)));
    typedef bf16 bvec __attribute__((ext_vector_type(VW)));
    const long m = blockIdx.x;
    const int c0 = threadIdx.x * EPT;
    fvec v[EPT / VW];
#pragma unroll
    for (int q = 0; q < EPT / VW; ++q) v[q] = *reinterpret_cast<const fvec*>(slab + m * N + c0 + q * VW);
    for (int s = 1; s < S; ++s) {
        const float* p = slab + s * slab_stride + m * N + c0;
#pragma unroll
        for (int q = 0; q < EPT / VW; ++q) v[q] += *reinterpret_cast<const fvec*>(p + q * VW);
    }
    float q2 = 0.f;
#pragma unroll
    for (int q = 0; q < EPT / VW; ++q) {
        const bvec hv = *reinterpret_cast<const bvec*>(h_in + m * N + c0 + q * VW);
        bvec ho;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const float t = round_bf(round_bf(v[q][e]) + bf2f(hv[e]));
            v[q][e] = t;
            q2 = fmaf(t, t, q2);
            ho[e] = f2bf(t);
        }
        *reinterpret_cast<bvec*>(h_out + m * N + c0 + q * VW) = ho;
    }
    q2 = wave_sum(q2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q2;
    __syncthreads();
    const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) * (1.0f / N) + eps);
#pragma unroll
    for (int q = 0; q < EPT / VW; ++q) {
        const bvec wv = *reinterpret_cast<const bvec*>(w + c0 + q * VW);
        bvec xo;
#pragma unroll
        for (int e = 0; e < VW; ++e) xo[e] = f2bf(v[q][e] * r * bf2f(wv[e]));
        *reinterpret_cast<bvec*>(x_out + m * N + c0 + q * VW) = xo;
    }
}

extern "C" int q3asr_slab_residual_rmsnorm(const float* slab, int n_slabs, int64_t slab_stride, const void* h_in, void* h_out,
                                           const void* w, void* x_out, int M, int N, float eps, void* stream) {
    Q3_REQUIRE(slab && h_in && h_out && w && x_out, "slab_residual_rmsnorm: null pointer");
    Q3_REQUIRE(M > 0 && n_slabs > 0 && (N == 512 || N == 1024 || N == 2048 || N == 4096), "slab_residual_rmsnorm: N=%d unsupported", N);
    dim3 grid(M), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16* hi = (const bf16*)h_in; bf16* ho = (bf16*)h_out; const bf16* ww = (const bf16*)w; bf16* xo = (bf16*)x_out;
    switch (N / 256) {
        case 2: hipLaunchKernelGGL(slab_residual_rmsnorm_kernel<2>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, eps); break;
        case 4: hipLaunchKernelGGL(slab_residual_rmsnorm_kernel<4>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, eps); break;
        case 8: hipLaunchKernelGGL(slab_residual_rmsnorm_kernel<8>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, eps); break;
        default: hipLaunchKernelGGL(slab_residual_rmsnorm_kernel<16>, grid, blk, 0, s, slab, n_slabs, (long)slab_stride, hi, ho, ww, xo, eps); break;
    }
    return q3_check_launch("slab_residual_rmsnorm");
}

// act[m][o] = bf16( silu(bf16(gate)) * bf16(up) ), slab columns 16-row interleaved: gate of feature
// o at column 32*(o/16) + o%16, up 16 columns later (decoder.py:97).
__global__ __launch_bounds__(256) void slab_swiglu_kernel(const float* __restrict__ slab, int S, long slab_stride,
                                                          bf16* __restrict__ act, int M, int I) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // one thread per 4 outputs
    const int per_row = I / 4;
    if (idx >= (long)M * per_row) return;
    const int m = (int)(idx / per_row), o = (int)(idx - (long)m * per_row) * 4;
    const int gc = 32 * (o >> 4) + (o & 15);
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f}, usum = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const float* p = slab + s * slab_stride + (long)m * 2 * I + gc;
        gsum += *reinterpret_cast<const f32x4*>(p);
        usum += *reinterpret_cast<const f32x4*>(p + 16);
    }
    bf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gt = round_bf(gsum[e]), up = round_bf(usum[e]);
        r[e] = f2bf(gt / (1.0f + __expf(-gt)) * up);
    }
    *reinterpret_cast<bf16x4*>(act + (long)m * I + o) = r;
}

extern "C" int q3asr_slab_swiglu(const float* slab, int n_slabs, int64_t slab_stride, void* act, int M, int I, void* stream) {
    Q3_REQUIRE(slab && act, "slab_swiglu: null pointer");
    Q3_REQUIRE(M > 0 && n_slabs > 0 && I % 16 == 0, "slab_swiglu: I=%d must be a multiple of 16", I);
    const long n = (long)M * (I / 4);
    hipLaunchKernelGGL(slab_swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slab, n_slabs,
                       (long)slab_stride, (bf16*)act, M, I);
    return q3_check_launch("slab_swiglu");
}

// h = E[tok[b]] ; x = RMSNorm(h) * w   (generate.py:294 + decoder.py:189 of layer 0)
template <int EPT>
__global__ __launch_bounds__(256) void embed_rmsnorm_kernel(const int* __restrict__ tok, const bf16* __restrict__ E,
                                                            bf16* __restrict__ h_out, const bf16* __restrict__ w,
                                                            bf16* __restrict__ x_out, float eps, int V) {
    __shared__ float red[4];
    constexpr int N = 256 * EPT;
    const long m = blockIdx.x;
    const int c0 = threadIdx.x * EPT;
    const bool ok = (unsigned)tok[m] < (unsigned)V;                    // an id outside [0, V) gathers a zero row
    const bf16* src = E + (long)(ok ? tok[m] : 0) * N + c0;
    float v[EPT];
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const bf16 hv = ok ? src[e] : f2bf(0.f);
        h_out[m * N + c0 + e] = hv;
        v[e] = bf2f(hv);
        q = fmaf(v[e], v[e], q);
    }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) * (1.0f / N) + eps);
#pragma unroll
    for (int e = 0; e < EPT; ++e) x_out[m * N + c0 + e] = f2bf(v[e] * r * bf2f(w[c0 + e]));
}

extern "C" int q3asr_embed_rmsnorm(const int32_t* tok, const void* E, int V, void* h_out, const void* w, void* x_out, int M, int N,
                                   float eps, void* stream) {
    Q3_REQUIRE(tok && E && h_out && w && x_out && V > 0, "embed_rmsnorm: null pointer or V <= 0");
    Q3_REQUIRE(M > 0 && (N == 512 || N == 1024 || N == 2048 || N == 4096), "embed_rmsnorm: N=%d unsupported", N);
    dim3 grid(M), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const bf16* e = (const bf16*)E; bf16* ho = (bf16*)h_out; const bf16* ww = (const bf16*)w; bf16* xo = (bf16*)x_out;
    switch (N / 256) {
        case 2: hipLaunchKernelGGL(embed_rmsnorm_kernel<2>, grid, blk, 0, s, tok, e, ho, ww, xo, eps, V); break;
        case 4: hipLaunchKernelGGL(embed_rmsnorm_kernel<4>, grid, blk, 0, s, tok, e, ho, ww, xo, eps, V); break;
        case 8: hipLaunchKernelGGL(embed_rmsnorm_kernel<8>, grid, blk, 0, s, tok, e, ho, ww, xo, eps, V); break;
        default: hipLaunchKernelGGL(embed_rmsnorm_kernel<16>, grid, blk, 0, s, tok, e, ho, ww, xo, eps, V); break;
    }
    return q3_check_launch("embed_rmsnorm");
}
