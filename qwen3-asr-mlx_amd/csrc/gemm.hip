// bf16 MFMA GEMM with fused epilogues:  C[M,N] = A[M,K] . W[N,K]^T
//
// Serves every MLX nn.Linear / nn.Conv2d call site on the large-M side of the
// path (SURVEY.md K3,K4,K7,K11,K16,K17): encoder q/k/v/out/fc1/fc2/proj, the
// conv stem as implicit GEMM (im2col addresses generated in the A loader, never
// materialised), decoder prefill projections.  Both operands are K-contiguous
// (activations row-major, weights in the checkpoint's (out,in) layout), so both
// are staged the same way: 16-byte global loads -> registers -> XOR-swizzled LDS
// tile [128 rows][64 bf16] -> ds_read_b128 fragments -> v_mfma_f32_16x16x32_bf16.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA
// tiles.  The MFMA is issued "swapped" (A operand = weight fragment, B operand =
// activation fragment), so a lane ends up with 4 consecutive output columns of
// one row: bias/residual loads and the store are 8/16-byte vectors, and the
// SwiGLU pair (gate, up) of a feature sits in the same lane and register.
// Bound: MFMA (dense bf16 peak 2.5 PFLOP/s); algorithmic flops = 2*M*N*K.
#include "common.h"
#include <atomic>
#include <type_traits>
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#define GROUP_M 8

struct GemmParams {
    const bf16* A; long lda;
    const bf16* W; long ldw;
    int M, N, K;
    // implicit-GEMM conv (k3 s2 p1, NHWC): A = input [imgs][Hin][Win][Cin], M = imgs*Hout*Wout, K = 9*Cin
    int Hin, Win, Cin, Hout, Wout;
    const float* bias;
    int act;                 // 0 none, 1 exact-erf GELU, 2 SwiGLU over 16-row interleaved (gate, up) weight rows
    const void* res; int res_kind; long ldres; int res_mod;   // 0 none, 1 fp32, 2 bf16 (MLX bf16 add: round, add, round)
    void* out; int out_kind; long ldo;                        // 0 bf16, 1 fp32
    const int* row_map;      // optional destination row per m (<0: skip)
    int lds_epi;             // 8-phase kernel: epilogue through the wave-private LDS transpose (host-checked preconditions)
    // fp8 (OCP e4m3) operands on v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (q3asr_gemm_fp8): A and W are bytes, C = (A8 . W8^T) * a_scale[m] * w_scale[n]
    const float* a_scale; const float* w_scale;
    unsigned long long* stamp;   // diagnostic (q3asr_gemm_set_stamp): workgroup stamp_wg records wall_clock64() at its phase boundaries
    int stamp_wg;
};

// 128-byte rows: two rows make one 256-byte bank row, so the XOR uses row >> 1 and the row parity supplies the 4th bit:
// the 16 lanes of a ds_read_b128 group (rows base..base+15, one k slot) land on 16 distinct 16-byte columns.
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// Epilogue shared by both GEMM kernels.  A lane holds, for accumulator tile (j, i): output row
// m = m_base + 16 i + li and the 4 consecutive columns n = n_base + 16 j + 4 g + r.
template <int MI, int NJ = 4, int MIA = MI>   // NJ: 16-column tiles per wave (4: 64 columns; 2: the 64x64-tile kernel's 32); MIA: row blocks the array holds (>= MI)
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[NJ][MIA], int m_base, int n_base, int n0, int wn,
                                              int g, int li) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + li;
        if (m >= p.M) continue;
        long orow = m;
        if (p.row_map) {
            orow = p.row_map[m];
            if (orow < 0) continue;
        }
        const long rrow = p.res_mod > 0 ? (m % p.res_mod) : m;
        if (p.act == 2) {
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
                const int n_in = n_base + j * 16 + 4 * g;            // gate rows; up rows are n_in + 16
                if (n_in >= p.N) continue;
                const int n_out = (n0 >> 1) + wn * (8 * NJ) + (j >> 1) * 16 + 4 * g;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gt = round_bf(acc[j][i][r]);            // gate_proj output is bf16 (decoder.py:97)
                    const float up = round_bf(acc[j + 1][i][r]);
                    o[r] = swiglu_bf(gt, up);
                }
                *reinterpret_cast<bf16x4*>((bf16*)p.out + orow * p.ldo + n_out) = o;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n_base + j * 16 + 4 * g;
            if (n >= p.N) continue;
            f32x4 v = acc[j][i];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if (p.res_kind == 1) {
                v += *reinterpret_cast<const f32x4*>((const float*)p.res + rrow * p.ldres + n);
            } else if (p.res_kind == 2) {
                const bf16x4 rv = *reinterpret_cast<const bf16x4*>((const bf16*)p.res + rrow * p.ldres + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = round_bf(v[r]) + bf2f(rv[r]);
            }
            if (p.out_kind == 1) {
                *reinterpret_cast<f32x4*>((float*)p.out + orow * p.ldo + n) = v;
            } else {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
                *reinterpret_cast<bf16x4*>((bf16*)p.out + orow * p.ldo + n) = o;
            }
        }
    }
}


// ---------------------------------------------------------------------------
// Epilogue through a wave-private LDS transpose (8-phase kernel).  In the accumulator layout a lane owns 4 consecutive
// columns of ONE row, so a direct store instruction covers 16 rows x 32 bytes: 32 such narrow stores per lane and tile made
// the epilogue the largest fixed cost of a short-K tile (K sweep, profiles/README.md: ~15 us per 256^2 tile, store-issue
// bound).  Here each 16-row x 64-column block of the wave's 128 x 64 output goes through 4 KiB of LDS that only this wave
// touches (no barrier: DS operations of one wave execute in order) and leaves as whole rows: 8 lanes x 16 B per 128-byte
// row (bf16), 16 lanes x 16 B per 256-byte row (fp32).  Bias / GELU / SwiGLU are applied before the transpose, residuals
// (read as whole rows too) after it; arithmetic and rounding points are those of gemm_epilogue, so results are bit-identical.
// Preconditions (checked on the host, else gemm_epilogue): N % 8 == 0, ldo / ldres multiples of 8, 16-byte aligned bases.
// ---------------------------------------------------------------------------
// PRE: the residual-ahead form exists (plain bf16 GEMM only: the fp8 and conv variants have no 16-32 registers to spare in their epilogues).
template <int MI = 8, int MIA = MI, bool PRE = false>   // MI: 16-row blocks per wave: 8 (128 rows), 7 (112) or 6 (96: the second wave row of a 208-row tile); MIA: blocks the array holds
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, f32x4 (&acc)[4][MIA], int m_base, int n_base, int n0, int wn,
                                                  char* st, int lane) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // lane-derived addresses are recomputed here from an opaque copy of the lane id: hoisted above the K loop they would
    // stay live across it, and the 8-phase loop has no registers to spare (a spill there is a VMEM op in the counted queue)
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, li = lane & 15;
    if (p.act == 2) {                                                     // SwiGLU: 16 rows x 32 output columns per block (64-byte rows)
        const int n_out0 = (n0 >> 1) + wn * 32;
        const int rr = lane >> 2, rc = lane & 3;
        if (p.lds_epi >= 2 && m_base + 16 * MI <= p.M && n_base + 64 <= p.N && !p.row_map) {   // interior sub-tile: see the bf16 path below
            const unsigned voff = (unsigned)rr * (unsigned)p.ldo + (unsigned)(n_out0 + 8 * rc);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                char* blk = st + (i & 1) * 2048;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gt = round_bf(acc[2 * jj][i][r]);
                        const float up = round_bf(acc[2 * jj + 1][i][r]);
                        o[r] = swiglu_bf(gt, up);
                    }
                    const int chunk = 2 * jj + (g >> 1);
                    *reinterpret_cast<bf16x4*>(blk + li * 64 + ((chunk ^ ((li >> 1) & 3)) << 4) + (g & 1) * 8) = o;
                }
                const u32x4 v = *reinterpret_cast<const u32x4*>(blk + rr * 64 + ((rc ^ ((rr >> 1) & 3)) << 4));
                *reinterpret_cast<u32x4*>((bf16*)p.out + (long)(m_base + i * 16) * p.ldo + voff) = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            char* blk = st + (i & 1) * 2048;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gt = round_bf(acc[2 * jj][i][r]);
                    const float up = round_bf(acc[2 * jj + 1][i][r]);
                    o[r] = swiglu_bf(gt, up);
                }
                const int chunk = 2 * jj + (g >> 1);
                *reinterpret_cast<bf16x4*>(blk + li * 64 + ((chunk ^ ((li >> 1) & 3)) << 4) + (g & 1) * 8) = o;
            }
            const u32x4 v = *reinterpret_cast<const u32x4*>(blk + rr * 64 + ((rc ^ ((rr >> 1) & 3)) << 4));
            const int m = m_base + i * 16 + rr, n_in = n_base + 32 * (rc >> 1);   // 16 output columns <- one 32-row (gate, up) block
            if (m < p.M && n_in < p.N) {
                long orow = m;
                if (p.row_map) orow = p.row_map[m];
                if (orow >= 0) *reinterpret_cast<u32x4*>((bf16*)p.out + orow * p.ldo + n_out0 + 8 * rc) = v;
            }
        }
        return;
    }
    if (p.out_kind == 0) {                                                // bf16 out: 16 rows x 64 columns per block (128-byte rows)
        const int rc = lane & 7;
        const int n = n_base + 8 * rc;
        // Round 4: a wave's sub-tile that lies wholly inside the matrix (every tile but those of the last row / column of tiles)
        // and has no row map stores through a uniform row pointer + one 32-bit lane offset, without per-row bound tests and
        // 64-bit address chains: a wave64 VALU instruction takes ~4 cycles on this part (tools/pmc_kernel.sh), and the general
        // path spends a few hundred of them per tile on addresses.  Same values, same rounding points.
        if (p.lds_epi >= 2 && m_base + 16 * MI <= p.M && n_base + 64 <= p.N && !p.row_map && p.res_mod <= 0) {
            const unsigned voff = (unsigned)(lane >> 3) * (unsigned)p.ldo + (unsigned)(n_base + 8 * rc);
            const unsigned roff = (unsigned)(lane >> 3) * (unsigned)p.ldres + (unsigned)(n_base + 8 * rc);
            f32x4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n_base + j * 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
            // Round 5: the residual rows of block i + 1 are requested before block i is transposed and stored (lds_epi == 3).  Read
            // where they are used, every 16-row block paid a memory round trip of its own -- 6 to 8 per sub-tile, the "burst" the
            // round-4 stamps showed as a 17-19 us epilogue on the tiles with a residual.  Same values, same adds.
            // (Two fp32 / four bf16 blocks ahead -- the registers exist in the 224- and 208-row variants -- measured the same as one: encoder out
            // 41.3-41.6 vs 41.3-41.5 us, fc2 97.4 vs 98.3, prefill o 101 vs 102: after the first block the stores and transposes set the pace.)
            const bool ahead = PRE && p.lds_epi == 3 && p.res_kind == 2;
            bf16x8 rn[2][2];
            auto load_res = [&](int i, bf16x8 (&r)[2]) {
#pragma unroll
                for (int h = 0; h < 2; ++h) r[h] = *reinterpret_cast<const bf16x8*>((const bf16*)p.res + (long)(m_base + i * 16 + 8 * h) * p.ldres + roff);
            };
            if (ahead) load_res(0, rn[0]);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                char* blk = st + (i & 1) * 2048;
                if (ahead && i + 1 < MI) load_res(i + 1, rn[(i + 1) & 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[j][i];
                    if (p.bias) v += bv[j];
                    if (p.act == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
                    const int chunk = 2 * j + (g >> 1);
                    *reinterpret_cast<bf16x4*>(blk + li * 128 + ((chunk ^ (li & 7)) << 4) + (g & 1) * 8) = o;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rr = (lane >> 3) + 8 * h;
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(blk + rr * 128 + ((rc ^ (rr & 7)) << 4));
                    const long mrow = m_base + i * 16 + 8 * h;                   // uniform
                    if (p.res_kind == 2) {                                   // MLX bf16 add: both operands bf16, sum rounded once
                        const bf16x8 rv = ahead ? rn[i & 1][h] : *reinterpret_cast<const bf16x8*>((const bf16*)p.res + mrow * p.ldres + roff);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(rv[e]));
                    }
                    *reinterpret_cast<bf16x8*>((bf16*)p.out + mrow * p.ldo + voff) = v;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            char* blk = st + (i & 1) * 2048;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[j][i];
                const int nb = n_base + j * 16 + 4 * g;
                if (p.bias && nb < p.N) v += *reinterpret_cast<const f32x4*>(p.bias + nb);
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
                const int chunk = 2 * j + (g >> 1);
                *reinterpret_cast<bf16x4*>(blk + li * 128 + ((chunk ^ (li & 7)) << 4) + (g & 1) * 8) = o;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = (lane >> 3) + 8 * h;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(blk + rr * 128 + ((rc ^ (rr & 7)) << 4));
                const int m = m_base + i * 16 + rr;
                if (m >= p.M || n >= p.N) continue;
                long orow = m;
                if (p.row_map) {
                    orow = p.row_map[m];
                    if (orow < 0) continue;
                }
                if (p.res_kind == 2) {                                   // MLX bf16 add: both operands bf16, sum rounded once
                    const long rrow = p.res_mod > 0 ? (m % p.res_mod) : m;
                    const bf16x8 rv = *reinterpret_cast<const bf16x8*>((const bf16*)p.res + rrow * p.ldres + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(rv[e]));
                }
                *reinterpret_cast<bf16x8*>((bf16*)p.out + orow * p.ldo + n) = v;
            }
        }
        return;
    }
    // fp32 out (+ optional fp32 residual): 16 rows x 64 columns per block (256-byte rows), one block in flight
    const int rc = lane & 15;
    const int n = n_base + 4 * rc;
    if (p.lds_epi >= 2 && m_base + 16 * MI <= p.M && n_base + 64 <= p.N && !p.row_map && p.res_mod <= 0) {   // interior sub-tile: see the bf16 path above
        const unsigned voff = (unsigned)(lane >> 4) * (unsigned)p.ldo + (unsigned)(n_base + 4 * rc);
        const unsigned roff = (unsigned)(lane >> 4) * (unsigned)p.ldres + (unsigned)(n_base + 4 * rc);
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n_base + j * 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        const bool ahead = PRE && p.lds_epi == 3 && p.res_kind == 1;    // residual rows one block ahead: see the bf16 path above
        f32x4 rn[2][4];
        auto load_res = [&](int i, f32x4 (&r)[4]) {
#pragma unroll
            for (int h = 0; h < 4; ++h) r[h] = *reinterpret_cast<const f32x4*>((const float*)p.res + (long)(m_base + i * 16 + 4 * h) * p.ldres + roff);
        };
        if (ahead) load_res(0, rn[0]);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (ahead && i + 1 < MI) load_res(i + 1, rn[(i + 1) & 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[j][i];
                if (p.bias) v += bv[j];
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
                *reinterpret_cast<f32x4*>(st + li * 256 + (((4 * j + g) ^ li) << 4)) = v;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int rr = (lane >> 4) + 4 * h;
                f32x4 v = *reinterpret_cast<const f32x4*>(st + rr * 256 + ((rc ^ rr) << 4));
                const long mrow = m_base + i * 16 + 4 * h;                       // uniform
                if (p.res_kind == 1) v += ahead ? rn[i & 1][h] : *reinterpret_cast<const f32x4*>((const float*)p.res + mrow * p.ldres + roff);
                *reinterpret_cast<f32x4*>((float*)p.out + mrow * p.ldo + voff) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[j][i];
            const int nb = n_base + j * 16 + 4 * g;
            if (p.bias && nb < p.N) v += *reinterpret_cast<const f32x4*>(p.bias + nb);
            if (p.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            *reinterpret_cast<f32x4*>(st + li * 256 + (((4 * j + g) ^ li) << 4)) = v;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int rr = (lane >> 4) + 4 * h;
            f32x4 v = *reinterpret_cast<const f32x4*>(st + rr * 256 + ((rc ^ rr) << 4));
            const int m = m_base + i * 16 + rr;
            if (m >= p.M || n >= p.N) continue;
            long orow = m;
            if (p.row_map) {
                orow = p.row_map[m];
                if (orow < 0) continue;
            }
            if (p.res_kind == 1) {
                const long rrow = p.res_mod > 0 ? (m % p.res_mod) : m;
                v += *reinterpret_cast<const f32x4*>((const float*)p.res + rrow * p.ldres + n);
            }
            *reinterpret_cast<f32x4*>((float*)p.out + orow * p.ldo + n) = v;
        }
    }
}

template <int CONV, bool FAST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * BM * BK * 2];   // A0 | W0 | A1 | W1, 16 KB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int group = GROUP_M * tiles_n;
    const int first_m = (t / group) * GROUP_M;
    const int gm = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (int)(t % group) % gm;
    const int tn = (int)(t % group) / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- loader state: this thread moves 4 A slots and 4 W slots (16 B each) per K tile
    const int slot = tid & 7, lrow = tid >> 3;
    long a_base[4];
    int a_ih0[4], a_iw0[4];
    bool a_ok[4], w_ok[4];
    long w_base[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int am = m0 + lrow + 32 * q;
        a_ok[q] = am < p.M;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int img = am / hw, rem = am - img * hw;
            const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
            a_ih0[q] = 2 * oh - 1;
            a_iw0[q] = 2 * ow - 1;
            a_base[q] = (long)img * p.Hin * p.Win * p.Cin;
        } else {
            a_base[q] = (long)am * p.lda;
            a_ih0[q] = a_iw0[q] = 0;
        }
        const int wn_ = n0 + lrow + 32 * q;
        w_ok[q] = wn_ < p.N;
        w_base[q] = (long)wn_ * p.ldw;
    }

    // FAST (plain GEMM, K % 64 == 0, operands < 4 GiB): edge rows are clamped instead of predicated (their outputs are
    // masked in the epilogue) and every load is uniform base + 32-bit lane offset: no branches, no 64-bit VALU per step.
    unsigned fa_off[4] = {0, 0, 0, 0}, fw_off[4] = {0, 0, 0, 0};
    if constexpr (FAST) {
        fa_off[0] = (unsigned)(((long)min(m0 + lrow, p.M - 1) * p.lda + slot * 8) * 2);
        fa_off[1] = (unsigned)(((long)min(m0 + lrow + 32, p.M - 1) * p.lda + slot * 8) * 2);
        fa_off[2] = (unsigned)(((long)min(m0 + lrow + 64, p.M - 1) * p.lda + slot * 8) * 2);
        fa_off[3] = (unsigned)(((long)min(m0 + lrow + 96, p.M - 1) * p.lda + slot * 8) * 2);
        fw_off[0] = (unsigned)(((long)min(n0 + lrow, p.N - 1) * p.ldw + slot * 8) * 2);
        fw_off[1] = (unsigned)(((long)min(n0 + lrow + 32, p.N - 1) * p.ldw + slot * 8) * 2);
        fw_off[2] = (unsigned)(((long)min(n0 + lrow + 64, p.N - 1) * p.ldw + slot * 8) * 2);
        fw_off[3] = (unsigned)(((long)min(n0 + lrow + 96, p.N - 1) * p.ldw + slot * 8) * 2);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[4], rw[4];
    auto load_tile = [&](int k0) {
        const int k = k0 + slot * 8;
        const bool kin = k < p.K;
        int tap_off = 0, kh = 0, kw = 0;
        bool cin_ok = true;
        if (CONV) {                                          // chunk-major K: k-tile kt = (64-channel chunk, tap), see conv3x3s2
            const int kt = k0 >> 6, chunk = kt / 9, tap = kt - 9 * chunk;
            kh = tap / 3;
            kw = tap - 3 * kh;
            tap_off = chunk * 64 + slot * 8;
            cin_ok = tap_off < p.Cin;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (CONV) {
                const int ih = a_ih0[q] + kh, iw = a_iw0[q] + kw;
                if (a_ok[q] && kin && cin_ok && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win)
                    v = *reinterpret_cast<const u32x4*>(p.A + a_base[q] + ((long)(ih * p.Win + iw) * p.Cin + tap_off));
            } else {
                if (a_ok[q] && kin) v = *reinterpret_cast<const u32x4*>(p.A + a_base[q] + k);
            }
            ra[q] = v;
            u32x4 w = {0u, 0u, 0u, 0u};
            if (w_ok[q] && kin) w = *reinterpret_cast<const u32x4*>(p.W + w_base[q] + k);
            rw[q] = w;
        }
    };
#define Q3_LOAD_FAST(K0)                                                                      \
    {                                                                                         \
        const char* pa_ = reinterpret_cast<const char*>(p.A) + (long)(K0) * 2;                \
        const char* pw_ = reinterpret_cast<const char*>(p.W) + (long)(K0) * 2;                \
        ra[0] = *reinterpret_cast<const u32x4*>(pa_ + fa_off[0]);                             \
        ra[1] = *reinterpret_cast<const u32x4*>(pa_ + fa_off[1]);                             \
        ra[2] = *reinterpret_cast<const u32x4*>(pa_ + fa_off[2]);                             \
        ra[3] = *reinterpret_cast<const u32x4*>(pa_ + fa_off[3]);                             \
        rw[0] = *reinterpret_cast<const u32x4*>(pw_ + fw_off[0]);                             \
        rw[1] = *reinterpret_cast<const u32x4*>(pw_ + fw_off[1]);                             \
        rw[2] = *reinterpret_cast<const u32x4*>(pw_ + fw_off[2]);                             \
        rw[3] = *reinterpret_cast<const u32x4*>(pw_ + fw_off[3]);                             \
    }
    auto store_tile = [&](int buf) {
        char* sa = smem + buf * (2 * BM * BK * 2);
        char* sw = sa + BM * BK * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lrow + 32 * q;
            *reinterpret_cast<u32x4*>(sa + lds_off(r, slot)) = ra[q];
            *reinterpret_cast<u32x4*>(sw + lds_off(r, slot)) = rw[q];
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    f32x4 acc[4][4];   // [j: n tile][i: m tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    auto compute_tile = [&](int buf) {
        const char* sa = smem + buf * (2 * BM * BK * 2);
        const char* sw = sa + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = *reinterpret_cast<const bf16x8*>(sw + lds_off(wn * 64 + j * 16 + li, kk * 4 + g));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(wm * 64 + i * 16 + li, kk * 4 + g));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
    };
    if constexpr (FAST) {
        // Two K tiles in flight in registers (round 3): on a grid of at most one round a workgroup has no neighbour whose MFMAs
        // hide its loads, and with one tile in flight the loop ran at a load latency per K tile (prefill down at 4 clips, 208
        // tiles, K = 6,144: 77 us).  Loads are unconditional (K tile clamped to the last one) so the counted vmcnt waits stay exact.
        u32x4 rb[4], rv[4];                                                   // the second register set
#define Q3_LOAD_FAST2(K0)                                                                     \
    {                                                                                         \
        const char* pa_ = reinterpret_cast<const char*>(p.A) + (long)(K0) * 2;                \
        const char* pw_ = reinterpret_cast<const char*>(p.W) + (long)(K0) * 2;                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                       \
            rb[q] = *reinterpret_cast<const u32x4*>(pa_ + fa_off[q]);                         \
            rv[q] = *reinterpret_cast<const u32x4*>(pw_ + fw_off[q]);                         \
        }                                                                                     \
    }
        auto store_tile2 = [&](int buf) {
            char* sa = smem + buf * (2 * BM * BK * 2);
            char* sw = sa + BM * BK * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = lrow + 32 * q;
                *reinterpret_cast<u32x4*>(sa + lds_off(r, slot)) = rb[q];
                *reinterpret_cast<u32x4*>(sw + lds_off(r, slot)) = rv[q];
            }
        };
        Q3_LOAD_FAST(0)
        Q3_LOAD_FAST2(min(1, nk - 1) * BK)
        store_tile(0);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 <= nk; kt += 2) {                                      // set (ra, rw) holds even K tiles, (rb, rv) odd ones
            Q3_LOAD_FAST(min(kt + 2, nk - 1) * BK)
            __builtin_amdgcn_sched_barrier(0);                               // the scheduler otherwise sinks the loads below the
            compute_tile(0);                                                  // LDS stores of the other set (one tile in flight again)
            store_tile2(1);
            __syncthreads();
            Q3_LOAD_FAST2(min(kt + 3, nk - 1) * BK)
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(1);
            store_tile(0);
            __syncthreads();
        }
        if (kt < nk) compute_tile(0);
#undef Q3_LOAD_FAST2
    } else {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tile((kt + 1) * BK);
            compute_tile(buf);
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    gemm_epilogue<4>(p, acc, m0 + wm * 64, n0 + wn * 64, n0, wn, g, li);
}

// ---------------------------------------------------------------------------
// 64x64x64 tiles for GEMMs whose 128^2 grid cannot cover the chip (round 3): a single 30 s clip is M = 390 / 408 rows, and
// every output keeps ONE k-ascending MFMA chain (no split-K: the same utterance must give the same bits alone and inside
// a batch of 32), so the only way to put a one-clip `down` projection (N = 2,048, K = 6,144: 64 workgroups of 128^2) on more
// than a quarter of the CUs is a smaller tile: 224 workgroups of 64^2.  Register-staged, XOR-swizzled, double-buffered LDS like
// gemm_bf16_kernel (4 waves, each 32 x 32 = 2 x 2 MFMA tiles), same accumulation order, same bits.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm64_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * BK * 2];   // A0 | W0 | A1 | W1, 8 KB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + 63) / 64, tiles_m = (p.M + 63) / 64;
    const unsigned t = xcd_remap(blockIdx.x, gridDim.x);
    const int group = GROUP_M * tiles_n;
    const int first_m = (t / group) * GROUP_M;
    const int gm = min(tiles_m - first_m, GROUP_M);
    const int m0 = (first_m + (int)(t % group) % gm) * 64;
    const int n0 = ((int)(t % group) / gm) * 64;
    // loader: 2 A slots and 2 W slots (16 B each) per thread and K tile; edge rows clamped (their outputs are masked)
    const int slot = tid & 7, lrow = tid >> 3;
    unsigned fa_off[2], fw_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        fa_off[q] = (unsigned)(((long)min(m0 + lrow + 32 * q, p.M - 1) * p.lda + slot * 8) * 2);
        fw_off[q] = (unsigned)(((long)min(n0 + lrow + 32 * q, p.N - 1) * p.ldw + slot * 8) * 2);
    }
    // Register staging G64_D K tiles deep: a workgroup is alone on its CU (the grid is at most one round), so nothing but its
    // own loads in flight hides the ~1 us of a global load; with one tile in flight the loop ran at that latency per K tile
    // (one clip's down projection, 96 K tiles: 48 us), with four the loads of tiles kt+1..kt+4 overlap the MFMAs of tile kt.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int G64_D = 4;
    u32x4 ra[G64_D][2], rw[G64_D][2];
    const int nk = p.K / BK;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    f32x4 acc[2][2];   // [j: n tile][i: m tile]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#define G64_LOAD(set, kt_)                                                                      \
    do {                                                                                        \
        const char* pa_ = reinterpret_cast<const char*>(p.A) + (long)(kt_) * (BK * 2);          \
        const char* pw_ = reinterpret_cast<const char*>(p.W) + (long)(kt_) * (BK * 2);          \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                         \
            ra[set][q] = *reinterpret_cast<const u32x4*>(pa_ + fa_off[q]);                      \
            rw[set][q] = *reinterpret_cast<const u32x4*>(pw_ + fw_off[q]);                      \
        }                                                                                       \
    } while (0)
#define G64_STORE(set, buf_)                                                                    \
    do {                                                                                        \
        char* sa_ = smem + (buf_) * (2 * 64 * BK * 2);                                          \
        char* sw_ = sa_ + 64 * BK * 2;                                                          \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                         \
            const int r = lrow + 32 * q;                                                        \
            *reinterpret_cast<u32x4*>(sa_ + lds_off(r, slot)) = ra[set][q];                     \
            *reinterpret_cast<u32x4*>(sw_ + lds_off(r, slot)) = rw[set][q];                     \
        }                                                                                       \
    } while (0)
    // Every step issues its loads unconditionally (K tile index clamped to the last one: at most G64_D redundant tiles per
    // workgroup), so the compiler's counted vmcnt waits stay exact; behind a uniform `if` it falls back to vmcnt(0).
#define G64_STEP(d, kt)                                                                         \
    do {                                                                                        \
        G64_LOAD(d, min((kt) + G64_D, nk - 1));          /* set d went to LDS one step ago */   \
        const char* sa = smem + ((d) & 1) * (2 * 64 * BK * 2);                                  \
        const char* sw = sa + 64 * BK * 2;                                                      \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                      \
            bf16x8 wf[2], af[2];                                                                \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                wf[j] = *reinterpret_cast<const bf16x8*>(sw + lds_off(wn * 32 + j * 16 + li, kk * 4 + g)); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                       \
                af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(wm * 32 + i * 16 + li, kk * 4 + g)); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                   \
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0); \
        }                                                                                       \
        G64_STORE(((d) + 1) % G64_D, ((d) + 1) & 1);     /* K tile kt + 1 (a clamped copy after the last) */ \
        __syncthreads();                                                                        \
    } while (0)
#pragma unroll
    for (int d = 0; d < G64_D; ++d) G64_LOAD(d, min(d, nk - 1));
    G64_STORE(0, 0);
    __syncthreads();
    int kt0 = 0;
    for (; kt0 + G64_D <= nk; kt0 += G64_D) {           // register set d holds K tile kt0 + d (G64_D is even: LDS buffer = d & 1)
        G64_STEP(0, kt0); G64_STEP(1, kt0 + 1); G64_STEP(2, kt0 + 2); G64_STEP(3, kt0 + 3);
    }
    if (kt0 < nk) { G64_STEP(0, kt0); }                  // K % 256 != 0: up to three more tiles
    if (kt0 + 1 < nk) { G64_STEP(1, kt0 + 1); }
    if (kt0 + 2 < nk) { G64_STEP(2, kt0 + 2); }
#undef G64_STEP
#undef G64_LOAD
#undef G64_STORE
    gemm_epilogue<2, 2>(p, acc, m0 + wm * 32, n0 + wn * 32, n0, wn, g, li);
}

#define G2_T 256

// ---------------------------------------------------------------------------
// 256x256x64 tile, 8 phases per pair of K tiles: the deep-pipelined variant for plain GEMMs (K % 128 == 0).
//
// 8 waves = 2 (M) x 4 (N), each 128 x 64 of the output = 4 quadrants of 4 x 2 MFMA tiles.  A K tile is staged as four
// 16 KiB half-tiles in the order S = B0, A0, B1, A1 (B* = weight rows of the n-half of every wave column, A* = activation
// rows of the m-half of both wave rows), written straight into LDS by global_load_lds_dwordx4 (no staging registers)
// into a ring of 8 slots (2 K tiles, 128 KiB).  The LDS image of a wave instruction is lane-linear (8 rows x 128 B),
// so the bank swizzle (k slot ^ (row >> 1) & 7) is applied to the SOURCE address and again on the fragment read.
//
// Phase g (4 per K tile):  read the fragments of S[g+1] (8 x ds_read_b128 for an A half, 4 for a B half), issue the
// DMA of S[g+7] into slot (g+7) & 7 (its previous tenant S[g-1] was last read two phases ago), s_waitcnt vmcnt(10)
// (S[g+2] has landed, five half-tiles stay in flight across the barriers), barrier, 16 MFMAs of quadrant g & 3, barrier.
// The two wave rows run one barrier apart, so on every SIMD one wave issues MFMAs while the other reads LDS and
// issues the DMA.  A buffer is read one phase after the wait that retires it (+ the barrier), never in the same phase.
// ---------------------------------------------------------------------------
#define G8_SLOT 16384
#define G8_LDS (8 * G8_SLOT + 8 * 4096)      // ring + per-wave epilogue staging = all 160 KiB of the CU
#define G8_BAR()                                   \
    do {                                           \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
        asm volatile("" ::: "memory");             \
    } while (0)

struct G8Ctx {
    const char* A; const char* W;
    unsigned offA[2][2], offW[2][2];      // [half][instruction]: byte offset of this lane's 16 bytes at k = 0
    char* ring;                            // LDS ring base
    int wave_lds;                          // wave * 1024
    unsigned rdA, rdB;                     // this lane's fragment read offsets inside a half-tile (kk = 0; kk = 1 is ^ 64)
    int nk;
    // implicit-GEMM conv (k3 s2 p1): offA = byte offset of this lane's 16 bytes at tap (0,0), chunk 0, relative to Atile (the
    // tile's first image, moved back by one input row and one pixel so that every offset is >= 0); ahw = 9-bit mask of the
    // taps whose input pixel lies inside the image
    unsigned ahw[2][2];
    const char* Atile;
    unsigned kslot[2];                     // 8 * (k slot this lane stages), per instruction
    int Hin, Win, Cin, K;
    // plain GEMM (persistent over tiles): offsets are rebuilt from the tile origin at every issue
    int rr, rw;                            // row of the staged image this lane fills: A half = q*128 + h*64 + rr, W half = q*128 + h*32 + rw
    unsigned lda2, ldw2;                   // row pitches in bytes
    int M, N;
};

__device__ __attribute__((aligned(16))) unsigned g8_zero[4];               // source of padding taps and of k >= K

// DMA of half-tile S[s]: kind = s & 3 (0 B0, 1 A0, 2 B1, 3 A1), K tile = s >> 2 (clamped: the ring keeps turning past the end)
template <int KIND, int SLOT, int CONV, int WR = 128>   // WR: rows of the output tile per wave row (128; 112 in the 224-row tile)
__device__ __forceinline__ void g8_issue(const G8Ctx& c, int ktile, int tm0, int tn0) {
    const int kt = min(ktile, c.nk - 1);
    constexpr int h = KIND >> 1;
    char* dst = c.ring + SLOT * G8_SLOT + c.wave_lds;
    if constexpr (CONV != 1) {                                              // plain rows (bf16: 64 k per 128-byte tile row; fp8: 128 k)
        const unsigned kb = (unsigned)kt * 128u + 2u * c.kslot[0];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const char* src;
            if constexpr (KIND & 1) src = c.A + ((unsigned)min(tm0 + q * WR + h * 64 + c.rr, c.M - 1) * c.lda2 + kb);
            else src = c.W + ((unsigned)min(tn0 + q * 128 + h * 32 + c.rw, c.N - 1) * c.ldw2 + kb);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + 8192 * q), 16, 0, 0);
        }
    } else if constexpr ((KIND & 1) == 0) {                                 // weights: rows of K (chunk-major), zero beyond K
        const char* zero = reinterpret_cast<const char*>(g8_zero);
        const bool k_ok = kt * 64 < c.K;                                    // scalar (odd tile counts are padded to even)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const char* src = k_ok ? c.W + (c.offW[h][q] + (unsigned)kt * 128u) : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + 8192 * q), 16, 0, 0);
        }
    } else {                                                                // activations: im2col address per lane
        // chunk-major K: K tile kt = (64-channel chunk kt / 9, tap kt % 9) -- the nine taps of a chunk are consecutive, so
        // the input lines they share (2.1 taps per line on average) are re-read while still in L2.  Per lane and DMA only
        // an add, a mask test and the zero-line select remain here (the phase's VALU issue slots are shared with the
        // partner wave's MFMAs: 44 VALU per phase made this loop issue-bound, 2.15 us per K tile against 1.28 plain).
        const int chunk = kt / 9, tap = kt - 9 * chunk;                     // scalar
        const int kh = tap / 3, kw = tap - 3 * kh;
        const unsigned soff = (unsigned)(((kh * c.Win + kw) * c.Cin + chunk * 64) * 2);
        const int cmax = (kt * 64 < c.K) ? c.Cin - chunk * 64 : 0;          // channels of this chunk that exist (0: padding tile)
        const char* zero = reinterpret_cast<const char*>(g8_zero);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool in = ((c.ahw[h][q] >> tap) & 1u) != 0 && (int)c.kslot[q] < cmax;
            const char* src = in ? c.Atile + (c.offA[h][q] + soff) : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + 8192 * q), 16, 0, 0);
        }
    }
}

template <int SLOT, int NI = 4>
__device__ __forceinline__ void g8_read_a(const G8Ctx& c, bf16x8 (&af)[4][2]) {
    const char* b = c.ring + SLOT * G8_SLOT;
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
        af[ii][0] = *reinterpret_cast<const bf16x8*>(b + (c.rdA + ii * 2048));
        af[ii][1] = *reinterpret_cast<const bf16x8*>(b + ((c.rdA + ii * 2048) ^ 64u));
    }
}
template <int SLOT>
__device__ __forceinline__ void g8_read_b(const G8Ctx& c, bf16x8 (&bf)[2][2]) {
    const char* b = c.ring + SLOT * G8_SLOT;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        bf[jj][0] = *reinterpret_cast<const bf16x8*>(b + (c.rdB + jj * 2048));
        bf[jj][1] = *reinterpret_cast<const bf16x8*>(b + ((c.rdB + jj * 2048) ^ 64u));
    }
}
template <int MH, int NH, bool F8 = false, int MI1 = 4, int MIA = 4 + MI1>   // MI1: 16-row tiles in the second M half of a wave (4; 3: 112-row wave row; 2: 96-row); MIA: array rows
__device__ __forceinline__ void g8_mfma(f32x4 (&acc)[4][MIA], const bf16x8 (&af)[4][2], const bf16x8 (&bf)[2][2]) {
    if constexpr (F8) {
        // fp8: ONE v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit block scales) per output tile and half-tile of K: the CDNA4
        // instruction that runs fp8 at twice the bf16 rate (the plain 16x16x32 fp8_fp8 form, four per 128 k, runs at the bf16 rate:
        // MI355X_MICROARCH.md, matrix cores).  Lane (g, i) supplies 32 bytes of row / column i: its two 16-byte fragments of the
        // tile row (chunks g and g ^ 4), the same bytes for A and W, i.e. one permutation of k applied to both sides, which a sum over
        // k does not see (lane map checked with exact integers: tools/micro/mfma128_probe.hip).
        typedef int v4i __attribute__((ext_vector_type(4)));
        typedef int v8i __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int ii = 0; ii < (MH ? MI1 : 4); ++ii) {
            const v8i a = __builtin_shufflevector(__builtin_bit_cast(v4i, af[ii][0]), __builtin_bit_cast(v4i, af[ii][1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const v8i w = __builtin_shufflevector(__builtin_bit_cast(v4i, bf[jj][0]), __builtin_bit_cast(v4i, bf[jj][1]), 0, 1, 2, 3, 4, 5, 6, 7);
                acc[NH * 2 + jj][MH * 4 + ii] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, acc[NH * 2 + jj][MH * 4 + ii], 0, 0, 0, 0x7F7F7F7F,
                                                                                                 0, 0x7F7F7F7F);
            }
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < (MH ? MI1 : 4); ++ii)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[NH * 2 + jj][MH * 4 + ii] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[jj][kk], af[ii][kk], acc[NH * 2 + jj][MH * 4 + ii], 0, 0, 0);
    }
}

// One phase.  PH = g & 7; `tile` = K tile whose half-tile S[g+7] is issued ((g + 7) >> 2).
// b0cur: B0 fragments of the K tile being computed; b0nxt: filled in phase 3 (mod 4) for the next K tile.
template <int PH, int CONV, int MI1 = 4, int MIA = 4 + MI1, int WRS = 16 * (4 + MI1)>   // WRS: rows between the two wave rows' images (rows of wave row 0)
__device__ __forceinline__ void g8_phase(const G8Ctx& c, int issue_tile, int tm0, int tn0, f32x4 (&acc)[4][MIA], bf16x8 (&af)[4][2],
                                         bf16x8 (&b1)[2][2], bf16x8 (&b0cur)[2][2], bf16x8 (&b0nxt)[2][2]) {
    constexpr int RS = (PH + 1) & 7, RK = (PH + 1) & 3;          // slot / kind read in this phase
    if constexpr (RK == 1) g8_read_a<RS>(c, af);
    else if constexpr (RK == 3) g8_read_a<RS, MI1>(c, af);
    else if constexpr (RK == 2) g8_read_b<RS>(c, b1);
    else g8_read_b<RS>(c, b0nxt);
    g8_issue<(PH + 7) & 3, (PH + 7) & 7, CONV, WRS>(c, issue_tile, tm0, tn0);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr ((PH & 3) == 0) g8_mfma<0, 0, CONV == 2, MI1, MIA>(acc, af, b0cur);
    else if constexpr ((PH & 3) == 1) g8_mfma<0, 1, CONV == 2, MI1, MIA>(acc, af, b1);
    else if constexpr ((PH & 3) == 2) g8_mfma<1, 1, CONV == 2, MI1, MIA>(acc, af, b1);
    else g8_mfma<1, 0, CONV == 2, MI1, MIA>(acc, af, b0cur);
    __builtin_amdgcn_s_setprio(0);
    G8_BAR();
}

// Plain GEMMs run persistently: gridDim.x workgroups walk the tile list (tile v, v + gridDim.x, ...), and the DMA ring
// keeps turning across the seam, so the first seven half-tiles of the next output tile are already in flight while the
// last phases of the current one compute and while the epilogue stores; the wave rows re-align for the epilogue and
// stagger again after it.  The conv variant takes one tile per workgroup (its per-row im2col state is per tile).
// MI1 = 3 (plain GEMMs only): a 224-row tile, 112 rows per wave row = a full 64-row first half and a 48-row second half (three
// of its four MFMA tiles; the half-tile images keep their 128-row slots, the unused 16 rows per wave row are staged and ignored).
// Round 3: M = 12,480 / 13,056 rows make 196 / 408 / 784 / 816 tiles of 256 rows on 256 CUs -- rounds that are 77-80 % full --
// and 224 / 472 / 896 / 944 tiles of 224: the same number of rounds, each 0.875 of the work (gemm_launch picks per shape).
// Round 5: MI1B < MI1 gives the two wave rows different heights: MI1 = 3, MI1B = 2 is a 208-row tile (112 + 96 rows).  At M = 12,480 /
// 13,056 rows every 8-phase launch of the encoder and the prefill runs the same number of ROUNDS on 208-row tiles as on 224-row
// ones (60 / 63 row tiles against 56 / 59: N = 1,024 -> 240 tiles, one round; N = 2,048 -> 504, two; N = 4,096 -> 1,008, four), and a
// round costs its tile height (gemm_launch: rounds x height).  The second wave row's code is instantiated on its own (uniform branch
// per wave: the phase bodies and the epilogue exist once per wave-row height); same chains per output, same bits.
template <int CONV, int MI1 = 4, int MI1B = MI1>      // CONV 0: plain bf16, 1: implicit-GEMM conv (bf16), 2: plain fp8 (e4m3 x e4m3, scales in the epilogue)
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const GemmParams p, int total_tiles) {
    constexpr int MI = 4 + MI1, WR = 16 * MI, TM = WR + 16 * (4 + MI1B);   // MFMA row tiles of a wave of wave row 0 (= the accumulator array's rows), its rows, rows per tile
    static_assert(MI1 == 4 || (MI1 == 3 && CONV != 1), "the conv variant keeps 256-row tiles");
    static_assert(MI1B <= MI1 && MI1B >= 2, "second wave row: at most as high as the first");
    extern __shared__ __attribute__((aligned(16))) char smem8[];          // the only LDS object: 8 ring slots x 16 KiB + 8 x 4 KiB epilogue staging
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + G2_T - 1) / G2_T, tiles_m = (p.M + TM - 1) / TM;
    const int group = GROUP_M * tiles_n;
    auto tile_origin = [&](int v, int& m0, int& n0) {                       // same grouped, XCD-aware order as the other kernels
        const unsigned t = xcd_remap((unsigned)v, (unsigned)total_tiles);
        const int first_m = (t / group) * GROUP_M;
        const int gm = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (int)(t % group) % gm) * TM;
        n0 = ((int)(t % group) / gm) * G2_T;
    };
    int v = blockIdx.x, m0, n0, m0n, n0n;
    tile_origin(v, m0, n0);
    m0n = m0; n0n = n0;
    if (v + (int)gridDim.x < total_tiles) tile_origin(v + gridDim.x, m0n, n0n);
    const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, li = lane & 15;

    G8Ctx c;
    c.A = reinterpret_cast<const char*>(p.A);
    c.W = reinterpret_cast<const char*>(p.W);
    c.ring = smem8;
    c.wave_lds = wave * 1024;
    c.nk = CONV == 1 ? (((p.K + BK - 1) / BK + 1) & ~1) : (CONV == 2 ? p.K / 128 : p.K / BK);   // conv: K padded with zero taps to an even tile count; fp8: 128 k per tile
    c.Hin = p.Hin; c.Win = p.Win; c.Cin = p.Cin; c.K = p.K;
    c.M = p.M; c.N = p.N;
    c.lda2 = (unsigned)(p.lda * (CONV == 2 ? 1 : 2)); c.ldw2 = (unsigned)(p.ldw * (CONV == 2 ? 1 : 2));
    {
        const int rr = 8 * wave + (lane >> 3);                              // q*64 + rr = row of the half-tile image this lane fills
        c.rr = rr;
        c.rw = (rr >> 5) * 64 + (rr & 31);
        const int slot = (lane & 7) ^ ((rr >> 1) & 7);                      // k slot whose bytes belong at that position (same for q = 0, 1)
        c.kslot[0] = c.kslot[1] = 8 * slot;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            c.offA[h][q] = c.offW[h][q] = c.ahw[h][q] = 0;
            if constexpr (CONV == 1) {
                const int r = 64 * q + c.rr;
                const int rowA = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), p.M - 1);
                const int rowW = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), p.N - 1);
                c.offW[h][q] = (unsigned)(((long)rowW * p.ldw + c.kslot[q]) * 2);
                const int hw = p.Hout * p.Wout;
                const int img0 = m0 / hw;                                   // first image of this tile (uniform)
                const int img = rowA / hw, rem = rowA - img * hw;
                const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
                c.Atile = reinterpret_cast<const char*>(p.A) + ((long)img0 * p.Hin * p.Win - (p.Win + 1)) * p.Cin * 2;
                c.offA[h][q] = (unsigned)(((img - img0) * p.Hin * p.Win + 2 * oh * p.Win + 2 * ow) * p.Cin + (int)c.kslot[q]) * 2u;
                unsigned mask = 0;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ih = 2 * oh - 1 + tap / 3, iw = 2 * ow - 1 + tap % 3;
                    if ((unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) mask |= 1u << tap;
                }
                c.ahw[h][q] = mask;
            }
        }
    c.rdA = (unsigned)((wr * 64 + li) * 128 + ((g ^ (li >> 1)) << 4));
    c.rdB = (unsigned)((wc * 32 + li) * 128 + ((g ^ (li >> 1)) << 4));

    int n_stamp = 0;
#define G8_STAMP() do { if (p.stamp && (int)blockIdx.x == p.stamp_wg && tid == 0 && n_stamp < 16) p.stamp[n_stamp++] = wall_clock64(); } while (0)
    G8_STAMP();                                                             // 0: entry (after the address setup)

    // The whole tile loop of ONE wave-row height (M1 = 16-row tiles in the second half of this wave's rows).  An asymmetric tile
    // instantiates it twice, each wave running the copy of its row: with one copy branching per phase the register allocator
    // carried both rows' accumulator live ranges through every phase and spilled 137 dwords (ISA, round 5).  Both copies execute
    // the same sequence of workgroup barriers.
    auto tile_loop = [&](auto m1c) {
        constexpr int M1 = decltype(m1c)::value, MIW = 4 + M1;
        f32x4 acc[4][MIW];
        bf16x8 af[4][2], b1[2][2], b0e[2][2], b0o[2][2];
        // ---- prologue: S[0..6] in flight, S[0] (B0 of tile 0) and S[1] landed before the first reads
        g8_issue<0, 0, CONV, WR>(c, 0, m0, n0); g8_issue<1, 1, CONV, WR>(c, 0, m0, n0); g8_issue<2, 2, CONV, WR>(c, 0, m0, n0);
        g8_issue<3, 3, CONV, WR>(c, 0, m0, n0); g8_issue<0, 4, CONV, WR>(c, 1, m0, n0); g8_issue<1, 5, CONV, WR>(c, 1, m0, n0);
        g8_issue<2, 6, CONV, WR>(c, 1, m0, n0);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        G8_BAR();
        g8_read_b<0>(c, b0e);
        G8_STAMP();                                                         // 1: pipeline filled

        const int iters = c.nk >> 1;                                        // two K tiles (8 phases) per trip
        const int last_it = iters - 1;                                      // the DMA runs 7 half-tiles ahead: the last trip feeds the next tile
        while (true) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MIW; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (wr == 1) G8_BAR();                                         // the second wave row runs one barrier behind
            for (int it = 0; it < iters; ++it) {
                const int t2 = 2 * it;
                const bool seam = CONV != 1 && it == last_it;              // uniform: issues beyond this tile's K belong to the next tile
                const int im0 = seam ? m0n : m0, in0 = seam ? n0n : n0;
                const int wrap = seam ? c.nk : 0;
                g8_phase<0, CONV, M1, MIW, WR>(c, t2 + 1, m0, n0, acc, af, b1, b0e, b0o);            // S[8it+7]  = A1 of tile 2it+1
                g8_phase<1, CONV, M1, MIW, WR>(c, t2 + 2 - wrap, im0, in0, acc, af, b1, b0e, b0o);   // S[8it+8]  = B0 of tile 2it+2
                g8_phase<2, CONV, M1, MIW, WR>(c, t2 + 2 - wrap, im0, in0, acc, af, b1, b0e, b0o);
                g8_phase<3, CONV, M1, MIW, WR>(c, t2 + 2 - wrap, im0, in0, acc, af, b1, b0e, b0o);   // reads B0 of tile 2it+1 into b0o
                g8_phase<4, CONV, M1, MIW, WR>(c, t2 + 2 - wrap, im0, in0, acc, af, b1, b0o, b0e);   // S[8it+11] = A1 of tile 2it+2
                g8_phase<5, CONV, M1, MIW, WR>(c, t2 + 3 - wrap, im0, in0, acc, af, b1, b0o, b0e);
                g8_phase<6, CONV, M1, MIW, WR>(c, t2 + 3 - wrap, im0, in0, acc, af, b1, b0o, b0e);
                g8_phase<7, CONV, M1, MIW, WR>(c, t2 + 3 - wrap, im0, in0, acc, af, b1, b0o, b0e);   // reads B0 of tile 2it+2 into b0e
            }
            if (wr == 0) G8_BAR();                                         // re-align the wave rows
            G8_STAMP();                                                     // 2 + 2t: K loop of tile t done
            const int m_wave = m0 + wr * WR;                                // first row of this wave's rows (wave row 1 starts after wave row 0's WR rows)
            if constexpr (CONV == 2) {                                      // fp8: per-token and per-weight-row scales (fp32)
                int lane2 = lane;
                asm volatile("" : "+v"(lane2));                            // keep these addresses out of the K loop's live ranges
                const int g2 = lane2 >> 4, li2 = lane2 & 15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nb = min(n0 + wc * 64 + j * 16 + 4 * g2, p.N - 4);
                    const f32x4 sw = *reinterpret_cast<const f32x4*>(p.w_scale + nb);
#pragma unroll
                    for (int i = 0; i < MIW; ++i) {
                        const float sa = p.a_scale[min(m_wave + i * 16 + li2, p.M - 1)];
                        acc[j][i] *= sw * sa;
                    }
                }
            }
            // all three variants (plain, conv, fp8) carry both the interior and the general path
            if (p.lds_epi) gemm_epilogue_lds<MIW, MIW, CONV == 0>(p, acc, m_wave, n0 + wc * 64, n0, wc, smem8 + 8 * G8_SLOT + wave * 4096, lane);
            else gemm_epilogue<MIW>(p, acc, m_wave, n0 + wc * 64, n0, wc, g, li);
            G8_STAMP();                                                     // 3 + 2t: epilogue of tile t issued
            v += gridDim.x;
            if (CONV == 1 || v >= total_tiles) break;                      // uniform
            m0 = m0n; n0 = n0n;
            if (v + (int)gridDim.x < total_tiles) tile_origin(v + gridDim.x, m0n, n0n);
        }
    };
    if (MI1B == MI1 || wr == 0) tile_loop(std::integral_constant<int, MI1>{});
    else tile_loop(std::integral_constant<int, MI1B>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the ring's tail DMAs (clamped re-loads) must land before exit
}

static int g_gemm_mode = 0;                                  // q3asr_gemm_force_kernel
static int g_gemm_interior = 2;                              // q3asr_gemm_set_interior (A/B hook): 0 = every tile takes the general epilogue,
                                                             // 1 = interior sub-tiles take the short one, 2 = ... and read their residual rows one block ahead
extern "C" int q3asr_gemm_set_interior(int on) { int was = g_gemm_interior; g_gemm_interior = on < 0 ? 0 : (on > 2 ? 2 : on); return was; }
static unsigned long long* g_gemm_stamp = nullptr;           // q3asr_gemm_set_stamp

// Diagnostic hook: workgroup 0 of every following 8-phase launch writes up to 16 wall_clock64() stamps (100 MHz) to `buf`
// (device memory, 16 x 8 bytes): entry, pipeline filled, then per tile "K loop done" / "epilogue issued".  nullptr switches
// it off.  tools/gemm_stamps.py prints where a tile's time goes.
extern "C" int q3asr_gemm_set_stamp(void* buf) { g_gemm_stamp = (unsigned long long*)buf; return 0; }
static int g_gemm_stamp_wg = 0;
extern "C" int q3asr_gemm_set_stamp_wg(int wg) { g_gemm_stamp_wg = wg; return 0; }     // which workgroup (blockIdx.x) stamps

extern "C" int q3asr_gemm_force_kernel(int mode) {
    Q3_REQUIRE(mode == 0 || mode == 128 || mode == 8 || mode == 64 || mode == 224 || mode == 208, "gemm_force_kernel: mode %d (0 auto, 64, 128, 8 = 8-phase with 256-row tiles, 224 / 208 = 8-phase with 224- / 208-row tiles)", mode);
    g_gemm_mode = mode;
    return 0;
}

static void gemm_launch(const GemmParams& p, bool conv, hipStream_t s, int force_mode = 0) {
    const unsigned tiles = (unsigned)((p.M + BM - 1) / BM) * (unsigned)((p.N + BN - 1) / BN);
    const bool fast = !conv && p.K % BK == 0 && (long)p.M * p.lda < (1L << 31) && (long)p.N * p.ldw < (1L << 31);
    const unsigned tiles256 = (unsigned)((p.M + G2_T - 1) / G2_T) * (unsigned)((p.N + G2_T - 1) / G2_T);
    const int mode = force_mode ? force_mode : g_gemm_mode;  // 0 auto; 64 / 128 / 8 / 224 pin a kernel (A/B runs, tests)
    // Kernel choice (measured on the prefill / encoder shapes, tools/bench_gemm.py): the 8-phase kernel wins wherever its
    // preconditions hold and the grid fills the chip: one 256^2 tile costs ~2.6 128^2 tiles, one workgroup per CU against
    // two (fc1 with its erf-GELU epilogue: 156 us against 181 us).
    const bool can8 = fast && p.K % 128 == 0 && p.K >= 256;
    const unsigned rounds8 = (tiles256 + 255) / 256, rounds128 = (tiles + 255) / 256;
    const bool want8 = can8 && 2.6f * rounds8 < 1.0f * rounds128;
    // a 256-row tile spans at most 256 / (Hout*Wout) + 2 images: their bytes must fit the 32-bit lane offsets
    const bool can8c = conv && p.K >= 256 && (long)p.N * p.ldw < (1L << 31) &&
                       (256L / (p.Hout * p.Wout) + 3) * p.Hin * p.Win * p.Cin * 2 < (1L << 31);
    const bool want8c = can8c && tiles256 >= 192;
    if (((mode == 8 || mode == 224 || mode == 208) && (can8 || can8c)) || (mode == 0 && (want8 || want8c))) {      // deep-pipelined 8-phase kernel
        // Tried (round 3, commit 'Experiment: 256x128 half-tile GEMM ...', profiles/r03_gemm2w_experiment.txt): 256 x 128 tiles, 4 waves and
        // an 80 KiB ring so that TWO workgroups share a CU (one's epilogue and pipeline fill under the other's K loop, half-size
        // scheduling units): bit-identical and race-screened, but no faster -- fc1 144.9 vs 143.7 us, encoder qkv 85.9 vs 82.3,
        // prefill down 322 vs 288: the half-tile moves 1.5x the operand bytes per flop, so a wave's serial LDS-read + DMA-issue
        // time per K tile (620 + 700 cycles, + 340 of waits) exceeds its 1,024 MFMA cycles and the pipes idle 30 % of the loop,
        // which eats what the overlap (two workgroups resident 55-88 % of the time) and the finer tail give back.
        // Tried (round 5, commit 53a19f3 'experiment: chained stream-K', profiles/r05_streamk_experiment.txt): every workgroup the same number
        // of K iterations, a tile shared by two workgroups CONTINUED from published accumulators (256 KiB fp32 hand-off, bounded wait,
        // recomputation as the fallback) -- bit-identical in every test, and 1-14 % SLOWER on every eligible shape (fc1 132.6 -> 143.9 us,
        // prefill o 109.2 -> 124.6): a K = 1024 tile's loop takes 20.6 us beside neighbours that are all loading and 25.7 us beside
        // neighbours that are storing, so desynchronised epilogues slow every K loop instead of hiding behind them.  Not kept.
        // Tried (round 2, tools/ab_gemm_split.py): giving the rows of a mostly empty last round (784 tiles = 3.06 rounds) to the
        // 128^2 kernel as a second launch -- fc1 142.5 -> 139.4 us, encoder qkv 88.7 -> 95.3 us, prefill qkv 200.6 -> 198.7 us: a
        // K = 1024 tile of the 128^2 kernel plus the extra launch costs what the idle CUs of the last round cost.  Not kept.
        GemmParams p8 = p;
        p8.stamp = g_gemm_stamp; p8.stamp_wg = g_gemm_stamp_wg;
        const long esz = p.out_kind == 1 ? 4 : 2;
        p8.lds_epi = p.N % 8 == 0 && p.ldo % 8 == 0 && ((uintptr_t)p.out & 15) == 0 && (p.ldo * esz) % 16 == 0 &&
                     (p.res_kind == 0 || (p.ldres % 8 == 0 && ((uintptr_t)p.res & 15) == 0)) &&
                     (!p.bias || ((uintptr_t)p.bias & 15) == 0);
        // 2: interior sub-tiles may take the short epilogue (32-bit lane offsets: the output and the residual must span < 4 Gi elements)
        if (p8.lds_epi && g_gemm_interior && (long)p.M * p.ldo < (1L << 32) && (p.res_kind == 0 || (long)p.M * p.ldres < (1L << 32))) p8.lds_epi = 1 + g_gemm_interior;
        static std::atomic<bool> attr8{false};
        if (!attr8) {
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<0, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
            attr8 = true;
        }
        if (conv) hipLaunchKernelGGL(gemm8p_kernel<1>, dim3(tiles256), dim3(512), G8_LDS, s, p8, (int)tiles256);
        else {
            // persistent (one workgroup per CU walking the tile list) where tiles are short and the epilogue light:
            // it hides the dispatch and the pipeline fill of every tile after the first, but its epilogue stores must
            // drain before the next tile's counted waits can pass (measured: +8 % at K <= 2048, -2..3 % on long-K or
            // SwiGLU tiles, tools/bench_gemm.py)
            const bool persist = p.K <= 2048 && p.act != 2;
            // Tile height: a workgroup owns a CU, so a launch costs about rounds x tile height: 224-row tiles where that is less than with
            // 256 rows (M = 12,480: 196 tiles of 256 rows or 224 of 224 for N = 1,024 -- one round either way, the second 1/8 shorter),
            // and 208-row tiles (112 + 96) where that is less again AND the K loop is short: at K = 1,024 (encoder fc1 133.4 / 128.8 / 120.5 us
            // and qkv 82.9 / 81.1 / 76.0 us at 256 / 224 / 208 rows) the rounds of a launch stay in step and its time follows the tile
            // height; at K >= 2,048 the 224- and 208-row forms measured within 1 % of each other (prefill qkv 176.8 / 178.0, o 108.1 /
            // 108.9, down 264.0 / 265.9 us) and the SwiGLU launch is fastest on 256 rows (497 / 534 / 529 us) -- tools/ab_gemm_height.py.
            const unsigned tn = (unsigned)((p.N + G2_T - 1) / G2_T);
            const unsigned tiles224 = (unsigned)((p.M + 223) / 224) * tn, tiles208 = (unsigned)((p.M + 207) / 208) * tn;
            const unsigned c256 = ((tiles256 + 255) / 256) * 256u, c224 = ((tiles224 + 255) / 256) * 224u, c208 = ((tiles208 + 255) / 256) * 208u;
            const int height = mode == 224 ? 224 : mode == 208 ? 208 : mode == 8 ? 256
                             : (c224 < c256 ? (p.K <= 1024 && c208 < c224 ? 208 : 224) : 256);
            const unsigned tiles8 = height == 208 ? tiles208 : (height == 224 ? tiles224 : tiles256);
            const unsigned grid = persist && tiles8 > 256u ? 256u : tiles8;
            if (height == 208) hipLaunchKernelGGL((gemm8p_kernel<0, 3, 2>), dim3(grid), dim3(512), G8_LDS, s, p8, (int)tiles8);
            else if (height == 224) hipLaunchKernelGGL((gemm8p_kernel<0, 3>), dim3(grid), dim3(512), G8_LDS, s, p8, (int)tiles8);
            else hipLaunchKernelGGL(gemm8p_kernel<0>, dim3(grid), dim3(512), G8_LDS, s, p8, (int)tiles8);
        }
        return;
    }
    // a 128^2 grid that leaves CUs idle: 64^2 tiles (one clip: M = 390 / 408 rows; small admission groups of the front door).
    // Measured (tools/bench_gemm_small.py, profiles/r03_gemm_small.txt): one clip's prefill o 27.0 -> 13.2 us, down 72.5 -> 48.0,
    // qkv 27.5 -> 18.8, encoder fc2 49.7 -> 32.6; a win up to 168 128^2-tiles, a tie at 208-224, a loss above.
    const unsigned tiles64 = (unsigned)((p.M + 63) / 64) * (unsigned)((p.N + 63) / 64);
    if (fast && (mode == 64 || (mode == 0 && tiles <= 176u))) {
        hipLaunchKernelGGL(gemm64_kernel, dim3(tiles64), dim3(256), 0, s, p);
        return;
    }
    if (conv) hipLaunchKernelGGL(gemm_bf16_kernel<1>, dim3(tiles), dim3(256), 0, s, p);
    else if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<0, true>), dim3(tiles), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_bf16_kernel<0>, dim3(tiles), dim3(256), 0, s, p);
}

static int gemm_validate(const GemmParams& p, const char* who) {
    Q3_REQUIRE(p.A && p.W && p.out, "%s: null pointer", who);
    Q3_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "%s: M=%d N=%d K=%d must be positive", who, p.M, p.N, p.K);
    Q3_REQUIRE(p.K % 8 == 0 && p.N % 4 == 0, "%s: K=%d must be a multiple of 8 and N=%d of 4", who, p.K, p.N);
    Q3_REQUIRE(p.ldw % 8 == 0 && p.ldw >= p.K, "%s: ldw=%ld must be >= K and a multiple of 8", who, p.ldw);
    Q3_REQUIRE(p.act >= 0 && p.act <= 2 && p.res_kind >= 0 && p.res_kind <= 2 && p.out_kind >= 0 && p.out_kind <= 1,
               "%s: bad epilogue selector", who);
    Q3_REQUIRE(p.res_kind == 0 || p.res, "%s: residual kind set without pointer", who);
    if (p.act == 2) {
        Q3_REQUIRE(p.N % 32 == 0 && p.out_kind == 0 && !p.bias && p.res_kind == 0,
                   "%s: SwiGLU epilogue needs N%%32==0, bf16 out, no bias/residual", who);
        Q3_REQUIRE(p.ldo % 4 == 0 && p.ldo >= p.N / 2, "%s: ldo=%ld too small for SwiGLU output", who, p.ldo);
    } else {
        Q3_REQUIRE(p.ldo % 4 == 0 && p.ldo >= p.N, "%s: ldo=%ld must be >= N and a multiple of 4", who, p.ldo);
    }
    Q3_REQUIRE(p.res_kind == 0 || (p.ldres % 4 == 0 && p.ldres >= p.N), "%s: bad ldres=%ld", who, p.ldres);
    return Q3_OK;
}

extern "C" int q3asr_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K,
                               const float* bias, int act, const void* res, int res_kind, int64_t ldres, int res_mod,
                               void* out, int out_kind, int64_t ldo, const int32_t* row_map, void* stream) {
    GemmParams p{};
    p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.act = act; p.res = res; p.res_kind = res_kind; p.ldres = ldres; p.res_mod = res_mod;
    p.out = out; p.out_kind = out_kind; p.ldo = ldo; p.row_map = row_map;
    if (int rc = gemm_validate(p, "gemm_bf16")) return rc;
    Q3_REQUIRE(lda % 8 == 0 && lda >= K, "gemm_bf16: lda=%ld must be >= K and a multiple of 8", (long)lda);
    gemm_launch(p, false, (hipStream_t)stream);
    return q3_check_launch("gemm_bf16");
}

// Conv2d k3 s2 p1 over NHWC bf16 as implicit GEMM (encoder.py:155-168,274-275).
// x [imgs][Hin][Win][Cin]; w in CHUNK-MAJOR K order [Cout][ceil(Cin/64)][9 taps][64] (tap = 3*kh + kw over x's two spatial
// dims; channels beyond Cin are zero) = ops.conv_weight_chunked: the nine taps of a 64-channel chunk are consecutive K
// tiles, so the input lines neighbouring taps share are re-read from L2 instead of from beyond it (r1 PMC: 3.4x the
// input bytes fetched with tap-major K).  All three kernels walk K in this order: bit-identical results.
// out row m = (img*Hout + oh)*Wout + ow, or row_map[m].
extern "C" int q3asr_conv3x3s2_bf16(const void* x, int imgs, int Hin, int Win, int Cin, const void* w, int Cout,
                                    const float* bias, int act, void* out, int out_kind, int64_t ldo,
                                    const int32_t* row_map, void* stream) {
    GemmParams p{};
    const int kc = ((Cin + 63) / 64) * 9 * 64;                             // chunk-major K, channels padded to 64
    p.A = (const bf16*)x; p.lda = 0; p.W = (const bf16*)w; p.ldw = kc;
    p.Hin = Hin; p.Win = Win; p.Cin = Cin;
    p.Hout = (Hin - 1) / 2 + 1; p.Wout = (Win - 1) / 2 + 1;
    p.M = imgs * p.Hout * p.Wout; p.N = Cout; p.K = kc;
    p.bias = bias; p.act = act; p.out = out; p.out_kind = out_kind; p.ldo = ldo; p.row_map = row_map;
    Q3_REQUIRE(imgs > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cin % 8 == 0, "conv3x3s2: bad geometry (Cin=%d must be a multiple of 8)", Cin);
    Q3_REQUIRE((long)imgs * p.Hout * p.Wout < 2147483647L, "conv3x3s2: M overflows int32");
    if (int rc = gemm_validate(p, "conv3x3s2")) return rc;
    Q3_REQUIRE(act != 2, "conv3x3s2: SwiGLU epilogue not valid here");
    gemm_launch(p, true, (hipStream_t)stream);
    return q3_check_launch("conv3x3s2");
}


// ---------------------------------------------------------------------------
// fp8 x fp8 GEMM on the CDNA4 fp8 MFMA (BASELINE.json configs[4]):  C[M,N] = (A8 . W8^T) * a_scale[m] * w_scale[n], then the
// same epilogues as q3asr_gemm_bf16 (SwiGLU, bf16 residual).  A8 = activations quantised per token (row) to OCP e4m3 by
// q3asr_rmsnorm_bf16_fp8 / q3asr_quantize_rows_fp8, W8 = the checkpoint weight quantised per output row at load.  Runs on the
// 8-phase kernel: a tile row is 128 bytes = 128 k (64 for bf16); a lane's two 16-byte fragments of a row are one operand of
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales), so per unit of K the MFMA cycles AND the LDS / L2 / HBM operand bytes are half
// those of the bf16 kernel (rounds 2-4 issued four v_mfma_f32_16x16x32_fp8_fp8 per 128 k, which run at the bf16 rate).
// ---------------------------------------------------------------------------
extern "C" int q3asr_gemm_fp8(const void* A8, int64_t lda, const float* a_scale, const void* W8, int64_t ldw, const float* w_scale,
                              int M, int N, int K, int act, const void* res, int res_kind, int64_t ldres, void* out, int64_t ldo,
                              void* stream) {
    GemmParams p{};
    p.A = (const bf16*)A8; p.lda = lda; p.W = (const bf16*)W8; p.ldw = ldw; p.M = M; p.N = N; p.K = K;
    p.act = act; p.res = res; p.res_kind = res_kind; p.ldres = ldres; p.out = out; p.out_kind = 0; p.ldo = ldo;
    p.a_scale = a_scale; p.w_scale = w_scale;
    Q3_REQUIRE(A8 && W8 && a_scale && w_scale && out, "gemm_fp8: null pointer");
    Q3_REQUIRE(M > 0 && N > 0 && K >= 256 && K % 256 == 0 && N % 8 == 0, "gemm_fp8: M=%d N=%d K=%d (K a multiple of 256, N of 8)", M, N, K);
    Q3_REQUIRE(lda >= K && ldw >= K && lda % 16 == 0 && ldw % 16 == 0 && ((uintptr_t)A8 & 15) == 0 && ((uintptr_t)W8 & 15) == 0,
               "gemm_fp8: rows must be 16-byte aligned (lda=%ld ldw=%ld)", (long)lda, (long)ldw);
    Q3_REQUIRE((long)M * lda < (1L << 32) && (long)N * ldw < (1L << 32), "gemm_fp8: operand larger than 4 GiB");
    Q3_REQUIRE((act == 0 || act == 2) && (res_kind == 0 || res_kind == 2), "gemm_fp8: epilogues are none / SwiGLU and none / bf16 residual");
    Q3_REQUIRE(res_kind == 0 || res, "gemm_fp8: residual kind set without pointer");
    if (act == 2) Q3_REQUIRE(N % 32 == 0 && res_kind == 0 && ldo % 8 == 0 && ldo >= N / 2, "gemm_fp8: SwiGLU needs N%%32==0, no residual, ldo >= N/2");
    else Q3_REQUIRE(ldo % 8 == 0 && ldo >= N, "gemm_fp8: ldo=%ld must be >= N and a multiple of 8", (long)ldo);
    Q3_REQUIRE(((uintptr_t)out & 15) == 0 && ((uintptr_t)w_scale & 15) == 0 && (res_kind == 0 || (ldres % 8 == 0 && ((uintptr_t)res & 15) == 0)),
               "gemm_fp8: out / w_scale / res must be 16-byte aligned");
    p.lds_epi = (g_gemm_interior && (long)M * ldo < (1L << 32) && (res_kind == 0 || (long)M * ldres < (1L << 32))) ? 1 + g_gemm_interior : 1;   // as in gemm_launch
    const unsigned tiles256 = (unsigned)((M + G2_T - 1) / G2_T) * (unsigned)((N + G2_T - 1) / G2_T);
    static std::atomic<bool> attr{false};
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
        attr = true;
    }
    // tile height as in gemm_launch (q3asr_gemm_force_kernel 8 / 224 / 208 pin one); K counts fp8 elements: 1,024 of them are 8 K tiles
    const unsigned tn = (unsigned)((N + G2_T - 1) / G2_T);
    const unsigned tiles224 = (unsigned)((M + 223) / 224) * tn, tiles208 = (unsigned)((M + 207) / 208) * tn;
    const unsigned c256 = ((tiles256 + 255) / 256) * 256u, c224 = ((tiles224 + 255) / 256) * 224u, c208 = ((tiles208 + 255) / 256) * 208u;
    const int height = g_gemm_mode == 224 ? 224 : g_gemm_mode == 208 ? 208 : g_gemm_mode == 8 ? 256
                     : (c224 < c256 ? (K <= 1024 && c208 < c224 ? 208 : 224) : 256);
    const unsigned tiles8 = height == 208 ? tiles208 : (height == 224 ? tiles224 : tiles256);
    const bool persist = act != 2 && tiles8 > 256u;
    if (height == 208) hipLaunchKernelGGL((gemm8p_kernel<2, 3, 2>), dim3(persist ? 256u : tiles8), dim3(512), G8_LDS, (hipStream_t)stream, p, (int)tiles8);
    else if (height == 224) hipLaunchKernelGGL((gemm8p_kernel<2, 3>), dim3(persist ? 256u : tiles8), dim3(512), G8_LDS, (hipStream_t)stream, p, (int)tiles8);
    else hipLaunchKernelGGL(gemm8p_kernel<2>, dim3(persist ? 256u : tiles8), dim3(512), G8_LDS, (hipStream_t)stream, p, (int)tiles8);
    return q3_check_launch("gemm_fp8");
}
