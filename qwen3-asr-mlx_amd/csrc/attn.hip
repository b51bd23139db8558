// Attention kernels (mx.fast.scaled_dot_product_attention call sites):
//  * attn_fwd<HD, CAUSAL>: flash-style forward over independent segments.
//      - encoder (encoder.py:82-84): HD=64, bidirectional, one segment per 104-token window
//        (the block-diagonal mask of encoder.py:210-229 is never materialised);
//      - decoder prefill (decoder.py:158-163): HD=128, causal, GQA, one segment per sequence,
//        K/V read from the KV cache the RoPE kernel just filled.
// (The decode step's attention, one new token per sequence against its cache, is attn_decode_fused_kernel in decode.hip.)
//
// attn_fwd maths per 64-key tile, per wave (16 query rows), MFMA 16x16x32 bf16:
//   S^T = K . Q^T      A operand = K rows (ds_read_b128 from the swizzled LDS tile), B = Q (registers)
//                      -> lane (g=lane>>4, li=lane&15) holds S[key=16*kt4+4g+r][q=li]: a query's scores
//                         live in 4 lanes, so row max/sum need two shuffles (xor 16, 32) only.
//   O^T += V^T . P^T   B operand = P straight from the S registers (bf16), with the k index of the
//                      MFMA re-labelled key(c,g,j) = 32c + 16(j>>2) + 4g + (j&3); A operand = V^T
//                      gathered with that same labelling by two ds_read_b64_tr_b16 per fragment.
//                      -> O^T[d=16dt+4g+r][q=li]: the softmax rescale is lane-local.
// Softmax in fp32; P rounded to bf16 for the PV MFMA; the row sum uses unrounded P.
#include "common.h"

#define KV_TILE 64

template <int HD> struct TileImg;
// 256-byte rows: 16-byte chunk ch of row r sits at 256 r + 16 (ch ^ f(r)), f(r) = (r & 7) << 1.  Round 4: the f of rounds 1-3,
// ((r & 3) << 2) | ((r >> 2) & 3), was 2-way conflicted on both read patterns -- in a ds_read_b128 lane group (lanes 0-3, 12-15,
// 20-27: rows li = 0-3 and 12-15 at chunk 4 ks, rows 4-11 at chunk 4 ks + 1) rows 4-7 landed on the slots of rows 0-3, and in
// a ds_read_b64_tr_b16 half-wave (8 rows x chunks 2 dt, 2 dt + 1) rows r and r + 4 swapped slots: SQ_LDS_BANK_CONFLICT was
// 39 % of SQ_LDS_IDX_ACTIVE in the prefill attention (tools/pmc_attn.sh).  This f is conflict-free for the K row reads, the
// transposing V reads and the staging writes under the bank model of MI355X_MICROARCH.md (64 banks x 4 B; lane groups per
// instruction), found by exhaustive search over the XOR-linear f (results are unchanged: a layout, not arithmetic).
template <> struct TileImg<128> {
    static __device__ __forceinline__ int off(int row, int ch) { return 256 * row + 16 * (ch ^ ((row & 7) << 1)); }
};
template <> struct TileImg<64> {
    static __device__ __forceinline__ int off(int row, int ch) { return 128 * row + 16 * (ch ^ (row & 7)); }
};

struct AttnParams {
    const bf16* Q; long q_stride;          // row stride (elements); head h at column h*HD
    const bf16* K; const bf16* V;
    const long* kv_seg_off;                // per segment: element offset of (key 0, kv head 0)
    long kv_head_stride, kv_row_stride;
    bf16* O; long o_stride;
    const int* seg_q_start;                // first Q/O row of the segment
    const int* seg_len;
    int group;                             // query heads per kv head
    float scale;
};

// Occupancy: 3 workgroups per CU (round 3; the compiler's free choice was 170 VGPRs = 2): 86-93 -> 76-77 us per prefill layer at
// 32 x 408 tokens.  Measured beside it and not kept: both query heads of a kv head in one 512-thread workgroup (K/V tile fetched
// and written to LDS once for two heads, bit-identical): 76.0 vs 77.0 us -- staging is not what a tile costs; a wave's serial
// QK^T -> softmax (~170 VALU per 32 MFMAs) -> PV chain is, at three waves per SIMD.
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KV_TILE * HD * 2];
    char* sK = smem;
    char* sV = smem + KV_TILE * HD * 2;
    // causal: a query tile's work grows with its index -- dispatch the heavy tiles of a (segment, head) first
    const int seg = blockIdx.z, head = blockIdx.y, qt = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int len = p.seg_len[seg];
    if (qt * 64 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int q_abs = qt * 64 + wave * 16 + li;
    const int q_eff = min(q_abs, len - 1);
    const long qrow = (long)p.seg_q_start[seg] + q_eff;
    constexpr int KS = HD / 32;            // k-steps of the S MFMA
    constexpr int DT = HD / 16;            // 16-wide d tiles of O^T
    constexpr int CPR = HD / 8;            // 16-byte chunks per row

    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(p.Q + qrow * p.q_stride + head * HD + 32 * ks + 8 * g);

    const bf16* Kb = p.K + p.kv_seg_off[seg] + (long)(head / p.group) * p.kv_head_stride;
    const bf16* Vb = p.V + p.kv_seg_off[seg] + (long)(head / p.group) * p.kv_head_stride;

    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int n_tiles = CAUSAL ? (qt + 1) : (len + KV_TILE - 1) / KV_TILE;
    // K and V tile: KV_TILE*CPR 16-byte chunks each, 256 threads; the next tile's chunks are loaded into registers while
    // the current tile computes (one HBM/L2 round trip per tile would otherwise sit between two barriers)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NCH = KV_TILE * CPR / 256;
    u32x4 kreg[NCH], vreg[NCH];
    auto fetch = [&](int kt_) {
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int c = tid + 256 * it;
            const int row = c / CPR, ch = c - row * CPR;
            const int key = min(kt_ * KV_TILE + row, len - 1);
            kreg[it] = *reinterpret_cast<const u32x4*>(Kb + (long)key * p.kv_row_stride + ch * 8);
            vreg[it] = *reinterpret_cast<const u32x4*>(Vb + (long)key * p.kv_row_stride + ch * 8);
        }
    };
    fetch(0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int c = tid + 256 * it;
            const int row = c / CPR, ch = c - row * CPR;
            *reinterpret_cast<u32x4*>(sK + TileImg<HD>::off(row, ch)) = kreg[it];
            *reinterpret_cast<u32x4*>(sV + TileImg<HD>::off(row, ch)) = vreg[it];
        }
        if (kt + 1 < n_tiles) fetch(kt + 1);
        __syncthreads();

        // ---- S^T = K Q^T
        f32x4 s[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            s[k4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + TileImg<HD>::off(16 * k4 + li, 4 * ks + g));
                s[k4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[k4], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (row = query li, spread over the 4 lane groups)
        float tmax = -INFINITY;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * KV_TILE + 16 * k4 + 4 * g + r;
                const bool ok = key < len && (!CAUSAL || key <= q_eff);
                const float v = ok ? s[k4][r] * p.scale : -INFINITY;
                s[k4][r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);          // finite: key 0 is visible to every query
        const float alpha = __expf(m_run - m_new);       // m_run = -inf on the first tile -> 0
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(s[k4][r] - m_new);
                psum += pv;
                pf[k4 >> 1][(k4 & 1) * 4 + r] = f2bf(pv);
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
        // ---- O^T += V^T P^T
        const int q4 = li >> 2, p4 = li & 3;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int ch = 2 * dt + (p4 >> 1);
                const int r_lo = 32 * c + 4 * g + q4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sV + TileImg<HD>::off(r_lo, ch) + 8 * (p4 & 1)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sV + TileImg<HD>::off(r_lo + 16, ch) + 8 * (p4 & 1)));
                union { struct { s16x4 a, b; } h; bf16x8 v; } u;
                u.h.a = lo;
                u.h.b = hi;
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf[c], o[dt], 0, 0, 0);
            }
        }
    }
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    if (q_abs < len) {
        const float inv = 1.0f / l_tot;
        bf16* orow = p.O + ((long)p.seg_q_start[seg] + q_abs) * p.o_stride + head * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[dt][r] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * dt + 4 * g) = ov;
        }
    }
}

extern "C" int q3asr_attn_fwd(const void* Q, int64_t q_stride, const void* K, const void* V, const int64_t* kv_seg_off,
                              int64_t kv_head_stride, int64_t kv_row_stride, void* O, int64_t o_stride,
                              const int32_t* seg_q_start, const int32_t* seg_len, int n_segs, int max_len, int n_heads,
                              int n_kv_heads, int head_dim, int causal, float scale, void* stream) {
    Q3_REQUIRE(Q && K && V && kv_seg_off && O && seg_q_start && seg_len, "attn_fwd: null pointer");
    Q3_REQUIRE(n_segs > 0 && max_len > 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0, "attn_fwd: bad sizes");
    Q3_REQUIRE(q_stride % 8 == 0 && kv_row_stride % 8 == 0 && kv_head_stride % 8 == 0 && o_stride % 4 == 0, "attn_fwd: strides must keep 16-byte alignment");
    AttnParams p;
    p.Q = (const bf16*)Q; p.q_stride = q_stride; p.K = (const bf16*)K; p.V = (const bf16*)V;
    p.kv_seg_off = (const long*)kv_seg_off; p.kv_head_stride = kv_head_stride; p.kv_row_stride = kv_row_stride;
    p.O = (bf16*)O; p.o_stride = o_stride; p.seg_q_start = seg_q_start; p.seg_len = seg_len;
    p.group = n_heads / n_kv_heads; p.scale = scale;
    dim3 grid((max_len + 63) / 64, n_heads, n_segs), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, blk, 0, s, p);
    else if (head_dim == 128 && causal) hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, blk, 0, s, p);
    else if (head_dim == 128 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, blk, 0, s, p);
    else if (head_dim == 64 && causal) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, blk, 0, s, p);
    else { q3asr_set_error("attn_fwd: head_dim=%d unsupported (64 or 128)", head_dim); return Q3_EBADARG; }
    return q3_check_launch("attn_fwd");
}
