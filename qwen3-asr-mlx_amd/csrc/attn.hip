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
#include <type_traits>

#define KV_TILE 64

template <int HD> struct TileImg;
// 256-byte rows: 16-byte chunk ch of row r sits at 256 r + 16 (ch ^ f(r)), f(r) = (r & 7) << 1.  Round 4: the f of rounds 1-3,
// ((r & 3) << 2) | ((r >> 2) & 3), was 2-way conflicted on both read patterns -- in a ds_read_b128 lane group (lanes 0-3, 12-15,
// 20-27: rows li = 0-3 and 12-15 at chunk 4 ks, rows 4-11 at chunk 4 ks + 1) rows 4-7 landed on the slots of rows 0-3, and in
// a ds_read_b64_tr_b16 half-wave (8 rows x chunks 2 dt, 2 dt + 1) rows r and r + 4 swapped slots: SQ_LDS_BANK_CONFLICT was
// 39 % of SQ_LDS_IDX_ACTIVE in the prefill attention (tools/pmc_attn.sh).  This f is conflict-free for the K row reads, the
// transposing V reads and the staging writes under the bank model of MI355X_MICROARCH.md (64 banks x 4 B; lane groups per
// instruction), found by exhaustive search over the XOR-linear f (results are unchanged: a layout, not arithmetic).
// Both images depend on the row only through row & 7: off(row + 8 j, ch) = off(row, ch) + 8 j ROWB.
template <> struct TileImg<128> {
    static constexpr int ROWB = 256;
    static __device__ __forceinline__ int off(int row, int ch) { return 256 * row + 16 * (ch ^ ((row & 7) << 1)); }
};
template <> struct TileImg<64> {
    static constexpr int ROWB = 128;
    static __device__ __forceinline__ int off(int row, int ch) { return 128 * row + 16 * (ch ^ (row & 7)); }
};

struct AttnParams {
    const bf16* Q; long q_stride;          // row stride (elements); head h at column h*HD
    const bf16* K; const bf16* V;
    const long* kv_seg_off;                // per segment: element offset of (key 0, kv head 0)
    long kv_head_stride, kv_row_stride;
    bf16* O; long o_stride;
    const int* seg_q_start;                // first Q/O row of the segment
    const int* seg_len;                    // query rows of the segment
    const int* seg_k_off;                  // optional: keys already in the cache before the segment's first query row (decoder.py:151-158,
                                           // a multi-token call on a non-empty cache); the segment then has seg_len + seg_k_off keys
    int group;                             // query heads per kv head
    float scale;
    int n_qt, n_heads;                     // query tiles per segment (grid), query heads: the 1-D grid is decoded in the kernel
    int xcd_order;
    unsigned long long* stamp; int stamp_wg;   // diagnostic (q3asr_attn_set_stamp): workgroup stamp_wg records wall_clock64() at its phase boundaries
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
    // Round 5: XCD-aware order (p.xcd_order; the launcher sets it for the bidirectional = encoder launches).  Workgroups are dealt
    // round-robin over the 8 XCDs, so with the plain (query tile, head, segment) grid the workgroups that read ONE kv head's K/V sit
    // on different XCDs and every private L2 fetches it again.  The 1-D grid is remapped so that an XCD gets a contiguous chunk of
    // the order (segment, kv head, query tile, query head of the group).  Same arithmetic per workgroup: same bits.  Measured
    // (tools/ab_attn.py, batch 32): encoder windows 35.7 -> 32.5 us per layer; the causal prefill 72.1 -> 73.9 us -- its K/V
    // loads are not what a key tile waits for (profiles/r05_attention.txt has the stamped timeline) -- so it keeps the grid order.
    int seg, head, qt;
    if (p.xcd_order) {
        const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
        const int per_kv = p.n_qt * p.group, per_seg = per_kv * (p.n_heads / p.group);
        seg = (int)(L / (unsigned)per_seg);
        const int r_seg = (int)(L % (unsigned)per_seg);
        const int kvh = r_seg / per_kv, w = r_seg % per_kv;
        head = kvh * p.group + w % p.group;
        qt = CAUSAL ? p.n_qt - 1 - w / p.group : w / p.group;          // causal: a query tile's work grows with its index -- heavy tiles first
    } else {                                                           // the order of rounds 1-4 (A/B: q3asr_set_option(Q3ASR_OPT_ATTN_XCD_ORDER, 0))
        const int b = (int)blockIdx.x;
        qt = CAUSAL ? p.n_qt - 1 - b % p.n_qt : b % p.n_qt;
        head = (b / p.n_qt) % p.n_heads;
        seg = b / (p.n_qt * p.n_heads);
    }
    const int len = p.seg_len[seg];
    if (qt * 64 >= len) return;
    const int koff = p.seg_k_off ? p.seg_k_off[seg] : 0;   // query row i sits at position koff + i and sees keys 0 .. koff + i
    const int klen = len + koff;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int n_stamp = 0;
    const bool stamping = p.stamp != nullptr && (int)blockIdx.x == p.stamp_wg;                // uniform
#define ATT_STAMP() do { if (stamping && tid == 0 && n_stamp < 63) p.stamp[1 + n_stamp++] = wall_clock64(); } while (0)
    ATT_STAMP();                                                                            // 0: entry
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
    const float c2 = p.scale * 1.4426950408889634f;      // scale * log2(e): the softmax exponent in base 2 (below)

    const int n_tiles = CAUSAL ? (min(klen, koff + qt * 64 + 64) + KV_TILE - 1) / KV_TILE : (klen + KV_TILE - 1) / KV_TILE;   // koff = 0: qt + 1
    // K and V tile: KV_TILE*CPR 16-byte chunks each, 256 threads; the next tile's chunks are loaded into registers while
    // the current tile computes (one HBM/L2 round trip per tile would otherwise sit between two barriers)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NCH = KV_TILE * CPR / 256;
    u32x4 kreg[NCH], vreg[NCH];
    // a key row's byte offset from the (uniform) tile base as ONE 32-bit value shared by the K and the V load (host-checked
    // to fit): ~3 VALU per row instead of the ~9 of two 64-bit address chains -- the loop is VALU-issue-bound once the LDS
    // round trips are grouped (round 4, tools/pmc_attn.sh)
    const unsigned row_bytes = (unsigned)p.kv_row_stride * 2u, ch_bytes = (unsigned)(tid % CPR) * 16u;
    const int row0 = tid / CPR;
    auto fetch = [&](int kt_) {
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int key = min(kt_ * KV_TILE + row0 + (256 / CPR) * it, klen - 1);
            const unsigned boff = (unsigned)key * row_bytes + ch_bytes;
            kreg[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Kb) + boff);
            vreg[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Vb) + boff);
        }
    };
    static_assert((256 / CPR) % 8 == 0, "staging rows of one thread must be a multiple of 8 apart");
    const int wbase = TileImg<HD>::off(row0, tid % CPR);
    // fragment read bases: the k-step / d-tile index enters the swizzled chunk by XOR (bits above the row term), so one base
    // each and an XOR per read replace 4 + 8 address registers
    const int kbase = TileImg<HD>::off(li, g);
    const int vbase = TileImg<HD>::off(4 * g + (li >> 2), (li & 3) >> 1) + 8 * (li & 1);
    fetch(0);
    // first key tile that can hold a key beyond the sequence or above the diagonal of one of this workgroup's queries: tile kt is
    // wholly visible iff 64 kt + 63 <= koff + 64 qt (the smallest query position), i.e. kt < qt + (koff + 1) / 64; with koff = 0
    // that leaves the last tile alone, with a cache offset the diagonal can straddle the last two
    int mask_from = CAUSAL ? qt + (koff + 1) / 64 : n_tiles - 1;
    asm volatile("" : "+s"(mask_from));                  // opaque: hipcc otherwise peels the masked tile out of the loop and spills at the join
    for (int kt = 0; kt < n_tiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NCH; ++it) {                  // rows row0 + (256 / CPR) it: a multiple of 8 apart, one base (ROWB note)
            *reinterpret_cast<u32x4*>(sK + wbase + it * (256 / CPR) * TileImg<HD>::ROWB) = kreg[it];
            *reinterpret_cast<u32x4*>(sV + wbase + it * (256 / CPR) * TileImg<HD>::ROWB) = vreg[it];
        }
        if (kt + 1 < n_tiles) fetch(kt + 1);
        __syncthreads();
        ATT_STAMP();                                         // 1 + 4 kt: tile staged (waited for its loads, written to LDS, both barriers)

        // ---- S^T = K Q^T.  Round 4: k-step outer, the four key tiles' fragments read as one group in front of their four
        // MFMAs.  Written key-tile outer, hipcc (at the 170-VGPR cap of three workgroups per CU) emitted ds_read ->
        // s_waitcnt lgkmcnt(0) -> MFMA sixteen times in a row, the four MFMAs of a key tile back to back on ONE accumulator:
        // 32 exposed LDS round trips and 12 dependent-MFMA stalls per key tile and wave (with PV below), which is what the
        // stamps of round 3 called "latency-bound".  Each accumulator still sees its k-steps in ascending order: same bits.
        f32x4 s[4];                                          // first k-step: accumulator operand = the inline constant 0 (no 16 v_mov per tile)
        bf16x8 kf[2][4];                                     // the next k-step's fragments are requested before this one's MFMAs
        auto read_k = [&](int ks, bf16x8 (&f)[4]) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)                   // off(16 k4 + li, 4 ks + g) = (off(li, g) ^ (ks << 6)) + 16 k4 ROWB
                f[k4] = *reinterpret_cast<const bf16x8*>(sK + ((kbase ^ (ks << 6)) + 16 * k4 * TileImg<HD>::ROWB));
        };
        read_k(0, kf[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) read_k(ks + 1, kf[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                s[k4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks & 1][k4], qf[ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : s[k4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- mask + online softmax (row = query li, spread over the 4 lane groups)
        // Only the last key tile(s) of a query tile can hold keys beyond the sequence or above the diagonal (mask_from above):
        // the mask runs there alone (uniform branch).
        // Round 5: the softmax runs on the RAW scores in the base-2 domain: exp(scale (s - m)) = 2^(c s - c m), c = scale log2(e)
        // (scale > 0, so the maximum of the raw scores is the maximum of the scaled ones).  One v_fma + one v_exp per score in
        // place of v_mul (scale), v_sub, v_mul (log2 e inside __expf), v_exp: 33 fewer VALU instructions of the ~187 a wave issues per
        // key tile, and the loop is VALU-ISSUE-bound -- a wave64 VALU instruction takes ~4.3 cycles on this part (r04 PMC), so a key
        // tile's 170 + 17 transcendental instructions are ~870 cycles of issue against 512 cycles of MFMA (profiles/r05_attention.txt).
        ATT_STAMP();                                         // 2 + 4 kt: S = K Q^T done
        // The whole softmax is written once and instantiated in both branches: the scores are only READ after the MFMAs (with the
        // mask applied in place, the join of the two branches cost the common path 16 register copies per tile).
        float alpha;
        bf16x8 pf[2];
        auto softmax = [&](auto masked) {
            float tmax = -INFINITY;
            float sv[4][4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[k4][r];
                    if constexpr (decltype(masked)::value) {
                        const int key = kt * KV_TILE + 16 * k4 + 4 * g + r;
                        v = (key < klen && (!CAUSAL || key <= q_eff + koff)) ? v : -INFINITY;
                    }
                    sv[k4][r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = lane_xor16_max(tmax);                 // = fmaxf(tmax, tmax[lane ^ 16]) without the LDS crossbar (common.h)
            tmax = lane_xor32_max(tmax);
            const float m_new = fmaxf(m_run, tmax);      // raw-score units; finite: key 0 is visible to every query
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);               // m_run = -inf on the first tile -> 0
            const float mc = -m_new * c2;
            float psum = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sv[k4][r], c2, mc));   // masked scores are -inf: 2^-inf = 0
                    psum += pv;
                    pf[k4 >> 1][(k4 & 1) * 4 + r] = f2bf(pv);
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
        };
        if (kt >= mask_from) softmax(std::true_type{});
        else softmax(std::false_type{});

        ATT_STAMP();                                         // 3 + 4 kt: softmax done
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
        // ---- O^T += V^T P^T
        // the transposing V reads of four d tiles as one group in front of their four MFMAs (independent accumulators), as above
        // group q = (key half c, four d tiles d0): rows 32 c + 4 g + (li >> 2) and + 16 (immediate row steps, ROWB note above),
        // chunk 2 dt + ((li & 3) >> 1): the d tile enters the swizzled chunk by XOR above bit 4; the next group's reads are
        // requested before this group's MFMAs
        union VFrag { struct { s16x4 a, b; } h; bf16x8 v; };
        constexpr int NG = 2 * (DT / 4);
        VFrag u[2][4];
        auto read_v = [&](int q, VFrag (&f)[4]) {
            const int c = q / (DT / 4), d0 = 4 * (q % (DT / 4));
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int vb = vbase ^ ((d0 + dd) << 5);
                f[dd].h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sV + vb + (32 * c) * TileImg<HD>::ROWB));
                f[dd].h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(sV + vb + (32 * c + 16) * TileImg<HD>::ROWB));
            }
        };
        read_v(0, u[0]);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if (q + 1 < NG) read_v(q + 1, u[(q + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int c = q / (DT / 4), d0 = 4 * (q % (DT / 4));
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) o[d0 + dd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u[q & 1][dd].v, pf[c], o[d0 + dd], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ATT_STAMP();                                         // 4 + 4 kt: O += P V issued
    }
    ATT_STAMP();                                             // loop done (the last tile's P V included)
    if (stamping && tid == 0) p.stamp[0] = (unsigned long long)n_stamp | ((unsigned long long)qt << 32);
    float l_tot = lane_xor16_sum(l_run);
    l_tot = lane_xor32_sum(l_tot);
    // the output addresses are rebuilt here from an opaque copy of the thread id: hoisted above the key loop they stayed live
    // across it, and at the 168-register cap of three workgroups per CU that cost spills
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int q_abs2 = qt * 64 + (tid2 >> 6) * 16 + (tid2 & 15), g2 = (tid2 & 63) >> 4;
    if (q_abs2 < len) {
        const float inv = 1.0f / l_tot;
        bf16* orow = p.O + ((long)p.seg_q_start[seg] + q_abs2) * p.o_stride + head * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[dt][r] * inv);
            *reinterpret_cast<bf16x4*>(orow + 16 * dt + 4 * g2) = ov;
        }
    }
}

static unsigned long long* g_attn_stamp = nullptr;
static int g_attn_stamp_wg = 0;
// Diagnostic: workgroup `wg` (blockIdx.x) of the following attn_fwd launches writes wall_clock64() stamps (100 MHz) to buf (64 x u64 device
// memory): [0] = count | query tile << 32, then entry and per key tile "staged / S done / softmax done / PV issued", then "loop done".
extern "C" int q3asr_attn_set_stamp(void* buf, int wg) { g_attn_stamp = (unsigned long long*)buf; g_attn_stamp_wg = wg; return 0; }

extern "C" int q3asr_attn_fwd(const void* Q, int64_t q_stride, const void* K, const void* V, const int64_t* kv_seg_off,
                              int64_t kv_head_stride, int64_t kv_row_stride, void* O, int64_t o_stride,
                              const int32_t* seg_q_start, const int32_t* seg_len, const int32_t* seg_k_off, int n_segs, int max_len,
                              int max_klen, int n_heads, int n_kv_heads, int head_dim, int causal, float scale, void* stream) {
    Q3_REQUIRE(Q && K && V && kv_seg_off && O && seg_q_start && seg_len, "attn_fwd: null pointer");
    Q3_REQUIRE(n_segs > 0 && max_len > 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0, "attn_fwd: bad sizes");
    Q3_REQUIRE(q_stride % 8 == 0 && kv_row_stride % 8 == 0 && kv_head_stride % 8 == 0 && o_stride % 4 == 0, "attn_fwd: strides must keep 16-byte alignment");
    Q3_REQUIRE(max_klen >= max_len && (seg_k_off || max_klen == max_len), "attn_fwd: max_klen=%d (keys) must be >= max_len=%d (query rows), equal without seg_k_off", max_klen, max_len);
    Q3_REQUIRE(kv_row_stride > 0 && (int64_t)max_klen * kv_row_stride * 2 < (1LL << 32), "attn_fwd: a segment's K/V rows must lie within 4 GiB of its first (max_klen=%d, kv_row_stride=%ld)", max_klen, (long)kv_row_stride);
    AttnParams p;
    p.Q = (const bf16*)Q; p.q_stride = q_stride; p.K = (const bf16*)K; p.V = (const bf16*)V;
    p.kv_seg_off = (const long*)kv_seg_off; p.kv_head_stride = kv_head_stride; p.kv_row_stride = kv_row_stride;
    p.O = (bf16*)O; p.o_stride = o_stride; p.seg_q_start = seg_q_start; p.seg_len = seg_len; p.seg_k_off = seg_k_off;
    p.group = n_heads / n_kv_heads; p.scale = scale;
    // XCD-aware order for the encoder's windows (32.5 vs 35.7 us per layer at batch 32); the causal prefill keeps the grid order of
    // rounds 1-4, where the same remap measured 73.9 vs 72.1 us (tools/ab_attn.py, profiles/r05_attention.txt)
    p.n_qt = (max_len + 63) / 64; p.n_heads = n_heads; p.xcd_order = g_q3_option[Q3ASR_OPT_ATTN_XCD_ORDER] && !causal;
    p.stamp = g_attn_stamp; p.stamp_wg = g_attn_stamp_wg;
    Q3_REQUIRE((int64_t)p.n_qt * n_heads * n_segs < (1LL << 31), "attn_fwd: grid of %d x %d x %d workgroups", p.n_qt, n_heads, n_segs);
    dim3 grid((unsigned)(p.n_qt * n_heads * n_segs)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, blk, 0, s, p);
    else if (head_dim == 128 && causal) hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, blk, 0, s, p);
    else if (head_dim == 128 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, blk, 0, s, p);
    else if (head_dim == 64 && causal) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, blk, 0, s, p);
    else { q3asr_set_error("attn_fwd: head_dim=%d unsupported (64 or 128)", head_dim); return Q3_EBADARG; }
    return q3_check_launch("attn_fwd");
}
