// Token selection for a batch of sequences, one 1024-thread workgroup per sequence
// (reference generate.py:88-215 `sample` + the bookkeeping of the loop at :278-313).
//
//   repetition penalty  (generate.py:88-131): unique ids among the last `ctx` GENERATED tokens get
//                       l>0 ? l/p : l*p, with p and the result rounded to bf16 (weakly-typed scalar
//                       against bf16 logits).  The logits row is never written: penalised ids are
//                       marked in an LDS bitmap (one byte per 16-byte load of 8 logits) and their new
//                       values kept in an LDS list, so every scan stays a pure streaming read.
//   greedy              (generate.py:203-204): argmax, lowest index wins ties (mx.argmax).  One pass.
//   temperature/top-k/top-p/categorical (generate.py:206-215,134-163):
//       pass 1  histogram of the row over the top 12 bits of the order-preserving 16-bit bf16 key
//               (4096 bins in LDS, one u64 atomic per logit: count << 44 | fixed-point mass relative
//               to the bin's largest value, so the mass of a bin needs no global max and every sum
//               is an integer -> the draw is reproducible for a seed);
//       scans   block prefix scans over the 4096 bins give the top-k bin, the nucleus bin and the
//               bin the draw lands in;
//       pass 2  exact per-key counts of a 128-bin window (2048 keys) that resolves those bins to
//               exact keys: exact top-k threshold (ties kept), nucleus = sort + exclusive cumsum with
//               ties ordered by index; repeated only if a later bin falls outside the window;
//       pass 3  the r-th logit (index order) carrying the drawn key: per-wave segment counts, one
//               block scan, one 1 KiB re-read by the owning wave.
//       The draw walks the kept set in descending key order (inverse CDF, Philox4x32-10 uniform).
//       Filter maths in fp32 (the reference's bf16 cumsum saturation is not reproduced:
//       distributional parity only).
//   bookkeeping         append to out_tokens unless finished, EOS / budget -> finished, advance pos.
// Round 5: rows of up to 155,648 logits (the 151,936 of Qwen3-ASR) stay ON CHIP across the passes: thread t owns the 16-byte
// chunks t, t + 1024, ... (19 of them), all requested before the first is consumed (one memory round trip instead of five
// groups of four loads in flight), and held after the penalty and the temperature have been applied once -- 16 chunks in
// registers (64 of the 128 VGPRs a 1024-thread workgroup may use), 3 in LDS -- so the window and locate passes read no memory.
// Longer rows stream as before (q3asr_set_option(Q3ASR_OPT_SAMPLER_RESIDENT, 0) forces that form: same tokens).
#include "common.h"
#include <atomic>

#define SMP_THREADS 1024
#define SMP_WAVES (SMP_THREADS / 64)
#define SMP_MAXCTX 1024
#define SMP_BINS 4096                 // key >> 4
#define SMP_WIN_BINS 128              // bins resolved to exact keys per window pass
#define SMP_WIN_KEYS (SMP_WIN_BINS * 16)
#define SMP_CNT_SHIFT 44              // hist word = count << 44 | sum of round(2^24 * exp(v - top(bin)))
#define SMP_MASS_ONE 16777216.0f
#define SMP_RES 19                    // resident 16-byte chunks per thread ...
#define SMP_RES_LDS_BYTES ((SMP_RES - 16) * 16 * SMP_THREADS)
#define SMP_RES_REG 16                // ... of which in registers across the passes; the other 3 sit in LDS (48 KiB), see resident_pass

struct SampleParams {
    const bf16* logits; long ld; int V;
    int* out_tokens; int max_out;
    int* n_gen; int* finished; int* cur_tok; int* pos; const int* max_new;
    float temperature, top_p, penalty; int top_k; int ctx;
    unsigned long long seed;
    int eos0, eos1;
    int row_base;             // global index of row 0 in its request: the draw of a row does not depend on how the batch was cut
};

__device__ __forceinline__ unsigned key16(float x) {          // order-preserving 16-bit key of a bf16-valued float
    const unsigned u = __float_as_uint(x) >> 16;
    return (u & 0x8000u) ? (~u & 0xFFFFu) : (u | 0x8000u);
}
__device__ __forceinline__ float key16_to_float(unsigned k) {
    const unsigned u = (k & 0x8000u) ? (k & 0x7FFFu) : (~k & 0xFFFFu);
    return __uint_as_float(u << 16);
}

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned a, unsigned b) {
    unsigned c[4] = {a, b, 0u, 0u};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

#ifdef SMP_TIMING
__device__ long long g_smp_t[32];
#define SMP_T(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_smp_t[i] = wall_clock64(); } while (0)
#else
#define SMP_T(i)
#endif

#define SMP_HASH 2048                  // open-addressing table of penalised ids (at most SMP_MAXCTX entries: load <= 1/2)

struct SmpShared {
    int hkey[SMP_HASH];                 // penalised token id, -1 = empty
    float hval[SMP_HASH];               // its logit after the penalty
    int scratch[SMP_THREADS];           // per-(wave, iteration) match counts of the locate pass
    float red_f[SMP_WAVES];
    int red_i[SMP_WAVES];
    float bc_f[8];
    int bc_i[8];
};

__device__ __forceinline__ unsigned smp_hash(int id) { return ((unsigned)id * 2654435761u) >> 21; }          // -> [0, 2048)

// penalty-adjusted, optionally temperature-scaled values of the 8 logits of chunk c
__device__ __forceinline__ void decode8(const bf16x8 raw, unsigned bits, int c, const SmpShared& sh, bool scaled,
                                        float t_bf, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(raw[e]);
    if (bits) {
        for (int e = 0; e < 8; ++e)
            if (bits & (1u << e)) {
                const int id = 8 * c + e;
                unsigned slot = smp_hash(id);
                for (int probe = 0; probe < SMP_HASH && sh.hkey[slot] != id; ++probe) slot = (slot + 1) & (SMP_HASH - 1);
                v[e] = sh.hval[slot];
            }
    }
    if (scaled) {
        // logits / temperature in bf16 (generate.py:206).  v * (1/t) is within 2 ulp of the quotient; the bf16 rounding
        // of the two can only differ next to a rounding midpoint, where the exact division is taken instead.
        const float rt = 1.0f / t_bf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float q = v[e] * rt;
            const unsigned u = __float_as_uint(q);
            if (__builtin_expect(((u & 0xFFFFu) - 0x7FF0u) <= 0x20u || (u & 0x7F800000u) == 0u, 0)) q = v[e] / t_bf;
            v[e] = round_bf(q);
        }
    }
}

// One coalesced pass over the row, 4 x 16-byte loads in flight per thread.  Thread t sees the chunks t + 1024 j in ascending j;
// f(j, chunk, values[8], valid) is called by EVERY thread for every j < ceil(nchunks / 1024) (wave-uniform control flow: f may
// reduce across the wave); `valid` is false past the end of the row, where the values are those of the last chunk.
template <typename F>
__device__ __forceinline__ void stream_row(const bf16* row, int nchunks, const unsigned char* bitmap, const SmpShared& sh,
                                           bool scaled, float t_bf, F&& f) {
    const int iters = (nchunks + SMP_THREADS - 1) / SMP_THREADS;
    for (int j0 = 0; j0 < iters; j0 += 4) {
        bf16x8 raw[4];
        unsigned bits[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min((j0 + u) * SMP_THREADS + (int)threadIdx.x, nchunks - 1);
            raw[u] = *reinterpret_cast<const bf16x8*>(row + 8 * c);
            bits[u] = bitmap[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u >= iters) break;
            const int c = (j0 + u) * SMP_THREADS + (int)threadIdx.x;
            float v[8];
            decode8(raw[u], bits[u], min(c, nchunks - 1), sh, scaled, t_bf, v);
            f(j0 + u, c, v, c < nchunks);
        }
    }
}

// The resident form.  The row lives in 19 x 4 registers per thread as raw 32-bit words (two bf16 each).
typedef unsigned smp_u32x4 __attribute__((ext_vector_type(4)));
// request the thread's SMP_RES chunks (all in flight at once) ...
__device__ __forceinline__ void resident_load(const bf16* row, int nchunks, smp_u32x4 (&rr)[SMP_RES]) {
#pragma unroll
    for (int j = 0; j < SMP_RES; ++j) {
        const int c = min(j * SMP_THREADS + (int)threadIdx.x, nchunks - 1);
        rr[j] = *reinterpret_cast<const smp_u32x4*>(row + 8 * c);
    }
}
__device__ __forceinline__ void unpack8(const smp_u32x4 w, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
}
// ... first pass: apply the penalty and the temperature once, keep the results and hand them to f ...
template <typename F>
__device__ __forceinline__ void resident_first_pass(int nchunks, const unsigned char* bitmap, const SmpShared& sh, bool scaled, float t_bf,
                                                    smp_u32x4 (&rr)[SMP_RES], smp_u32x4* lrow, F&& f) {
    const int iters = (nchunks + SMP_THREADS - 1) / SMP_THREADS;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                                       // the 19 clamped chunk indices of resident_load are not kept alive for this loop
#pragma unroll
    for (int j = 0; j < SMP_RES; ++j) {
        if (j < iters) {                                                 // uniform; no break: the loop must unroll (static register indices)
            const int c = j * SMP_THREADS + tid_, cc = min(c, nchunks - 1);
            float v[8];
            decode8(__builtin_bit_cast(bf16x8, rr[j]), bitmap[cc], cc, sh, scaled, t_bf, v);
            smp_u32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i)                                  // exact: v holds bf16 values
                w[i] = (__float_as_uint(v[2 * i]) >> 16) | (__float_as_uint(v[2 * i + 1]) & 0xFFFF0000u);
            if (j < SMP_RES_REG) rr[j] = w;
            else lrow[(j - SMP_RES_REG) * SMP_THREADS + tid_] = w;       // the thread's own slot: no barrier between its write and its reads
            f(j, c, v, c < nchunks);
        }
        __builtin_amdgcn_sched_barrier(0);                               // one chunk's temporaries at a time
    }
}
// ... and every later pass is a walk over registers (chunks 0-15) and LDS (16-18: the phase logic between the passes keeps ~60
// registers alive, and 76 + 60 > 128), with stream_row's calling convention.  The empty asm makes a chunk's registers opaque
// at each use: the passes run inside the phase loop, where hipcc otherwise hoists the 152 unpacked values (and their keys) of the
// loop-invariant row out of the loop and spills them (3 KiB of scratch per lane).
// (What remains in scratch: 18 loop-invariant dwords of the phase logic -- shuffle addresses, 1 / temperature, wave id -- stored once
// and re-read between the passes, none of them inside a pass.)
template <typename F>
__device__ __forceinline__ void resident_pass(int nchunks, smp_u32x4 (&rr)[SMP_RES], const smp_u32x4* lrow, F&& f) {
    const int iters = (nchunks + SMP_THREADS - 1) / SMP_THREADS;
#pragma unroll
    for (int j = 0; j < SMP_RES; ++j) {
        if (j < iters) {
            const int c = j * SMP_THREADS + (int)threadIdx.x;
            smp_u32x4 w;
            if (j < SMP_RES_REG) { asm volatile("" : "+v"(rr[j])); w = rr[j]; }
            else w = lrow[(j - SMP_RES_REG) * SMP_THREADS + threadIdx.x];
            float v[8];
            unpack8(w, v);
            f(j, c, v, c < nchunks);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Repetition penalty set-up: bitmap of penalised ids + hash table id -> new value in LDS.  Ends with a barrier.
__device__ __forceinline__ void penalty_setup(const SampleParams& p, const bf16* row, int b, int n, unsigned char* bitmap,
                                              SmpShared& sh) {
    const int tid = threadIdx.x, nchunks = p.V / 8;
    for (int i = tid; i < (nchunks + 3) / 4; i += SMP_THREADS) reinterpret_cast<unsigned*>(bitmap)[i] = 0u;
    for (int i = tid; i < SMP_HASH; i += SMP_THREADS) sh.hkey[i] = -1;
    __syncthreads();
    // over the last min(n, ctx) generated tokens (first token: none, generate.py:284); an id listed twice is penalised once
    const int cnt = (p.penalty != 1.0f && p.ctx > 0) ? min(min(n, p.ctx), SMP_MAXCTX) : 0;
    if (tid < cnt) {
        const int id = p.out_tokens[(long)b * p.max_out + n - cnt + tid];
        if (id >= 0 && id < p.V) {
            const float l = bf2f(row[id]);
            const float pb = round_bf(p.penalty);
            const float nv = round_bf(l > 0.f ? l / pb : l * pb);
            unsigned slot = smp_hash(id);
            for (int probe = 0; probe < SMP_HASH; ++probe) {
                const int prev = atomicCAS(&sh.hkey[slot], -1, id);
                if (prev == -1 || prev == id) { sh.hval[slot] = nv; break; }     // duplicates store the same value
                slot = (slot + 1) & (SMP_HASH - 1);
            }
            atomicOr(reinterpret_cast<unsigned*>(bitmap) + (id >> 5), 1u << (id & 31));
        }
    }
    __syncthreads();
}

// bookkeeping (generate.py:286-313): EOS is recorded, then the sequence stops
__device__ __forceinline__ void commit_token(const SampleParams& p, int b, int n, int token) {
    // A row of NaN logits (a diverged model, a corrupt checkpoint, a kernel fault upstream) has no maximum and no sampled
    // index.  The sentinel must not reach the embedding gather of the next step as a row number: the row records token 0,
    // stops, and its `finished` flag becomes Q3ASR_ROW_FAILED (2) instead of 1 -- the device-side error flag the host looks
    // at whenever it reads `finished` (generate.py raises Q3AsrError naming the row; include/q3asr.h).
    const bool bad = token < 0 || token >= p.V;
    if (bad) token = 0;
    if (!p.finished[b]) {
        if (n < p.max_out) p.out_tokens[(long)b * p.max_out + n] = token;
        const int n1 = n + 1;
        p.n_gen[b] = n1;
        p.cur_tok[b] = token;
        if (bad) p.finished[b] = 2;
        else if (token == p.eos0 || token == p.eos1 || n1 >= p.max_new[b] || n1 >= p.max_out) p.finished[b] = 1;
        else p.pos[b] = p.pos[b] + 1;
    }
}

// ------------------------------------------------------------------------------------------- greedy
// (The resident form of the draw kernel below does not pay here: one pass, nothing to re-read, and 88 instead of 40 VGPRs --
// 15.8 vs 14.8 us per launch at 40 rows, tools/ab_sampler.py.)
__global__ __launch_bounds__(SMP_THREADS) void sample_greedy_kernel(const SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ SmpShared sh;
    unsigned char* bitmap = dyn;                                        // V/8 bytes
    const int b = blockIdx.x, tid = threadIdx.x;
    const bf16* row = p.logits + (long)b * p.ld;
    const int nchunks = p.V / 8;
    const int n = p.n_gen[b];
    penalty_setup(p, row, b, n, bitmap, sh);

    float best = -INFINITY;
    int best_i = 0x7fffffff;
    stream_row(row, nchunks, bitmap, sh, false, 1.0f, [&](int, int c, const float (&v)[8], bool valid) {
        if (!valid) return;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (v[e] > best) { best = v[e]; best_i = 8 * c + e; }
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    if ((tid & 63) == 0) { sh.red_f[tid >> 6] = best; sh.red_i[tid >> 6] = best_i; }
    __syncthreads();
    if (tid == 0) {
        float bv = sh.red_f[0];
        int bi = sh.red_i[0];
        for (int i = 1; i < SMP_WAVES; ++i)
            if (sh.red_f[i] > bv || (sh.red_f[i] == bv && sh.red_i[i] < bi)) { bv = sh.red_f[i]; bi = sh.red_i[i]; }
        commit_token(p, b, n, bi);
    }
}

// ----------------------------------------------------------------------------------------- sampling
template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(v, o, 64);
        if ((int)(threadIdx.x & 63) >= o) v += t;
    }
    return v;
}
// exclusive prefix of v over the workgroup in thread order and the workgroup total (fixed summation tree: deterministic)
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T* red, T& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T incl = wave_incl_scan(v);
    T excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = (T)0;
    __syncthreads();                                                    // red may still be read by a previous scan
    if (lane == 63) red[w] = incl;
    __syncthreads();
    T pre = (T)0, tot = (T)0;
    for (int i = 0; i < SMP_WAVES; ++i) {
        const T x = red[i];
        if (i < w) pre += x;
        tot += x;
    }
    total = tot;
    return pre + excl;
}

// Thread t owns the 4 bins d = 4t..4t+3 in DESCENDING key order (bin = 4095 - d) with weights m[]; `pre` is the
// running sum of the weights of all bins before the thread's.  Zeroing or shrinking weights BELOW a bin leaves the
// prefixes at and above it valid, so one scan serves every search.
// Finds the first d with m > 0 whose inclusive running sum reaches `target` (strict: exceeds it); if rounding leaves
// none, the last d with m > 0.  Returns d (or -1 when every weight is zero) and the exclusive running sum there.
template <typename T>
__device__ __forceinline__ int find_crossing(const T (&m)[4], T pre, T target, bool strict, SmpShared& sh, T& excl_out) {
    const int tid = threadIdx.x;
    if (tid == 0) { sh.bc_i[4] = 0x7fffffff; sh.bc_i[5] = -1; }
    __syncthreads();
    T run = pre;
    int cross = 0x7fffffff, last = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (m[i] > (T)0) {
            const T incl = run + m[i];
            const bool hit = strict ? incl > target : incl >= target;
            if (hit && cross == 0x7fffffff) cross = 4 * tid + i;
            last = 4 * tid + i;
            run = incl;
        }
    }
    if (cross != 0x7fffffff) atomicMin(&sh.bc_i[4], cross);
    if (last >= 0) atomicMax(&sh.bc_i[5], last);
    __syncthreads();
    const int d = sh.bc_i[4] != 0x7fffffff ? sh.bc_i[4] : sh.bc_i[5];
    if (d >= 0 && (d >> 2) == tid) {
        T e = pre;
        for (int i = 0; i < (d & 3); ++i) e += m[i];
        *reinterpret_cast<T*>(&sh.bc_f[4]) = e;
    }
    __syncthreads();
    excl_out = d >= 0 ? *reinterpret_cast<const T*>(&sh.bc_f[4]) : (T)0;
    __syncthreads();
    return d;
}
// running sum of the weights before descending position d (owner thread broadcasts)
__device__ __forceinline__ float mass_before(const float (&m)[4], float pre, int d, SmpShared& sh) {
    if ((d >> 2) == (int)threadIdx.x) {
        float e = pre;
        for (int i = 0; i < (d & 3); ++i) e += m[i];
        sh.bc_f[5] = e;
    }
    __syncthreads();
    const float e = sh.bc_f[5];
    __syncthreads();
    return e;
}

template <bool RES>
__global__ __launch_bounds__(SMP_THREADS) void sample_draw_kernel(const SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ SmpShared sh;
    unsigned long long* hist = reinterpret_cast<unsigned long long*>(dyn);               // [4096]
    unsigned* fine = reinterpret_cast<unsigned*>(dyn + SMP_BINS * 8);                    // [2][2048]
    smp_u32x4* lrow = reinterpret_cast<smp_u32x4*>(dyn + SMP_BINS * 8 + 2 * SMP_WIN_KEYS * 4);    // RES: [3][1024] chunks
    unsigned char* bitmap = dyn + SMP_BINS * 8 + 2 * SMP_WIN_KEYS * 4 + (RES ? SMP_RES_LDS_BYTES : 0);   // V/8 bytes
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bf16* row = p.logits + (long)b * p.ld;
    const int nchunks = p.V / 8;
    const int n = p.n_gen[b];
    const float t_bf = round_bf(p.temperature);
    SMP_T(0);
    smp_u32x4 rr[SMP_RES];
    if constexpr (RES) resident_load(row, nchunks, rr);                 // in flight during the penalty set-up
    penalty_setup(p, row, b, n, bitmap, sh);
    // one pass over the row's (penalised, scaled) values: registers (after the first pass has put them there) or a streaming re-read
    auto each = [&](auto&& f) {
        if constexpr (RES) resident_pass(nchunks, rr, lrow, f);
        else stream_row(row, nchunks, bitmap, sh, true, t_bf, f);
    };
    SMP_T(1);

    // ---- pass 1: 4096-bin histogram (count + mass relative to each bin's largest value)
    for (int i = tid; i < SMP_BINS; i += SMP_THREADS) hist[i] = 0ull;
    __syncthreads();
    auto to_hist = [&](int, int, const float (&v)[8], bool valid) {
        if (!valid) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned k = key16(v[e]);
            const float top = key16_to_float(k | 15u);
            const float rel = v[e] == top ? 1.0f : __expf(v[e] - top);                   // (-inf) - (-inf) stays out of exp
            atomicAdd(&hist[k >> 4], (1ull << SMP_CNT_SHIFT) | (unsigned long long)(unsigned)(rel * SMP_MASS_ONE + 0.5f));
        }
    };
    if constexpr (RES) resident_first_pass(nchunks, bitmap, sh, true, t_bf, rr, lrow, to_hist);
    else stream_row(row, nchunks, bitmap, sh, true, t_bf, to_hist);
    __syncthreads();

    SMP_T(2);
    // ---- per-thread view of its 4 bins, descending
    int cnt[4];
    float m[4];                                                         // kept mass of the bin relative to exp(ref)
    int top_bin = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int bin = SMP_BINS - 1 - (4 * tid + i);
        cnt[i] = (int)(hist[bin] >> SMP_CNT_SHIFT);
        if (cnt[i] > 0) top_bin = max(top_bin, bin);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) top_bin = max(top_bin, __shfl_xor(top_bin, o, 64));
    if (lane == 0) sh.red_i[wv] = top_bin;
    __syncthreads();
    for (int i = 0; i < SMP_WAVES; ++i) top_bin = max(top_bin, sh.red_i[i]);
    const float ref = key16_to_float(((unsigned)top_bin << 4) | 15u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int bin = SMP_BINS - 1 - (4 * tid + i);
        const float fx = (float)(hist[bin] & ((1ull << SMP_CNT_SHIFT) - 1)) * (1.0f / SMP_MASS_ONE);
        const float w = __expf(key16_to_float(((unsigned)bin << 4) | 15u) - ref);
        m[i] = cnt[i] > 0 ? fx * w : 0.f;
        if (!(m[i] >= 0.f)) m[i] = 0.f;                                  // NaN logits carry no mass
    }

    auto key_mass = [&](unsigned key) { return __expf(key16_to_float(key) - ref); };

    SMP_T(3);
    // One scan of the bin masses (and of the counts for top-k) in descending key order.
    float total_m;
    const float pre_m = block_excl_scan(m[0] + m[1] + m[2] + m[3], sh.red_f, total_m);
    const bool use_k = p.top_k > 0 && p.top_k < p.V, use_p = p.top_p < 1.0f;
    int total_c = 0;
    const int pre_c = use_k ? block_excl_scan(cnt[0] + cnt[1] + cnt[2] + cnt[3], sh.red_i, total_c) : 0;
    const float u01 = philox_uniform(p.seed, (unsigned)n, (unsigned)(b + p.row_base));

    // Three resolutions of a coarse bin to exact keys, in order: 0 = top-k threshold (generate.py:134-150, ties kept),
    // 1 = nucleus crossing (generate.py:152-163: key fk, of which the first keep_eq logits in index order stay),
    // 2 = the draw (inverse CDF over the kept set in descending key order).  fine[] holds exact key counts of two
    // 128-bin windows [wbA, wbA+128) and [wbB, wbB+128), rebuilt by a streaming pass only when a phase needs a bin
    // outside them; the nucleus phase also asks for the bin the draw will most likely land in (kept mass ~ top_p * Z).
    unsigned thr_k = 0, fk = 0, key_d = 0;
    int bin_k = -1, keep_eq = 0x7fffffff, wbA = -1, wbB = -1, r = 0;
    float kept = total_m;                                                // mass of the kept set so far
    for (int phase = use_k ? 0 : (use_p ? 1 : 2); phase < 3; phase += (phase == 0 && !use_p) ? 2 : 1) {
        int bin, bin2 = -1, before_i = 0;
        float before = 0.f, tgt = 0.f;
        if (phase == 0) {
            const int d = find_crossing(cnt, pre_c, p.top_k, false, sh, before_i);
            bin = SMP_BINS - 1 - d;
            before = mass_before(m, pre_m, d, sh);
        } else {
            tgt = phase == 1 ? p.top_p * kept : u01 * kept;
            const int d = find_crossing(m, pre_m, tgt, phase == 2, sh, before);
            bin = d >= 0 ? SMP_BINS - 1 - d : max(top_bin, 0);
            if (phase == 1) {
                float unused;
                const int d2 = find_crossing(m, pre_m, u01 * tgt, true, sh, unused);
                bin2 = d2 >= 0 ? SMP_BINS - 1 - d2 : bin;
            }
        }
        SMP_T(4 + 4 * phase);
        const bool inA = wbA >= 0 && bin >= wbA && bin < wbA + SMP_WIN_BINS;
        const bool inB = wbB >= 0 && bin >= wbB && bin < wbB + SMP_WIN_BINS;
        if (!inA && !inB) {                                              // uniform across the workgroup
            wbA = bin;
            wbB = (bin2 >= 0 && (bin2 < wbA || bin2 >= wbA + SMP_WIN_BINS)) ? bin2 : -1;
            for (int i = tid; i < 2 * SMP_WIN_KEYS; i += SMP_THREADS) fine[i] = 0u;
            __syncthreads();
            const unsigned baseA = (unsigned)wbA << 4, baseB = wbB >= 0 ? (unsigned)wbB << 4 : 0x10000u;
            each([&](int, int, const float (&v)[8], bool valid) {
                if (!valid) return;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned k = key16(v[e]);
                    if (k - baseA < (unsigned)SMP_WIN_KEYS) atomicAdd(&fine[k - baseA], 1u);
                    else if (k - baseB < (unsigned)SMP_WIN_KEYS) atomicAdd(&fine[SMP_WIN_KEYS + k - baseB], 1u);
                }
            });
            __syncthreads();
        }
        const int wb = (wbA >= 0 && bin >= wbA && bin < wbA + SMP_WIN_BINS) ? wbA : wbB;
        const unsigned* fine_w = fine + (wb == wbA ? 0 : SMP_WIN_KEYS);
        if (tid == 0) {
            const unsigned* f = fine_w + 16 * (bin - wb);
            const unsigned kb = (unsigned)bin << 4;
            if (phase == 0) {
                int need = p.top_k - before_i, j = 15;
                for (; j > 0; --j) {
                    if ((int)f[j] >= need) break;
                    need -= (int)f[j];
                }
                float part = 0.f;                                        // mass of the kept keys of this bin
                for (int jj = 15; jj >= j; --jj) part += (float)f[jj] * key_mass(kb | jj);
                sh.bc_i[0] = (int)(kb | j);
                sh.bc_f[0] = part;
            } else if (phase == 1) {
                const int j_lo = bin == bin_k ? (int)(thr_k & 15u) : 0;
                int j = 15, last_j = -1;
                float a = before, pm = 0.f;
                bool crossed = false;
                for (; j >= j_lo; --j) {
                    if (f[j] == 0u) continue;
                    pm = key_mass(kb | j);
                    last_j = j;
                    if (a + pm * (float)f[j] >= tgt) { crossed = true; break; }
                    a += pm * (float)f[j];
                }
                int keep;
                if (crossed) {                                           // logit i (0-based among equals) survives iff a + i*pm < target
                    keep = (int)ceilf((tgt - a) / pm);
                    keep = max(1, min(keep, (int)f[j]));
                } else {                                                 // fixed-point bin mass vs exact key masses: keep the whole bin
                    j = max(last_j, j_lo);
                    keep = (int)f[j];
                    pm = key_mass(kb | j);
                    a -= pm * (float)keep;
                }
                sh.bc_i[0] = (int)(kb | j);
                sh.bc_i[1] = keep;
                sh.bc_f[0] = (a - before) + pm * (float)keep;             // kept mass of this bin
            } else {
                const int j_lo = (int)(bin == (int)(fk >> 4) ? (fk & 15u) : 0u);
                int j = 15, last_j = j_lo, rank = 0;
                float run = before;
                bool found = false;
                for (; j >= j_lo; --j) {
                    const int allowed = (kb | j) == fk ? min((int)f[j], keep_eq) : (int)f[j];
                    if (allowed <= 0) continue;
                    const float pm = key_mass(kb | j);
                    last_j = j;
                    if (run + pm * (float)allowed > tgt) {
                        rank = min((int)((tgt - run) / pm), allowed - 1);
                        found = true;
                        break;
                    }
                    run += pm * (float)allowed;
                }
                if (!found) {
                    j = last_j;
                    rank = ((kb | j) == fk ? min((int)f[j], keep_eq) : (int)f[j]) - 1;
                }
                sh.bc_i[0] = (int)(kb | j);
                sh.bc_i[1] = max(rank, 0);
            }
        }
        __syncthreads();
        SMP_T(6 + 4 * phase);
        if (phase == 0) { thr_k = fk = (unsigned)sh.bc_i[0]; bin_k = bin; kept = before + sh.bc_f[0]; }
        else if (phase == 1) { fk = (unsigned)sh.bc_i[0]; keep_eq = sh.bc_i[1]; kept = before + sh.bc_f[0]; }
        else { key_d = (unsigned)sh.bc_i[0]; r = sh.bc_i[1]; }
        if (phase < 2) {                                                 // bins below the boundary leave the kept set
            const float part = sh.bc_f[0];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bi = SMP_BINS - 1 - (4 * tid + i);
                if (bi < bin) { m[i] = 0.f; cnt[i] = 0; }
                else if (bi == bin) m[i] = part;
            }
        }
        __syncthreads();
    }

    SMP_T(16);
    int token = 0;
    {
        // ---- the r-th logit (index order) whose key is key_d.  Pass j covers the 1024 consecutive chunks 1024 j + tid, a wave 64
        // of them: index order = (pass, wave, lane, element), one segment count per (pass, wave).
        const int iters = (nchunks + SMP_THREADS - 1) / SMP_THREADS;     // <= 64 (host check), so iters * SMP_WAVES <= 1024
        each([&](int j, int, const float (&v)[8], bool valid) {
            int hits = 0;
            if (valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) hits += (key16(v[e]) == key_d);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o, 64);
            if (lane == 0) sh.scratch[j * SMP_WAVES + wv] = hits;
        });
        __syncthreads();
        SMP_T(17);
        const int seg_hits = tid < SMP_WAVES * iters ? sh.scratch[tid] : 0;
        int seg_total;
        const int seg_before = block_excl_scan(seg_hits, sh.red_i, seg_total);
        if (tid == 0) sh.bc_i[2] = -1;
        __syncthreads();
        if (seg_hits > 0 && seg_before <= r && r < seg_before + seg_hits) { sh.bc_i[2] = tid; sh.bc_i[3] = r - seg_before; }
        __syncthreads();
        const int seg = sh.bc_i[2];
        if (seg >= 0 && wv == seg % SMP_WAVES) {                         // one 1 KiB re-read by the owning wave (a register of a
            const int c = (seg / SMP_WAVES) * SMP_THREADS + tid;         // run-time index would live in scratch memory)
            float v[8];
            int hits = 0;
            if (c < nchunks) {
                decode8(*reinterpret_cast<const bf16x8*>(row + 8 * c), bitmap[c], c, sh, true, t_bf, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) hits += (key16(v[e]) == key_d);
            }
            const int before = wave_incl_scan(hits) - hits;
            const int local = sh.bc_i[3] - before;
            if (hits > 0 && local >= 0 && local < hits) {
                int seen = 0;
                for (int e = 0; e < 8; ++e)
                    if (key16(v[e]) == key_d && seen++ == local) sh.bc_i[6] = 8 * c + e;
            }
        }
        __syncthreads();
        token = seg >= 0 ? sh.bc_i[6] : 0;
        SMP_T(18);
    }
    if (tid == 0) commit_token(p, b, n, token);
}

extern "C" int q3asr_sample_step(const void* logits, int64_t ld, int B, int V, int32_t* out_tokens, int max_out,
                                 int32_t* n_gen, int32_t* finished, int32_t* cur_tok, int32_t* pos, const int32_t* max_new,
                                 float temperature, float top_p, int top_k, float penalty, int ctx, uint64_t seed,
                                 int eos0, int eos1, int row_base, void* stream) {
    Q3_REQUIRE(logits && out_tokens && n_gen && finished && cur_tok && pos && max_new, "sample_step: null pointer");
    Q3_REQUIRE(B > 0 && V > 0 && V % 32 == 0 && ld % 8 == 0 && ld >= V, "sample_step: V=%d must be a multiple of 32, ld aligned", V);
    Q3_REQUIRE(V <= 64 * 8 * SMP_THREADS, "sample_step: V=%d too large (at most %d)", V, 64 * 8 * SMP_THREADS);
    Q3_REQUIRE(temperature >= 0.0f && max_out > 0, "sample_step: bad temperature/max_out");
    SampleParams p;
    p.logits = (const bf16*)logits; p.ld = ld; p.V = V; p.out_tokens = out_tokens; p.max_out = max_out;
    p.n_gen = n_gen; p.finished = finished; p.cur_tok = cur_tok; p.pos = pos; p.max_new = max_new;
    p.temperature = temperature; p.top_p = top_p; p.penalty = penalty; p.top_k = top_k; p.ctx = ctx; p.seed = seed;
    p.eos0 = eos0; p.eos1 = eos1; p.row_base = row_base;
    const size_t bitmap = (size_t)((V / 8 + 15) / 16) * 16;
    const bool res = V <= SMP_RES * 8 * SMP_THREADS && g_q3_option[Q3ASR_OPT_SAMPLER_RESIDENT];
    if (temperature == 0.0f) {
        hipLaunchKernelGGL(sample_greedy_kernel, dim3(B), dim3(SMP_THREADS), bitmap, (hipStream_t)stream, p);
    } else {
        const size_t lds = SMP_BINS * 8 + 2 * SMP_WIN_KEYS * 4 + bitmap + (res ? SMP_RES_LDS_BYTES : 0);   // up to 116 KiB: needs the opt-in (once per process)
        static std::atomic<bool> attr_set{false};
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)sample_draw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      SMP_BINS * 8 + 2 * SMP_WIN_KEYS * 4 + SMP_RES_LDS_BYTES + SMP_RES * SMP_THREADS);
            (void)hipFuncSetAttribute((const void*)sample_draw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      SMP_BINS * 8 + 2 * SMP_WIN_KEYS * 4 + 64 * SMP_THREADS);
            attr_set = true;
        }
        if (res) hipLaunchKernelGGL(sample_draw_kernel<true>, dim3(B), dim3(SMP_THREADS), lds, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(sample_draw_kernel<false>, dim3(B), dim3(SMP_THREADS), lds, (hipStream_t)stream, p);
    }
    return q3_check_launch("sample_step");
}
