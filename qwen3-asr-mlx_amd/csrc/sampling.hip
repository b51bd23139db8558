// Token selection for a batch of sequences, one 1024-thread workgroup per sequence
// (reference generate.py:88-215 `sample` + the bookkeeping of the loop at :278-313).
//
//   repetition penalty  (generate.py:88-131): unique ids among the last `ctx` GENERATED tokens get
//                       l>0 ? l/p : l*p, with p and the result rounded to bf16 (weakly-typed scalar
//                       against bf16 logits).  The logits row is never written: penalised ids are
//                       marked in an LDS bitmap (one byte per 16-byte load of 8 logits) and their new
//                       values kept in an LDS list, so the scan stays a pure streaming read.
//   greedy              (generate.py:203-204): argmax, lowest index wins ties (mx.argmax).
//   temperature/top-k/top-p/categorical (generate.py:206-215,134-163): exact top-k threshold (ties
//                       kept) by a two-level radix select on the 16-bit bf16 keys; nucleus by mass
//                       histograms over the same keys (equivalent to sort + exclusive cumsum with
//                       ties ordered by index); inverse-CDF draw with Philox4x32-10.  Filter maths in
//                       fp32 (the reference's bf16 cumsum saturation is not reproduced: distributional
//                       parity only).
//   bookkeeping         append to out_tokens unless finished, EOS / budget -> finished, advance pos.
#include "common.h"

#define SMP_THREADS 1024
#define SMP_MAXCTX 1024

struct SampleParams {
    const bf16* logits; long ld; int V;
    int* out_tokens; int max_out;
    int* n_gen; int* finished; int* cur_tok; int* pos; const int* max_new;
    float temperature, top_p, penalty; int top_k; int ctx;
    unsigned long long seed;
    int eos0, eos1;
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

struct SmpShared {
    int ids[SMP_MAXCTX];
    float vals[SMP_MAXCTX];
    int n_over;
    float red_f[SMP_THREADS / 64];
    int red_i[SMP_THREADS / 64];
    float hist_f[256];
    int hist_i[256];
    float scan_f[SMP_THREADS];
    int scan_i[SMP_THREADS];
    float bc_f[4];
    int bc_i[4];
};

// 8 consecutive (penalty-adjusted, optionally temperature-scaled) logits of chunk c
__device__ __forceinline__ void load8(const bf16* row, int c, const unsigned char* bitmap, const SmpShared& sh,
                                      bool scaled, float t_bf, float (&v)[8]) {
    const bf16x8 raw = *reinterpret_cast<const bf16x8*>(row + 8 * c);
    const unsigned bits = bitmap[c];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(raw[e]);
    if (bits) {
        for (int e = 0; e < 8; ++e)
            if (bits & (1u << e)) {
                const int id = 8 * c + e;
                for (int j = 0; j < sh.n_over; ++j)
                    if (sh.ids[j] == id) { v[e] = sh.vals[j]; break; }
            }
    }
    if (scaled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = round_bf(v[e] / t_bf);      // logits / temperature in bf16 (generate.py:206)
    }
}

__device__ __forceinline__ float block_sum_f(float v, SmpShared& sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red_f[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < SMP_THREADS / 64; ++i) t += sh.red_f[i];
    return t;
}

__global__ __launch_bounds__(SMP_THREADS) void sample_step_kernel(const SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ SmpShared sh;
    unsigned char* bitmap = dyn;                                        // V/8 bytes
    const int b = blockIdx.x, tid = threadIdx.x;
    const bf16* row = p.logits + (long)b * p.ld;
    const int nchunks = p.V / 8;
    const int n = p.n_gen[b];

    for (int i = tid; i < (nchunks + 3) / 4; i += SMP_THREADS) reinterpret_cast<unsigned*>(bitmap)[i] = 0u;
    if (tid == 0) sh.n_over = 0;
    __syncthreads();
    // ---- repetition penalty over the last min(n, ctx) generated tokens (first token: none, generate.py:284)
    const int cnt = (p.penalty != 1.0f && p.ctx > 0) ? min(min(n, p.ctx), SMP_MAXCTX) : 0;
    int my_id = -1;
    if (tid < cnt) my_id = p.out_tokens[(long)b * p.max_out + n - cnt + tid];
    if (tid < cnt) sh.scan_i[tid] = my_id;
    __syncthreads();
    if (tid < cnt && my_id >= 0 && my_id < p.V) {
        bool first = true;
        for (int j = 0; j < tid; ++j)
            if (sh.scan_i[j] == my_id) { first = false; break; }
        if (first) {
            const float l = bf2f(row[my_id]);
            const float pb = round_bf(p.penalty);
            const float nv = round_bf(l > 0.f ? l / pb : l * pb);
            const int slot = atomicAdd(&sh.n_over, 1);
            sh.ids[slot] = my_id;
            sh.vals[slot] = nv;
            atomicOr(reinterpret_cast<unsigned*>(bitmap) + (my_id >> 5), 1u << (my_id & 31));
        }
    }
    __syncthreads();

    const bool sampling = p.temperature != 0.0f;
    const float t_bf = sampling ? round_bf(p.temperature) : 1.0f;
    int token;

    // ---- pass 1: max value, lowest index (this IS greedy)
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int c0 = tid; c0 < nchunks; c0 += 4 * SMP_THREADS) {            // 4 x 16-byte loads in flight per thread
        bf16x8 raw[4];
        unsigned bits[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min(c0 + u * SMP_THREADS, nchunks - 1);
            raw[u] = *reinterpret_cast<const bf16x8*>(row + 8 * c);
            bits[u] = bitmap[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * SMP_THREADS;
            if (c >= nchunks) break;
            float v[8];
            if (bits[u] || sampling) {
                load8(row, c, bitmap, sh, sampling, t_bf, v);             // rare: penalised id in this chunk, or temperature path
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = bf2f(raw[u][e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (v[e] > best) { best = v[e]; best_i = 8 * c + e; }
        }
    }
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
        for (int i = 1; i < SMP_THREADS / 64; ++i)
            if (sh.red_f[i] > bv || (sh.red_f[i] == bv && sh.red_i[i] < bi)) { bv = sh.red_f[i]; bi = sh.red_i[i]; }
        sh.bc_f[0] = bv;
        sh.bc_i[0] = bi;
    }
    __syncthreads();
    const float xmax = sh.bc_f[0];
    token = sh.bc_i[0];

    if (sampling) {
        // ---- top-k threshold key (keep key >= thr_k); 0 = keep everything
        unsigned thr_k = 0;
        if (p.top_k > 0 && p.top_k < p.V) {
            for (int lvl = 0; lvl < 2; ++lvl) {
                if (tid < 256) sh.hist_i[tid] = 0;
                __syncthreads();
                const unsigned hi_sel = sh.bc_i[1];
                for (int c = tid; c < nchunks; c += SMP_THREADS) {
                    float v[8];
                    load8(row, c, bitmap, sh, true, t_bf, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned k = key16(v[e]);
                        if (lvl == 0) atomicAdd(&sh.hist_i[k >> 8], 1);
                        else if ((k >> 8) == hi_sel) atomicAdd(&sh.hist_i[k & 255], 1);
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    int need = lvl == 0 ? p.top_k : sh.bc_i[2];
                    int bin = 255;
                    for (; bin > 0; --bin) {
                        if (sh.hist_i[bin] >= need) break;
                        need -= sh.hist_i[bin];
                    }
                    if (lvl == 0) { sh.bc_i[1] = bin; sh.bc_i[2] = need; }
                    else sh.bc_i[3] = (sh.bc_i[1] << 8) | bin;
                }
                __syncthreads();
            }
            thr_k = (unsigned)sh.bc_i[3];
        }
        // ---- nucleus: find the crossing key fk and how many equal-key tokens (by index) survive
        unsigned fk = thr_k;
        int keep_eq = 0x7fffffff;
        float Z = 0.f;
        {   // total mass of the top-k set
            float z = 0.f;
            for (int c = tid; c < nchunks; c += SMP_THREADS) {
                float v[8];
                load8(row, c, bitmap, sh, true, t_bf, v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (key16(v[e]) >= thr_k) z += __expf(v[e] - xmax);
            }
            Z = block_sum_f(z, sh);
        }
        if (p.top_p < 1.0f) {
            const float target = p.top_p * Z;
            for (int lvl = 0; lvl < 2; ++lvl) {
                if (tid < 256) { sh.hist_f[tid] = 0.f; sh.hist_i[tid] = 0; }
                __syncthreads();
                const unsigned hi_sel = sh.bc_i[1];
                for (int c = tid; c < nchunks; c += SMP_THREADS) {
                    float v[8];
                    load8(row, c, bitmap, sh, true, t_bf, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned k = key16(v[e]);
                        if (k < thr_k) continue;
                        if (lvl == 0) atomicAdd(&sh.hist_f[k >> 8], __expf(v[e] - xmax));
                        else if ((k >> 8) == hi_sel) atomicAdd(&sh.hist_i[k & 255], 1);
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    if (lvl == 0) {
                        float above = 0.f;
                        int bin = 255;
                        for (; bin > 0; --bin) {
                            if (above + sh.hist_f[bin] >= target) break;
                            above += sh.hist_f[bin];
                        }
                        sh.bc_i[1] = bin;
                        sh.bc_f[1] = above;
                    } else {
                        float above = sh.bc_f[1];
                        int bin = 255;
                        float pm = 0.f;
                        for (; bin >= 0; --bin) {
                            pm = __expf(key16_to_float(((unsigned)sh.bc_i[1] << 8) | bin) - xmax);
                            const float mass = pm * sh.hist_i[bin];
                            if (sh.hist_i[bin] > 0 && above + mass >= target) break;
                            above += mass;
                        }
                        if (bin < 0) { bin = 0; sh.bc_i[2] = 0x7fffffff; }
                        else {
                            // token i (0-based among equals, index order) survives iff above + i*pm < target
                            int k = (int)ceilf((target - above) / pm);
                            k = max(1, min(k, sh.hist_i[bin]));
                            sh.bc_i[2] = k;
                        }
                        sh.bc_i[3] = (sh.bc_i[1] << 8) | bin;
                        sh.bc_f[1] = above;
                    }
                }
                __syncthreads();
            }
            const unsigned cross = (unsigned)sh.bc_i[3];
            if (cross >= thr_k) { fk = cross; keep_eq = sh.bc_i[2]; }
        }
        // ---- categorical draw over the kept set, inverse CDF in index order.
        // contiguous id ranges per thread so ranks among equal keys follow index order
        const int per = (nchunks + SMP_THREADS - 1) / SMP_THREADS;
        const int c_lo = min(tid * per, nchunks), c_hi = min(c_lo + per, nchunks);
        int eq_cnt = 0;
        for (int c = c_lo; c < c_hi; ++c) {
            float v[8];
            load8(row, c, bitmap, sh, true, t_bf, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) eq_cnt += (key16(v[e]) == fk);
        }
        sh.scan_i[tid] = eq_cnt;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int i = 0; i < SMP_THREADS; ++i) { const int t = sh.scan_i[i]; sh.scan_i[i] = run; run += t; }
        }
        __syncthreads();
        int rank = sh.scan_i[tid];
        const int rank0 = rank;
        float mass = 0.f;
        for (int c = c_lo; c < c_hi; ++c) {
            float v[8];
            load8(row, c, bitmap, sh, true, t_bf, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned k = key16(v[e]);
                const bool keep = k > fk || (k == fk && rank < keep_eq);
                rank += (k == fk);
                if (keep) mass += __expf(v[e] - xmax);
            }
        }
        sh.scan_f[tid] = mass;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int i = 0; i < SMP_THREADS; ++i) tot += sh.scan_f[i];
            const float u = philox_uniform(p.seed, (unsigned)n, (unsigned)b);
            const float tgt = u * tot;
            float run = 0.f;
            int owner = -1, last_nonempty = 0;
            for (int i = 0; i < SMP_THREADS; ++i) {
                if (sh.scan_f[i] > 0.f) last_nonempty = i;
                if (owner < 0 && sh.scan_f[i] > 0.f && run + sh.scan_f[i] > tgt) { owner = i; sh.bc_f[2] = tgt - run; }
                if (owner < 0) run += sh.scan_f[i];
            }
            if (owner < 0) { owner = last_nonempty; sh.bc_f[2] = 3.0e38f; }
            sh.bc_i[0] = owner;
        }
        __syncthreads();
        if (tid == sh.bc_i[0]) {
            float rem = sh.bc_f[2], run = 0.f;
            int r2 = rank0, pick = -1, last = token;
            for (int c = c_lo; c < c_hi && pick < 0; ++c) {
                float v[8];
                load8(row, c, bitmap, sh, true, t_bf, v);
                for (int e = 0; e < 8; ++e) {
                    const unsigned k = key16(v[e]);
                    const bool keep = k > fk || (k == fk && r2 < keep_eq);
                    r2 += (k == fk);
                    if (!keep) continue;
                    last = 8 * c + e;
                    run += __expf(v[e] - xmax);
                    if (run > rem) { pick = last; break; }
                }
            }
            sh.bc_i[1] = pick >= 0 ? pick : last;
        }
        __syncthreads();
        token = sh.bc_i[1];
    }

    // ---- bookkeeping (generate.py:286-313): EOS is recorded, then the sequence stops
    if (tid == 0) {
        if (!p.finished[b]) {
            if (n < p.max_out) p.out_tokens[(long)b * p.max_out + n] = token;
            const int n1 = n + 1;
            p.n_gen[b] = n1;
            p.cur_tok[b] = token;
            if (token == p.eos0 || token == p.eos1 || n1 >= p.max_new[b] || n1 >= p.max_out) p.finished[b] = 1;
            else p.pos[b] = p.pos[b] + 1;
        }
    }
}

extern "C" int q3asr_sample_step(const void* logits, int64_t ld, int B, int V, int32_t* out_tokens, int max_out,
                                 int32_t* n_gen, int32_t* finished, int32_t* cur_tok, int32_t* pos, const int32_t* max_new,
                                 float temperature, float top_p, int top_k, float penalty, int ctx, uint64_t seed,
                                 int eos0, int eos1, void* stream) {
    Q3_REQUIRE(logits && out_tokens && n_gen && finished && cur_tok && pos && max_new, "sample_step: null pointer");
    Q3_REQUIRE(B > 0 && V > 0 && V % 32 == 0 && ld % 8 == 0 && ld >= V, "sample_step: V=%d must be a multiple of 32, ld aligned", V);
    Q3_REQUIRE(V <= 131072 * 8, "sample_step: V=%d too large for the LDS bitmap", V);
    Q3_REQUIRE(temperature >= 0.0f && max_out > 0, "sample_step: bad temperature/max_out");
    SampleParams p;
    p.logits = (const bf16*)logits; p.ld = ld; p.V = V; p.out_tokens = out_tokens; p.max_out = max_out;
    p.n_gen = n_gen; p.finished = finished; p.cur_tok = cur_tok; p.pos = pos; p.max_new = max_new;
    p.temperature = temperature; p.top_p = top_p; p.penalty = penalty; p.top_k = top_k; p.ctx = ctx; p.seed = seed;
    p.eos0 = eos0; p.eos1 = eos1;
    const size_t lds = (size_t)((V / 8 + 15) / 16) * 16;
    hipLaunchKernelGGL(sample_step_kernel, dim3(B), dim3(SMP_THREADS), lds, (hipStream_t)stream, p);
    return q3_check_launch("sample_step");
}
