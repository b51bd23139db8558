/* libq3asr_hip.so -- C-ABI of the MI355X (gfx950) Qwen3-ASR hot path.
 *
 * The reference (gabrimatic/qwen3-asr-mlx) is pure Python with no FFI: its hot path is a sequence
 * of calls into the MLX runtime.  This library is what a maintainer would bind INSTEAD of those
 * MLX calls (ctypes stub in INTEGRATION.md); each entry point names the reference call site(s) it
 * replaces (paths relative to src/qwen3_asr_mlx/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; the caller owns all memory;
 *     nothing here allocates, frees or synchronises; every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*) and is legal inside hipGraph stream capture;
 *   - bf16 buffers are passed as void* (16-bit brain-float, row-major);
 *   - return value: 0 = ok, -1 = bad argument, -2 = HIP launch/runtime error;
 *     q3asr_last_error() returns a thread-local message for the last failure.
 */
#ifndef Q3ASR_H
#define Q3ASR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int q3asr_version(void);             /* 100 = 0.1.0 */
const char* q3asr_arch(void);        /* "gfx950" */
const char* q3asr_last_error(void);

/* ---- audio.py:211-278  log_mel_spectrogram (numpy STFT loop + filterbank matmul) ------------
 * audio  fp32 [B][audio_stride] zero-padded, lens_dev int32 [B] (samples, >= 160);
 * window400 fp32 [400] symmetric Hann; twiddle_cs fp64 [400][2] = (cos, sin)(2*pi*i/400);
 * fb fp32 [128][201]; fb_range int32 [128][2] = [first nonzero, last nonzero + 1);
 * out fp32 [B][128][Fmax] (frames >= len/160 are written as 0); gmax_key uint32 [B] scratch. */
int q3asr_log_mel(const float* audio, int64_t audio_stride, const int32_t* lens_dev, int B, int Fmax,
                  const float* window400, const double* twiddle_cs, const float* fb, const int32_t* fb_range,
                  float* out, uint32_t* gmax_key, void* stream);

/* ---- audio.py:238-246  log_mel_spectrogram for ANY (n_fft, hop_length, n_mels) of the reference signature (the call above
 * is the specialisation for its defaults 400 / 160 / 128, which is what the encoder consumes).
 * window fp32 [n_fft] = np.hanning(n_fft); twiddle_cs fp64 [n_fft][2] = (cos, sin)(2*pi*i/n_fft); fb fp32 [n_mels][n_fft/2+1];
 * frames per utterance = (len + 2*(n_fft/2) - n_fft) / hop_length (the reference drops the last frame, audio.py:266);
 * out fp32 [B][n_mels][Fmax], frames beyond an utterance's count are written as 0; gmax_key uint32 [B] scratch. */
int q3asr_log_mel_generic(const float* audio, int64_t audio_stride, const int32_t* lens_dev, int B, int Fmax,
                          int n_fft, int hop_length, int n_mels, const float* window, const double* twiddle_cs,
                          const float* fb, float* out, uint32_t* gmax_key, void* stream);

/* ---- encoder.py:254-273  chunking + conv2d1 + GELU -------------------------------------------
 * mel fp32 [B][128][Fmax]; chunk ci = frames [chunk_f0[ci], +100) of utterance chunk_b[ci];
 * w9 fp32 [C1][kt][kf]; out bf16 [n_chunks][50][64][C1] (time-major NHWC). */
int q3asr_conv1_gelu(const float* mel, int Fmax, const int32_t* chunk_b, const int32_t* chunk_f0, int n_chunks,
                     const float* w9, const float* bias, void* out, int C1, void* stream);

/* the same with float32 output (reference-semantics encoder mode: fp32 activations, see "parity mode" below) */
int q3asr_conv1_gelu_f32(const float* mel, int Fmax, const int32_t* chunk_b, const int32_t* chunk_f0, int n_chunks,
                         const float* w9, const float* bias, float* out, int C1, void* stream);

/* ---- encoder.py:274-275  conv2d2 / conv2d3 (+GELU) as implicit GEMM --------------------------
 * x bf16 [imgs][Hin][Win][Cin]; w bf16 in chunk-major K order [Cout][ceil(Cin/64)][9 taps][64] (tap = 3*kh + kw over
 * x's two spatial dims, zero channels beyond Cin: the nine taps of a 64-channel chunk are consecutive K tiles);
 * out row m = (img*Hout + oh)*Wout + ow (or row_map[m], <0 = skip); act: 0 none, 1 GELU;
 * out_kind: 0 bf16, 1 fp32. */
int q3asr_conv3x3s2_bf16(const void* x, int imgs, int Hin, int Win, int Cin, const void* w, int Cout,
                         const float* bias, int act, void* out, int out_kind, int64_t ldo,
                         const int32_t* row_map, void* stream);

/* ---- every nn.Linear on the large-M side: encoder.py:74-76,86,118-119,279,320-321;
 *      decoder.py:92-97,136-138,165 (prefill) ---------------------------------------------------
 * out[M,N] = epilogue(A[M,K] . W[N,K]^T): + bias fp32[N]; act 0 none / 1 exact GELU /
 * 2 SwiGLU over 16-row-interleaved (gate,up) weight rows (out has N/2 columns);
 * res_kind 0 none / 1 fp32 add / 2 bf16 add with MLX rounding (bf16(bf16(acc)+res));
 * res_mod > 0: residual row = m % res_mod (per-chunk positional embedding, encoder.py:284-286);
 * row_map: optional destination row per m, <0 = drop (pad-token strip, encoder.py:289-293). */
int q3asr_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K,
                    const float* bias, int act, const void* res, int res_kind, int64_t ldres, int res_mod,
                    void* out, int out_kind, int64_t ldo, const int32_t* row_map, void* stream);
/* Tuning / test hook: pin the GEMM kernel (0 = automatic choice, 64 = 64x64 tile, 128 = 128x128 tile, 8 = 256x256 8-phase LDS-DMA
 * pipeline where its preconditions hold, 224 / 208 = the same pipeline on 224x256 / 208x256 tiles).  All kernels give bit-identical results. */
int q3asr_gemm_force_kernel(int mode);
/* Diagnostic: 0 = every sub-tile of the 8-phase kernels takes the general (bounds-tested) epilogue, 1 = interior sub-tiles take the short one, 2 (default) = ... and the plain bf16 GEMM reads a sub-tile's residual rows one 16-row block ahead of their use. Same bits in every setting; returns the previous one. Used by tools/ab_gemm_epilogue.py */
int q3asr_gemm_set_interior(int on);
/* Diagnostic: switches between two launch forms with identical results, for A/B measurements in one process (tools/ab_step.py,
 * tools/bench_head.py).  Every option defaults to 1; returns the previous value, or -1 for an unknown option.  A captured
 * graph keeps the form it was captured with. */
#define Q3ASR_OPT_HEAD_XCD_MAP 1    /* tied head: workgroups that share a 128-byte logits line run on one XCD */
#define Q3ASR_OPT_HEAD_STREAM 2     /* tied head (bf16, K = 2048, geometry 2/8/8, <= 32 rows): one workgroup per CU walking its tile groups */
#define Q3ASR_OPT_ATTN_XCD_ORDER 3  /* attn_fwd: workgroups that read one kv head's K/V run on one XCD (0: the (query tile, head, segment) grid order) */
#define Q3ASR_OPT_SAMPLER_RESIDENT 4 /* sample_step with temperature > 0: rows of <= 155,648 logits are held in registers / LDS across the passes (0: every pass re-reads the row) */
#define Q3ASR_OPT_ATTN_HALF_WG 5 /* attn_decode_fused with more workgroups than CUs (33-64 rows): four-wave workgroups, two per CU, each wave computing two of the eight per-wave partials (0: eight-wave workgroups in two rounds) */
int q3asr_set_option(int option, int value);
/* Diagnostic: workgroup 0 of the following 8-phase launches records wall_clock64() stamps into buf (16 x u64 device memory; NULL = off) */
int q3asr_gemm_set_stamp(void* buf);
int q3asr_gemm_set_stamp_wg(int workgroup);   /* which workgroup (blockIdx.x) records them; default 0 */

/* ---- nn.LayerNorm (encoder.py:112,117,319): x fp32 [M][D] -> y bf16 [M][D] ------------------ */
int q3asr_layernorm_f32_bf16(const float* x, const float* w, const float* b, void* y, int M, int D, float eps,
                             void* stream);

/* ---- parity mode of the encoder (reference semantics: float32 activations through encoder.py:272-323, nothing is cast
 * after audio.py:278).  An fp32 activation is split into three bf16 pieces hi + mid + lo laid side by side along K,
 * [M][3K] = [hi | mid | lo]; q3asr_gemm_bf16 / q3asr_conv3x3s2_bf16 on it against [W | W | W] give an fp32-accurate
 * x . W^T (exact bf16 products, fp32 accumulation).  csrc/parity.hip. */
int q3asr_split3_f32(const float* x, int64_t ldx, void* y, int64_t M, int K, void* stream);
/* LayerNorm with the fp32 result split: y bf16 [M][3D] */
int q3asr_layernorm_f32_split3(const float* x, const float* w, const float* b, void* y, int M, int D, float eps, void* stream);
/* mx.fast.scaled_dot_product_attention at encoder.py:82-84 on fp32 q/k/v: qkv fp32 [N][ld >= 3*d_model] (q | k | v),
 * windows (seg_start, seg_len) of at most 128 tokens, head_dim 64; out fp32 [N][ldo]. */
int q3asr_attn_fwd_f32(const float* qkv, int64_t ld, int d_model, const int32_t* seg_start, const int32_t* seg_len, int n_segs,
                       int max_len, int n_heads, float scale, float* out, int64_t ldo, void* stream);

/* ---- fp8 (OCP e4m3) GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (BASELINE.json configs[4]: prefill projections of
 * decoder.py:92-97,136-165 with fp8 weights): C = (A8 . W8^T) * a_scale[m] * w_scale[n] -> bf16 [M][ldo].
 * A8 bytes [M][lda] (per-token scale a_scale[M], from the two quantisers below), W8 bytes [N][ldw] (per-row scale w_scale[N]).
 * act: 0 none, 2 SwiGLU (weights interleaved as for q3asr_gemm_bf16); res_kind: 0 none, 2 bf16 residual. */
int q3asr_gemm_fp8(const void* A8, int64_t lda, const float* a_scale, const void* W8, int64_t ldw, const float* w_scale,
                   int M, int N, int K, int act, const void* res, int res_kind, int64_t ldres, void* out, int64_t ldo, void* stream);

/* RMSNorm (decoder.py:189,194) with the bf16 result quantised per row: y8 = e4m3(y / scale[m]), scale = max|y| / 448 */
int q3asr_rmsnorm_bf16_fp8(const void* x, int64_t ldx, const void* w, void* y8, int64_t ldy, float* scale, int M, int D, float eps,
                           void* stream);
/* per-row quantisation of a bf16 matrix (attention output, SwiGLU output) */
int q3asr_quantize_rows_fp8(const void* x, int64_t ldx, void* y8, int64_t ldy, float* scale, int M, int D, void* stream);

/* ---- nn.RMSNorm (decoder.py:189,194,251): bf16 -> bf16; row_idx optional source-row gather -- */
int q3asr_rmsnorm_bf16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, const int32_t* row_idx,
                       int M, int D, float eps, void* stream);

/* ---- mx.fast.scaled_dot_product_attention (encoder.py:82-84 with the block mask of :210-229;
 *      decoder.py:158-163 prefill) ---------------------------------------------------------------
 * Independent segments s: Q/O rows [seg_q_start[s], +seg_len[s]); keys 0..seg_k_off[s]+seg_len[s]-1 at
 * K/V + kv_seg_off[s] + kv_head*kv_head_stride + key*kv_row_stride (elements).
 * seg_k_off (optional, NULL = all 0): keys already cached in front of the segment's first query row -- query row i sits at
 * position seg_k_off[s] + i and, when causal, sees keys 0..seg_k_off[s]+i: a multi-token call on a non-empty cache, i.e.
 * decoder.py:151-158 with cache.offset > 0, the case create_additive_causal_mask(N, offset) (decoder.py:71-80) exists for.
 * max_len = max query rows of a segment (grid), max_klen = max keys of a segment (= max_len when seg_k_off is NULL). */
int q3asr_attn_fwd(const void* Q, int64_t q_stride, const void* K, const void* V, const int64_t* kv_seg_off,
                   int64_t kv_head_stride, int64_t kv_row_stride, void* O, int64_t o_stride,
                   const int32_t* seg_q_start, const int32_t* seg_len, const int32_t* seg_k_off, int n_segs, int max_len,
                   int max_klen, int n_heads, int n_kv_heads, int head_dim, int causal, float scale, void* stream);

/* Diagnostic: workgroup `workgroup` of the following attn_fwd launches records wall_clock64() stamps into buf (64 x u64 device memory; NULL = off) */
int q3asr_attn_set_stamp(void* buf, int workgroup);

/* ---- generate.py:20-81  prepare_inputs ------------------------------------------------------- */
/* E: embedding table with V rows; enc: n_enc encoder rows.  Index vectors are device-resident and cannot be validated
 * by the host: a token id outside [0, V) or an encoder row >= n_enc yields a zero row, never an out-of-bounds read. */
int q3asr_embed_merge(const int32_t* ids, const int32_t* audio_src, const void* E, int V, const float* enc, int n_enc,
                      void* out, int M, int H, void* stream);

/* ---- decoder.py:140-156  per-head QK-RMSNorm + nn.RoPE + KVCache.update ----------------------
 * qkv: bf16 [M][ld] (is_slab=0) or fp32 split-K slabs [n_slabs][..] (is_slab=1, ld/slab_stride in
 * floats); q_out bf16 [M][nh*128]; caches bf16 [B][nkv][Lmax][128] of ONE layer;
 * row_b/row_pos int32 [M]: sequence and absolute position of each row. */
int q3asr_qk_norm_rope_kv(const void* qkv, int is_slab, int64_t ld, int n_slabs, int64_t slab_stride,
                          const int32_t* row_b, const int32_t* row_pos, const void* qn_w, const void* kn_w,
                          const float* inv_freq, float eps, void* q_out, void* k_cache, void* v_cache, int M,
                          int nh, int nkv, int head_dim, int Lmax, void* stream);

/* ---- decode step (generate.py:289-313 -> decoder.py:219-253 at T=1), M = batch <= 64 --------- */
/* ---- streaming layouts of the graph-captured decode step (csrc/decode.hip) ---------------------
 * W16 [N/16][K/32][64][8]: lane (g=lane>>4, li=lane&15) holds W[16rt+li][32ks+8g+j];
 * X16 [K/32][MT][64][8]:   lane holds X[16mt+li][32ks+8g+j], MT = 1/2/4 for M <= 16/32/64.
 * skinny3: one wave per 16-row weight tile and K slice, global -> MFMA, no LDS.
 *   epi 0: fp32 slabs [n_splits][M][N]; epi 1: bf16 row-major out [M][ldo]; epi 2: SwiGLU of
 *   (gate tile, up tile) pairs -> bf16 X16 (K' = N/2).  waves_per_wg, tiles_per_wave in {1,2,4};
 *   wg_ksplit != 0: the waves of a workgroup split its K slice and reduce through LDS (fixed order).
 *   w_scale != NULL: W16 is fp8 e4m3 in W16F8 layout [N/16][K/64][64][16 B] with one fp32 scale per row
 *   (BASELINE.json configs[4]); NULL: bf16 W16. */
int q3asr_skinny3_gemm(const void* X16, int M, const void* W16, const float* w_scale, int N, int K, int n_splits,
                       int epi, int waves_per_wg, int tiles_per_wave, int wg_ksplit, float* slab, void* out,
                       int64_t ldo, void* stream);
/* The same GEMM with the residual add + RMSNorm that produces its activation operand fused into its prologue, for
 * M <= 2 (single-stream latency): every workgroup rebuilds x = RMSNorm(bf16(h_in + bf16(sum in_slab)))*norm_w from the
 * previous GEMM's slabs (arithmetic of q3asr_slab_resnorm_x16, bit-identical) while its first weight tiles are in flight;
 * workgroup 0 also writes the new residual stream to h_out (must differ from h_in; in_slab must differ from slab).
 * waves_per_wg must be 4, epi 0 or 2.  Replaces q3asr_slab_resnorm_x16 + q3asr_skinny3_gemm (decoder.py:189-196). */
int q3asr_skinny3_gemm_norm(const float* in_slab, int in_n_slabs, int64_t in_slab_stride, const void* h_in, void* h_out,
                            const void* norm_w, float eps, int M, const void* W16, const float* w_scale, int N, int K,
                            int n_splits, int epi, int waves_per_wg, int tiles_per_wave, int wg_ksplit, float* slab,
                            void* out, int64_t ldo, void* stream);
/* skinny4: the same product C[M, N] = X . W^T on the same W16 / W16F8 / X16 layouts (decoder.py:92-97,136-165 at T = 1, and the
 * tied head decoder.py:252), one workgroup of `waves` waves per `tiles_per_wg` consecutive 16-row weight tiles; wave w of slab s
 * owns k-steps [(s waves + w) ksteps_per_wave, +ksteps_per_wave) and issues all of its loads before its first MFMA; the waves'
 * partial sums are added in wave order through LDS.  n_splits = K / (32 ksteps_per_wave waves) slabs (epi 0), which must be 1
 * for epi 1 / 2.  Results depend on (ksteps_per_wave, waves) only, never on M or tiles_per_wg.  Instantiated geometries
 * (tiles, k-steps, waves): epi 0 (2,2,8) (2,4,8) (2,6,8) (2,3,16); epi 1 (2,2,8) (2,8,8) (4,8,8); epi 2 (2,2,8) (4,4,16)
 * (2,4,16); fp8 weights (w_scale != NULL) need an even k-step count; M > 32 needs waves * tiles <= 40. */
int q3asr_skinny4_gemm(const void* X16, int M, const void* W16, const float* w_scale, int N, int K, int epi, int tiles_per_wg,
                       int ksteps_per_wave, int waves, float* slab, void* out, int64_t ldo, void* stream);
/* q3asr_skinny4_gemm with the residual add + RMSNorm of its activation operand fused into the prologue (M <= 2, epi 0 or 2): the
 * arguments of q3asr_skinny3_gemm_norm, the geometry of q3asr_skinny4_gemm; bit-identical to q3asr_slab_resnorm_x16 followed by
 * q3asr_skinny4_gemm (decoder.py:189-196 at batch 1). */
int q3asr_skinny4_gemm_norm(const float* in_slab, int in_n_slabs, int64_t in_slab_stride, const void* h_in, void* h_out,
                            const void* norm_w, float eps, int M, const void* W16, const float* w_scale, int N, int K, int epi,
                            int tiles_per_wg, int ksteps_per_wave, int waves, float* slab, void* out, int64_t ldo, void* stream);
/* h = bf16(h + bf16(sum slabs)); x16 = RMSNorm(h)*w in X16 layout (decoder.py:191-196,189,251) */
int q3asr_slab_resnorm_x16(const float* slab, int n_slabs, int64_t slab_stride, const void* h_in, void* h_out,
                           const void* w, void* x16, int M, int N, float eps, void* stream);
/* h = E[tok[m]] (tok NULL: h = E[m]); x16 = RMSNorm(h)*w (generate.py:294 + decoder.py:189).  E has V rows; an id outside
 * [0, V) gathers a zero row (device-resident ids cannot be validated by the host). */
int q3asr_embed_norm_x16(const int32_t* tok, const void* E, int V, void* h_out, const void* w, void* x16, int M, int N,
                         float eps, void* stream);
/* decode attention with decoder.py:144-156 (QK-RMSNorm, RoPE, cache append) fused into its prologue;
 * input: the qkv projection as n_slabs fp32 slabs [n_slabs][B][ld]; output att in X16 layout.
 * The cache is processed in 1024-token chunks, one workgroup each; sequences longer than a chunk are merged
 * through `workspace` (q3asr_attn_decode_fused_workspace(B, nkv, Lmax) bytes, zero-filled once by the caller,
 * left zeroed/ready by every launch; launches sharing a workspace must be ordered on one stream). */
int64_t q3asr_attn_decode_fused_workspace(int B, int nkv, int Lmax);
/* Test hook: pin the number of 1,024-token cache chunks one workgroup sweeps (0 = automatic).  Results do not depend on it. */
int q3asr_attn_decode_set_chunks_per_wg(int n);
int q3asr_attn_decode_fused(const float* qkv_slab, int64_t ld, int n_slabs, int64_t slab_stride, const void* qn_w,
                            const void* kn_w, const float* inv_freq, float eps, void* k_cache, void* v_cache,
                            const int32_t* pos, void* out16, int B, int nh, int nkv, int head_dim, int Lmax,
                            float scale, void* workspace, void* stream);

/* ---- generate.py:88-215 `sample` + loop bookkeeping :286-313, one launch per step ------------ */
/* finished[b]: 0 running, 1 ended (EOS or budget), Q3ASR_ROW_FAILED: the row's logits had no valid maximum / draw (non-finite
 * logits).  A failed row records token 0 and stops; the host raises when it reads the flag (generate.py). */
#define Q3ASR_ROW_FAILED 2
int q3asr_sample_step(const void* logits, int64_t ld, int B, int V, int32_t* out_tokens, int max_out,
                      int32_t* n_gen, int32_t* finished, int32_t* cur_tok, int32_t* pos, const int32_t* max_new,
                      float temperature, float top_p, int top_k, float penalty, int ctx, uint64_t seed,
                      int eos0, int eos1, int row_base, void* stream);
/* row_base: index of this launch's row 0 inside its request -- the random draw of a row is keyed by (seed, token index,
 * row_base + b), so a request decoded as two half-batches draws what the whole batch would. */

/* ---- hipGraph capture of the decode step; HIP events on an explicit stream -------------------- */
int q3asr_graph_begin(void* stream);
int q3asr_graph_end(void* stream, void** graph_exec_out);
int q3asr_graph_launch(void* graph_exec, void* stream);
int q3asr_graph_destroy(void* graph_exec);
int q3asr_event_create(void** ev);
int q3asr_event_record(void* ev, void* stream);
int q3asr_event_elapsed_ms(void* start, void* stop, float* ms);
int q3asr_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* Q3ASR_H */
