// 128-bin log-mel frontend (reference audio.py:211-278) for a batch of utterances.
//
// K1+K2 of SURVEY.md: reflect-pad 200, symmetric Hann(400), 400-point real DFT
// per frame, |.|^2, 128x201 filterbank, log10(max(.,1e-10)), per-utterance
// max-8 floor, (x+4)/4.  The DFT is a direct 201x400 sum accumulated in fp64
// (FP64 vector rate on MI355X is half the fp32 rate; the stage is <1% of a
// batch), which lands within 1-2 ulp of the reference's pocketfft result.
// HBM-bound in principle (1.92 MB in, 1.54 MB out per 30 s clip); one
// workgroup owns 16 consecutive frames so the 2800-sample span is read once.
#include "common.h"

#define MEL_FT 16
#define NFFT 400
#define HOPLEN 160
#define NBINS 201
#define NMELS 128

__device__ __forceinline__ int reflect_index(int j, int len) {
    // numpy.pad(mode="reflect") for any pad width (audio.py:224)
    if (len == 1) return 0;
    const int period = 2 * (len - 1);
    int r = j % period;
    if (r < 0) r += period;
    return r < len ? r : period - r;
}

__device__ __forceinline__ unsigned float_to_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void mel_logpower_kernel(
    const float* __restrict__ audio, long audio_stride, const int* __restrict__ lens,
    const float* __restrict__ window, const double* __restrict__ tw, const float* __restrict__ fb,
    const int* __restrict__ fb_range, float* __restrict__ out, int Fmax, unsigned* __restrict__ gmax_key) {
    __shared__ __attribute__((aligned(16))) double twl[2 * NFFT];      // (cos, sin)
    // The real input folds: X[k] = x0 + x200 cos(pi k) + sum_{n=1}^{199} (x_n + x_{400-n}) cos(2 pi k n / 400)
    //                              - i sum_{n=1}^{199} (x_n - x_{400-n}) sin(2 pi k n / 400)
    // -- the sums and differences of two float32 samples are exact in fp64 and shared by all 201 bins, so a bin costs 200
    // cosine and 199 sine fp64 FMAs per frame instead of 800 (the stage is bound by the fp64 FMA rate: 1.22 -> 0.87 ms per batch).
    __shared__ __attribute__((aligned(16))) double ev[(NFFT / 2 + 1) * MEL_FT];   // [n][f], n = 0 .. 200: x0, pair sums, x200
    __shared__ __attribute__((aligned(16))) double od[(NFFT / 2) * MEL_FT];       // [n][f], n = 0 .. 199: (unused), pair differences
    float* xw = reinterpret_cast<float*>(ev);                          // after the DFT: the power spectrum P[k][f] (ev is dead by then)

    const int b = blockIdx.y;
    const int f0 = blockIdx.x * MEL_FT;
    const int len = lens[b];
    const int nfr = len / HOPLEN;          // 1 + (len + 400 - 400)//160 frames, last dropped (audio.py:226,266)
    if (f0 >= nfr) return;
    const int tid = threadIdx.x;
    const float* x = audio + (long)b * audio_stride;

    for (int i = tid; i < 2 * NFFT; i += 256) twl[i] = tw[i];
    for (int i = tid; i < (NFFT / 2 + 1) * MEL_FT; i += 256) {
        const int f = i / (NFFT / 2 + 1), n = i - f * (NFFT / 2 + 1);          // consecutive threads: consecutive samples
        const int j = (f0 + f) * HOPLEN + n - NFFT / 2;
        const double a = (double)(x[reflect_index(j, len)] * window[n]);       // float32 product (audio.py:232), exact in fp64
        if (n == 0 || n == NFFT / 2) ev[n * MEL_FT + f] = a;
        else {
            const double bq = (double)(x[reflect_index(j + NFFT - 2 * n, len)] * window[NFFT - n]);
            ev[n * MEL_FT + f] = a + bq;
            od[n * MEL_FT + f] = a - bq;
        }
    }
    __syncthreads();

    float pw[MEL_FT];
    if (tid < NBINS) {
        double re[MEL_FT], im[MEL_FT];
#pragma unroll
        for (int f = 0; f < MEL_FT; ++f) { re[f] = ev[f]; im[f] = 0.0; }         // n = 0: cos = 1, sin = 0
        int idx = tid;                                                          // (k n) mod 400 at n = 1
        for (int n = 1; n < NFFT / 2; ++n) {
            const double c = twl[2 * idx], s = twl[2 * idx + 1];
            idx += tid;
            if (idx >= NFFT) idx -= NFFT;
            const double* er = &ev[n * MEL_FT];
            const double* orow = &od[n * MEL_FT];
#pragma unroll
            for (int f = 0; f < MEL_FT; ++f) {
                re[f] = fma(er[f], c, re[f]);
                im[f] = fma(orow[f], s, im[f]);
            }
        }
        {
            const double c = twl[2 * idx];                                       // n = 200: cos(pi k) = +-1, sin = 0
            const double* er = &ev[(NFFT / 2) * MEL_FT];
#pragma unroll
            for (int f = 0; f < MEL_FT; ++f) re[f] = fma(er[f], c, re[f]);
        }
#pragma unroll
        for (int f = 0; f < MEL_FT; ++f) {
            const float mag = hypotf((float)re[f], (float)im[f]);   // np.abs(complex64)
            pw[f] = mag * mag;                                       // ** 2 in float32 (audio.py:266)
        }
    }
    __syncthreads();
    if (tid < NBINS) {
#pragma unroll
        for (int f = 0; f < MEL_FT; ++f) xw[tid * MEL_FT + f] = pw[f];
    }
    __syncthreads();

    const int m = tid & (NMELS - 1);
    const int h = tid >> 7;                                          // frames 8h..8h+7
    const int lo = fb_range[2 * m], hi = fb_range[2 * m + 1];
    float acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = 0.0f;
    for (int k = lo; k < hi; ++k) {
        const float w = fb[m * NBINS + k];
        const f32x4* pr = reinterpret_cast<const f32x4*>(&xw[k * MEL_FT + 8 * h]);
        const f32x4 p0 = pr[0], p1 = pr[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] = fmaf(w, p0[e], acc[e]);
            acc[4 + e] = fmaf(w, p1[e], acc[4 + e]);
        }
    }
    float vmax = -3.0e38f;
    float* orow = out + ((long)b * NMELS + m) * Fmax;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const int fr = f0 + 8 * h + f;
        if (fr < nfr) {
            const float v = log10f(fmaxf(acc[f], 1e-10f));           // audio.py:274
            orow[fr] = v;
            vmax = fmaxf(vmax, v);
        }
    }
    vmax = wave_max(vmax);
    if ((tid & 63) == 0 && vmax > -1.0e38f) atomicMax(&gmax_key[b], float_to_key(vmax));
}

__global__ __launch_bounds__(256) void mel_normalize_kernel(float* __restrict__ out, const int* __restrict__ lens,
                                                            const unsigned* __restrict__ gmax_key, int Fmax) {
    const int b = blockIdx.z, m = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= Fmax) return;
    const int nfr = lens[b] / HOPLEN;
    float* p = out + ((long)b * NMELS + m) * Fmax + f;
    if (f < nfr) {
        const float thr = key_to_float(gmax_key[b]) - 8.0f;          // audio.py:275
        *p = (fmaxf(*p, thr) + 4.0f) / 4.0f;                          // audio.py:276
    } else {
        *p = 0.0f;                                                    // zero padding of the last chunk (encoder.py:262-266)
    }
}

extern "C" int q3asr_log_mel(const float* audio, int64_t audio_stride, const int32_t* lens_dev, int B, int Fmax,
                             const float* window400, const double* twiddle_cs, const float* fb,
                             const int32_t* fb_range, float* out, uint32_t* gmax_key, void* stream) {
    Q3_REQUIRE(audio && lens_dev && window400 && twiddle_cs && fb && fb_range && out && gmax_key, "log_mel: null pointer");
    Q3_REQUIRE(B > 0 && Fmax > 0, "log_mel: B=%d Fmax=%d must be positive", B, Fmax);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(gmax_key, 0, sizeof(uint32_t) * B, s) != hipSuccess) return q3_check_launch("log_mel memset");
    dim3 g1((Fmax + MEL_FT - 1) / MEL_FT, B);
    hipLaunchKernelGGL(mel_logpower_kernel, g1, dim3(256), 0, s, audio, (long)audio_stride, lens_dev, window400,
                       twiddle_cs, fb, fb_range, out, Fmax, gmax_key);
    dim3 g2((Fmax + 255) / 256, NMELS, B);
    hipLaunchKernelGGL(mel_normalize_kernel, g2, dim3(256), 0, s, out, lens_dev, gmax_key, Fmax);
    return q3_check_launch("log_mel");
}

// ---------------------------------------------------------------------------
// The same frontend for ANY (n_fft, hop_length, n_mels) the reference signature accepts (audio.py:238-246): one workgroup
// per frame, the windowed frame in LDS as fp64 (the float32 product x * w is exact there), each thread a direct fp64 DFT of
// its bins (n_fft FMAs per part and bin; the twiddle index k n mod n_fft walks a host-built (cos, sin) table), power in
// float32 as np.abs(complex64) ** 2, then the filterbank rows in ascending k.  Not a hot path (the product's encoder is built
// for the default triple, which keeps the specialised kernel above): written for coverage of the exported function, not speed.
// Frames: 1 + (len + 2 (n_fft / 2) - n_fft) / hop, last dropped (audio.py:224-226,266).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int mel_frames(int len, int n_fft, int hop) {
    const int padded = len + 2 * (n_fft / 2);
    return padded < n_fft ? 0 : (padded - n_fft) / hop;
}

__global__ __launch_bounds__(256) void mel_logpower_generic_kernel(
    const float* __restrict__ audio, long audio_stride, const int* __restrict__ lens, int n_fft, int hop, int n_mels,
    const float* __restrict__ window, const double* __restrict__ tw, const float* __restrict__ fb,
    float* __restrict__ out, int Fmax, unsigned* __restrict__ gmax_key) {
    extern __shared__ __attribute__((aligned(16))) double mel_sm[];    // n_fft doubles (frame) + n_bins floats (power)
    const int n_bins = n_fft / 2 + 1;
    double* xw = mel_sm;
    float* pw = reinterpret_cast<float*>(mel_sm + n_fft);
    const int b = blockIdx.y, fr = blockIdx.x, tid = threadIdx.x;
    const int len = lens[b];
    const int nfr = mel_frames(len, n_fft, hop);
    if (fr >= nfr) return;                                              // uniform
    const float* x = audio + (long)b * audio_stride;
    for (int n = tid; n < n_fft; n += 256) xw[n] = (double)(x[reflect_index(fr * hop + n - n_fft / 2, len)] * window[n]);
    __syncthreads();
    for (int k = tid; k < n_bins; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;                                                    // (k n) mod n_fft
        for (int n = 0; n < n_fft; ++n) {
            const double c = tw[2 * idx], sn = tw[2 * idx + 1];
            re = fma(xw[n], c, re);
            im = fma(xw[n], sn, im);                                    // the sign of the imaginary part does not reach |.|
            idx += k;
            if (idx >= n_fft) idx -= n_fft;
        }
        const float mag = hypotf((float)re, (float)im);
        pw[k] = mag * mag;
    }
    __syncthreads();
    float vmax = -3.0e38f;
    for (int m = tid; m < n_mels; m += 256) {
        const float* w = fb + (long)m * n_bins;
        float acc = 0.0f;
        for (int k = 0; k < n_bins; ++k) acc = fmaf(w[k], pw[k], acc);
        const float v = log10f(fmaxf(acc, 1e-10f));
        out[((long)b * n_mels + m) * Fmax + fr] = v;
        vmax = fmaxf(vmax, v);
    }
    vmax = wave_max(vmax);                                              // lanes without a mel row hold -3e38 (values masked, not lanes)
    if ((tid & 63) == 0 && vmax > -1.0e38f) atomicMax(&gmax_key[b], float_to_key(vmax));
}

__global__ __launch_bounds__(256) void mel_normalize_generic_kernel(float* __restrict__ out, const int* __restrict__ lens,
                                                                    const unsigned* __restrict__ gmax_key, int Fmax, int n_fft,
                                                                    int hop, int n_mels) {
    const int b = blockIdx.z, m = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= Fmax) return;
    float* p = out + ((long)b * n_mels + m) * Fmax + f;
    if (f < mel_frames(lens[b], n_fft, hop)) *p = (fmaxf(*p, key_to_float(gmax_key[b]) - 8.0f) + 4.0f) / 4.0f;
    else *p = 0.0f;
}

extern "C" int q3asr_log_mel_generic(const float* audio, int64_t audio_stride, const int32_t* lens_dev, int B, int Fmax,
                                     int n_fft, int hop_length, int n_mels, const float* window, const double* twiddle_cs,
                                     const float* fb, float* out, uint32_t* gmax_key, void* stream) {
    Q3_REQUIRE(audio && lens_dev && window && twiddle_cs && fb && out && gmax_key, "log_mel_generic: null pointer");
    Q3_REQUIRE(B > 0 && Fmax > 0, "log_mel_generic: B=%d Fmax=%d must be positive", B, Fmax);
    Q3_REQUIRE(n_fft >= 2 && n_fft <= 8192 && hop_length >= 1 && n_mels >= 1 && n_mels <= 65535,
               "log_mel_generic: n_fft=%d (2..8192) hop_length=%d (>= 1) n_mels=%d (1..65535)", n_fft, hop_length, n_mels);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(gmax_key, 0, sizeof(uint32_t) * B, s) != hipSuccess) return q3_check_launch("log_mel_generic memset");
    const size_t lds = sizeof(double) * n_fft + sizeof(float) * (n_fft / 2 + 1);
    if (lds > 64 * 1024)                                                 // n_fft > ~6,500: beyond the default dynamic-LDS limit (idempotent call)
        (void)hipFuncSetAttribute((const void*)mel_logpower_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mel_logpower_generic_kernel, dim3(Fmax, B), dim3(256), lds, s, audio, (long)audio_stride, lens_dev, n_fft,
                       hop_length, n_mels, window, twiddle_cs, fb, out, Fmax, gmax_key);
    hipLaunchKernelGGL(mel_normalize_generic_kernel, dim3((Fmax + 255) / 256, n_mels, B), dim3(256), 0, s, out, lens_dev, gmax_key,
                       Fmax, n_fft, hop_length, n_mels);
    return q3_check_launch("log_mel_generic");
}
