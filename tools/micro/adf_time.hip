// Stage timing of attn_decode_fused_kernel at batch 1 (built with -DADF_TIMING): wall_clock64 stamps (100 MHz) of wave 0 of
// workgroup (0, 0, 0), inside a graph of 28 "layers" = [skinny4 qkv projection (writes the fp32 slabs) -> attention], each layer
// with its own weights, slabs and K/V cache, so that every launch finds its operands where the real step finds them.
// Build: hipcc --offload-arch=gfx950 -O3 -DADF_TIMING -I../../qwen3-asr-mlx_amd/csrc adf_time.hip ../../qwen3-asr-mlx_amd/csrc/core.hip -o adf_time
// Usage: ./adf_time [pos=470] [batch=1]
#include "../../qwen3-asr-mlx_amd/csrc/decode.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
extern "C" const char* q3asr_last_error(void);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static unsigned short bf(float f) { unsigned x; memcpy(&x, &f, 4); return (unsigned short)((x + 0x7FFFu + ((x >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int pos0 = argc > 1 ? atoi(argv[1]) : 470, B = argc > 2 ? atoi(argv[2]) : 1;
    const int LAYERS = 28, H = 2048, nh = 16, nkv = 8, hd = 128, Lmax = 1024, QKV = (nh + 2 * nkv) * hd, S = 2;
    const int MT = B <= 16 ? 1 : 2;
    srand(1);
    std::vector<unsigned short> hw((size_t)QKV * H), hx((size_t)(H / 32) * MT * 512), hkv((size_t)B * nkv * Lmax * hd), hn(hd);
    for (auto& v : hw) v = bf((rand() / (float)RAND_MAX - 0.5f) * 0.04f);
    for (auto& v : hx) v = bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& v : hkv) v = bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& v : hn) v = bf(1.0f);
    std::vector<float> hf(64);
    for (int i = 0; i < 64; ++i) hf[i] = powf(1e6f, -(float)i / 64.f);
    void *w[LAYERS], *kc[LAYERS], *vc[LAYERS], *x16, *out16, *qn, *kn, *ws;
    float *slab[LAYERS], *invf;
    int* pos;
    for (int l = 0; l < LAYERS; ++l) {
        CK(hipMalloc(&w[l], hw.size() * 2)); CK(hipMemcpy(w[l], hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&kc[l], hkv.size() * 2)); CK(hipMemcpy(kc[l], hkv.data(), hkv.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&vc[l], hkv.size() * 2)); CK(hipMemcpy(vc[l], hkv.data(), hkv.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&slab[l], (size_t)S * B * QKV * 4));
    }
    CK(hipMalloc(&x16, hx.size() * 2)); CK(hipMemcpy(x16, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out16, (size_t)(nh * hd / 32) * MT * 512 * 2));
    CK(hipMalloc(&qn, hd * 2)); CK(hipMemcpy(qn, hn.data(), hd * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&kn, hd * 2)); CK(hipMemcpy(kn, hn.data(), hd * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&invf, 64 * 4)); CK(hipMemcpy(invf, hf.data(), 64 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&pos, B * 4));
    std::vector<int> hp(B, pos0);
    CK(hipMemcpy(pos, hp.data(), B * 4, hipMemcpyHostToDevice));
    const int64_t wsb = q3asr_attn_decode_fused_workspace(B, nkv, Lmax);
    CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto enqueue = [&](bool with_attn, bool with_gemm) {
        for (int l = 0; l < LAYERS; ++l) {
            if (with_gemm && q3asr_skinny4_gemm(x16, B, w[l], nullptr, QKV, H, 0, 2, 4, 8, slab[l], nullptr, QKV, st)) { printf("gemm: %s\n", q3asr_last_error()); exit(1); }
            if (with_attn && q3asr_attn_decode_fused(slab[l], QKV, S, (int64_t)B * QKV, qn, kn, invf, 1e-6f, kc[l], vc[l], pos, out16, B, nh, nkv, hd, Lmax,
                                                     0.08838834764831845f, ws, st)) { printf("attn: %s\n", q3asr_last_error()); exit(1); }
        }
    };
    hipGraphExec_t ex[3];
    for (int v = 0; v < 3; ++v) {                         // 0: gemm + attention, 1: gemm only, 2: attention only
        hipGraph_t gr;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        enqueue(v != 1, v != 2);
        CK(hipStreamEndCapture(st, &gr));
        CK(hipGraphInstantiate(&ex[v], gr, nullptr, nullptr, 0));
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms[3] = {0, 0, 0};
    for (int v = 0; v < 3; ++v) {
        for (int it = 0; it < 3; ++it) CK(hipGraphLaunch(ex[v], st));
        CK(hipEventRecord(a, st));
        for (int it = 0; it < 20; ++it) CK(hipGraphLaunch(ex[v], st));
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[v], a, b));
        ms[v] = ms[v] * 1000.f / 20 / LAYERS;
    }
    printf("batch %d, context %d: per layer  gemm + attention %.2f us | gemm alone %.2f | attention alone %.2f  => attention in the chain %.2f us\n", B, pos0 + 1,
           ms[0], ms[1], ms[2], ms[0] - ms[1]);
    for (int it = 0; it < 3; ++it) {
        CK(hipGraphLaunch(ex[0], st)); CK(hipStreamSynchronize(st));
        long long t[16]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_adf_t), sizeof(t)));
        printf("  slab loads issued %.2f | pos statement passed %.2f | first key tile requested %.2f\n", (t[9] - t[0]) / 100.0, (t[10] - t[0]) / 100.0, (t[11] - t[0]) / 100.0);
        printf("stamps of the last layer's workgroup (0,0,0), wave 0, us since its entry: slabs summed %.2f | norm+RoPE %.2f | barrier %.2f | tile 1 %.2f | tile 2 %.2f | loop done %.2f | partials in LDS %.2f | merged %.2f\n",
               (t[1] - t[0]) / 100.0, (t[2] - t[0]) / 100.0, (t[3] - t[0]) / 100.0, (t[4] - t[0]) / 100.0, (t[5] - t[0]) / 100.0, (t[6] - t[0]) / 100.0, (t[7] - t[0]) / 100.0,
               (t[8] - t[0]) / 100.0);
    }
    return 0;
}
