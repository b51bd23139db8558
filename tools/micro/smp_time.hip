// Stage timing of sample_draw_kernel (built with -DSMP_TIMING): wall_clock64 stamps of workgroup 0.
// Build: hipcc --offload-arch=gfx950 -O3 -DSMP_TIMING -I../../qwen3-asr-mlx_amd/csrc smp_time.hip ../../qwen3-asr-mlx_amd/csrc/core.hip -o smp_time
#include "../../qwen3-asr-mlx_amd/csrc/sampling.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

int main(int argc, char** argv) {
    const int B = 40, V = 151936;
    const float top_p = argc > 1 ? atof(argv[1]) : 0.9f;
    const int top_k = argc > 2 ? atoi(argv[2]) : 0;
    const float penalty = argc > 3 ? atof(argv[3]) : 1.0f;
    const int hist_n = 100, max_out = 256;
    unsigned short* h = (unsigned short*)malloc((size_t)B * V * 2);
    srand(1);
    for (size_t i = 0; i < (size_t)B * V; ++i) {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
        float g = 0.9f * sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2);
        unsigned x; memcpy(&x, &g, 4);
        h[i] = (unsigned short)((x + 0x8000u) >> 16);
    }
    void* d; int *out, *st;
    hipMalloc(&d, (size_t)B * V * 2); hipMalloc(&out, B * max_out * 4); hipMalloc(&st, 5 * B * 4);
    hipMemcpy(d, h, (size_t)B * V * 2, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) {
        hipMemset(st, 0, 5 * B * 4);
        int mx[B], ng[B]; for (int i = 0; i < B; ++i) { mx[i] = 250; ng[i] = hist_n; }
        hipMemcpy(st + 4 * B, mx, B * 4, hipMemcpyHostToDevice);
        hipMemcpy(st, ng, B * 4, hipMemcpyHostToDevice);
        int* ho = (int*)malloc(B * max_out * 4);
        for (int i = 0; i < B * max_out; ++i) ho[i] = rand() % V;
        hipMemcpy(out, ho, B * max_out * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        int rc = q3asr_sample_step(d, V, B, V, out, max_out, st, st + B, st + 2 * B, st + 3 * B, st + 4 * B, 0.7f, top_p, top_k, penalty, 100, 5 + it, -1, -1, 0, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long t[32]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_smp_t), sizeof(t));
        printf("rc=%d  %.1f us total;  stamps (us since start, 100 MHz clock):", rc, ms * 1000);
        for (int i = 1; i < 19; ++i) printf(" [%d]%.1f", i, (t[i] - t[0]) / 100.0);
        printf("\n");
    }
    return 0;
}
