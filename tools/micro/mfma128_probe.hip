// Lane map of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit scales (gfx950), checked with exact integer data:
// assumed: lane l supplies A[row l & 15][k = 32 (l >> 4) + j] and B[k = 32 (l >> 4) + j][col l & 15] in byte j = 0..31 of its 8 VGPRs;
// result C[row = 4 (l >> 4) + r][col = l & 15] in register r (the 16x16 C/D map of every other dtype).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma128_probe.hip -o tools/micro/mfma128_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned char* A, const unsigned char* B, float* C) {     // A [16][128] row-major, B [16 cols][128] (column-major: Bt[col][k])
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8i a = *reinterpret_cast<const v8i*>(A + r * 128 + 32 * g);
    v8i b = *reinterpret_cast<const v8i*>(B + r * 128 + 32 * g);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}
static unsigned char e4m3(int v) {            // exact small integers -8..8
    static const unsigned char pos[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return v >= 0 ? pos[v] : (unsigned char)(pos[-v] | 0x80);
}
int main() {
    unsigned char hA[16 * 128], hB[16 * 128]; int iA[16][128], iB[16][128];
    srand(3);
    for (int r = 0; r < 16; ++r) for (int kk = 0; kk < 128; ++kk) {
        iA[r][kk] = rand() % 17 - 8; iB[r][kk] = rand() % 17 - 8;
        hA[r * 128 + kk] = e4m3(iA[r][kk]); hB[r * 128 + kk] = e4m3(iB[r][kk]);
    }
    unsigned char *dA, *dB; float* dC; float hC[256];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) {
        long ref = 0; for (int kk = 0; kk < 128; ++kk) ref += (long)iA[n][kk] * iB[m][kk];
        if ((float)ref != hC[n * 16 + m]) { if (bad < 5) printf("C[%d][%d] = %.1f, expected %ld\n", n, m, hC[n * 16 + m], ref); ++bad; }
    }
    printf("%s: %d of 256 outputs differ\n", bad ? "MISMATCH" : "lane map confirmed (exact integers)", bad);
    return bad != 0;
}
