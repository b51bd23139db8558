// Micro-benchmark: what an empty launch costs inside a hipGraph chain, by grid size, workgroup size and dynamic LDS size.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/empty.hip -o tools/micro/empty
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(1024) void empty_kernel(int* sink) {
    extern __shared__ int lds[];
    if (sink && threadIdx.x == 9999) *sink = lds[0];
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int L = 56;
    for (int threads : {64, 256, 512, 1024})
        for (int grid : {32, 256, 512, 1024})
            for (int lds : {0, 32 * 1024, 68 * 1024, 132 * 1024}) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int l = 0; l < L; ++l) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), lds, st, (int*)nullptr);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                std::vector<float> ts;
                for (int r = 0; r < 8; ++r) {
                    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r) ts.push_back(ms * 1000.f / L);
                }
                std::sort(ts.begin(), ts.end());
                printf("threads %4d grid %4d lds %6d: %5.2f us per launch\n", threads, grid, lds, ts[ts.size() / 2]);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
    return 0;
}
