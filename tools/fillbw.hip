// tools/fillbw.hip — diagnostic (not product): what a pure streaming-store / copy kernel achieves on this GPU, to put
// roofline.frac (priced against the 8 TB/s spec peak) next to the write bandwidth the part really delivers.
// build: hipcc -O3 --offload-arch=gfx950 tools/fillbw.hip -o gpurun_out/fillbw    run: gpurun_out/fillbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void fill4(float4* __restrict__ dst, size_t n4, float v) {
    const float4 x = make_float4(v, v, v, v);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = x;
}
__global__ void fill4_nt(float4* __restrict__ dst, size_t n4, float v) {
    const v4f x = {v, v, v, v};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(x, reinterpret_cast<v4f*>(dst) + i);
}
__global__ void copy4(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {98ull << 20, 1ull << 30, 6ull << 30};
    for (size_t bytes : sizes) {
        float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        const size_t n4 = bytes / 16;
        for (int grid : {2048, 8192, 65536}) {
            for (int mode = 0; mode < 3; ++mode) {
                const int reps = bytes > (1ull << 30) ? 5 : 20;
                for (int w = 0; w < 2; ++w) { if (mode == 0) fill4<<<grid, 256>>>(a, n4, 1.f); else if (mode == 1) fill4_nt<<<grid, 256>>>(a, n4, 1.f); else copy4<<<grid, 256>>>(a, b, n4); }
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) { if (mode == 0) fill4<<<grid, 256>>>(a, n4, 1.f); else if (mode == 1) fill4_nt<<<grid, 256>>>(a, n4, 1.f); else copy4<<<grid, 256>>>(a, b, n4); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double per = ms / reps * 1e-3;
                printf("%-8s bytes %6.0f MB grid %6d: %8.1f us  write %.2f TB/s%s\n", mode == 0 ? "fill" : mode == 1 ? "fill_nt" : "copy", bytes / 1048576.0, grid,
                       per * 1e6, bytes / per / 1e12, mode == 2 ? " (+ equal read)" : "");
            }
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
