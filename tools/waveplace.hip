// tools/waveplace.hip — diagnostic (not product): on which SIMD does wave w of a 256-thread workgroup land when 4 such workgroups
// share a CU (k_env's C2 launch shape)? If every workgroup's wave 0 sat on the same SIMD, the wave-0-only phases of k_env would
// serialise there.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256, 4) void k(unsigned* out, int spin) {
    extern __shared__ char sm[];
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
    // keep the workgroup resident for a while so that all 1024 are co-resident like k_env's tiles
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { sm[threadIdx.x] = (char)hw; }
}
int main() {
    const int nb = 1024;
    unsigned* d; hipMalloc(&d, nb * 4 * 4);
    k<<<nb, 256, 22528>>>(d, 30000);
    hipDeviceSynchronize();
    unsigned* h = (unsigned*)malloc(nb * 4 * 4); hipMemcpy(h, d, nb * 4 * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {{0}}; int distinct[5] = {0};
    for (int b = 0; b < nb; ++b) {
        int seen = 0;
        for (int w = 0; w < 4; ++w) { const int simd = (h[b * 4 + w] >> 4) & 3; hist[w][simd]++; seen |= 1 << simd; }
        distinct[__builtin_popcount(seen)]++;
    }
    for (int w = 0; w < 4; ++w) printf("wave %d of the workgroup -> SIMD0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("workgroups whose 4 waves sit on 1/2/3/4 distinct SIMDs: %d %d %d %d\n", distinct[1], distinct[2], distinct[3], distinct[4]);
    printf("first workgroups (hw_id hex per wave): ");
    for (int b = 0; b < 6; ++b) printf("[%x %x %x %x] ", h[b * 4], h[b * 4 + 1], h[b * 4 + 2], h[b * 4 + 3]);
    printf("\n");
    return 0;
}
