// tools/launchfloor.hip — diagnostic (not product): back-to-back launch period of an EMPTY kernel with k_env's launch shape
// (683 workgroups x 256 threads, 40 KiB dynamic LDS) for a 1.1 KB by-value kernarg vs an 8-byte one, plain launches vs a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { double pad[140]; int* out; };
__global__ void k_big(const Big p) { extern __shared__ char sm[]; if (threadIdx.x == 1023) { sm[0] = 1; p.out[blockIdx.x] = sm[0] + (int)p.pad[threadIdx.x & 127]; } }
__global__ void k_small(int* out) { extern __shared__ char sm[]; if (threadIdx.x == 1023) { sm[0] = 1; out[blockIdx.x] = sm[0]; } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    int* out; CK(hipMalloc(&out, 1 << 20));
    Big b; for (int i = 0; i < 140; ++i) b.pad[i] = i; b.out = out;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 2000;
    for (int lds : {0, 40960}) for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 50; ++w) { if (mode) k_big<<<683, 256, lds, st>>>(b); else k_small<<<683, 256, lds, st>>>(out); }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < K; ++r) { if (mode) k_big<<<683, 256, lds, st>>>(b); else k_small<<<683, 256, lds, st>>>(out); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("plain  launches, kernarg %4zu B, LDS %5d: %.2f us per launch\n", mode ? sizeof(Big) : sizeof(int*), lds, ms / K * 1e3);
    }
    // hipGraph of 100 launches, replayed 20 times
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int r = 0; r < 100; ++r) k_big<<<683, 256, 40960, st>>>(b);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipGraph (100 kernels), kernarg %zu B, LDS 40960: %.2f us per launch\n", sizeof(Big), ms / 2000 * 1e3);
    return 0;
}
