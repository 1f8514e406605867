// tools/tilebw.hip — diagnostic (not product): what store rate does the ROLLOUT kernel's store geometry reach on its own? 1024 persistent workgroups
// (4 per CU) x W streaming waves, each workgroup writing its tile's contiguous blocks (adj 64 KB + node 25.6 KB + obs 3 KB at c2) once per "step" into
// slot (step % T) of [T, ...] storage, nontemporal or ordinary stores, optionally with a busy gap per step that stands for wave 0's chain.
// build: hipcc -O3 --offload-arch=gfx950 tools/tilebw.hip -o tools/tilebw.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void tile_store(v4f* __restrict__ adj, v4f* __restrict__ node, v4f* __restrict__ obs, int K, int T, size_t adj_slot4, size_t node_slot4,
                                                  size_t obs_slot4, int adj4, int node4, int obs4, int streamers, long long gap, int nosync = 0) {
    const int tid = threadIdx.x, w = tid >> 6;
    const v4f val = {1.f, 2.f, 3.f, (float)tid};
    const int first = 256 - streamers * 64;                            // the LAST `streamers` waves store
    for (int k = 0; k < K; ++k) {
        const int s = k % T;
        if (tid >= first) {
            const int l = tid - first, nl = streamers * 64;
            v4f* a = adj + (size_t)s * adj_slot4 + (size_t)blockIdx.x * adj4;
            for (int q = l; q < adj4; q += nl) { if (NT) __builtin_nontemporal_store(val, a + q); else a[q] = val; }
            v4f* n = node + (size_t)s * node_slot4 + (size_t)blockIdx.x * node4;
            for (int q = l; q < node4; q += nl) { if (NT) __builtin_nontemporal_store(val, n + q); else n[q] = val; }
            v4f* o = obs + (size_t)s * obs_slot4 + (size_t)blockIdx.x * obs4;
            for (int q = l; q < obs4; q += nl) { if (NT) __builtin_nontemporal_store(val, o + q); else o[q] = val; }
        } else if (w == 0 && gap > 0) {
            const long long t0 = __builtin_readcyclecounter();
            while ((long long)__builtin_readcyclecounter() - t0 < gap) __builtin_amdgcn_s_sleep(8);
        }
        if (!nosync) __syncthreads();                                     // nosync: every wave free-runs through the K steps (what a tile whose streaming waves are decoupled from its step barrier could reach)
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// With arguments — tilebw.bin <label> <tiles> <adj float4 per tile> <node float4 per tile> <obs float4 per tile> <T slots> <K steps> — it measures ONE geometry (streamers 3 / 4,
// plain / nontemporal, no gap) and prints a JSON line with the best rate: the store ceiling bench.py prices `frac_of_measured_fill` against (tools/fillbw_r04.sh -> profiles/r04_fillbw.json).
static int g_nosync = 0;
static int one_geometry(const char* label, int tiles, int adj4, int node4, int obs4, int T, int K) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double step_bytes = (double)tiles * (adj4 + node4 + obs4) * 16;
    v4f *adj, *node, *obs;
    CK(hipMalloc(&adj, (size_t)T * tiles * adj4 * 16)); CK(hipMalloc(&node, (size_t)T * tiles * node4 * 16)); CK(hipMalloc(&obs, (size_t)T * tiles * obs4 * 16));
    double best = 0; int best_s = 0, best_nt = 0;
    for (int streamers : {3, 4}) for (int nt = 0; nt < 2; ++nt) {
        auto launch = [&] {
            if (nt) tile_store<true><<<tiles, 256>>>(adj, node, obs, K, T, (size_t)tiles * adj4, (size_t)tiles * node4, (size_t)tiles * obs4, adj4, node4, obs4, streamers, 0, g_nosync);
            else tile_store<false><<<tiles, 256>>>(adj, node, obs, K, T, (size_t)tiles * adj4, (size_t)tiles * node4, (size_t)tiles * obs4, adj4, node4, obs4, streamers, 0, g_nosync);
        };
        launch(); CK(hipDeviceSynchronize());
        double ms_best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < ms_best) ms_best = ms;
        }
        const double tbps = step_bytes / (ms_best / K * 1e-3) / 1e12;
        fprintf(stderr, "%s T=%d streamers %d %s: %.2f us per step %.2f TB/s\n", label, T, streamers, nt ? "nt" : "plain", ms_best / K * 1e3, tbps);
        if (tbps > best) { best = tbps; best_s = streamers; best_nt = nt; }
    }
    printf("{\"geometry\": \"%s\", \"tiles\": %d, \"slots\": %d, \"step_MB\": %.1f, \"pass_GB\": %.3f, \"best_GBps\": %.0f, \"best_streaming_waves\": %d, \"best_nontemporal\": %d}\n",
           label, tiles, T, step_bytes / 1e6, step_bytes * T / 1e9, best * 1e3, best_s, best_nt);
    CK(hipFree(adj)); CK(hipFree(node)); CK(hipFree(obs));
    return 0;
}
int main(int argc, char** argv) {
    if (argc == 9) { g_nosync = atoi(argv[8]); argc = 8; }                   // optional 8th argument: 1 = no per-step workgroup barrier
    if (argc == 8) return one_geometry(argv[1], atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));

    const int tiles = 1024, adj4 = 4 * 10 * 400 / 4, node4 = 4 * 10 * 20 * 8 / 4, obs4 = 4 * 10 * 13 / 4 + 2;   // c2: 4 envs per tile
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double step_bytes = (double)tiles * (adj4 + node4 + obs4) * 16;
    printf("step bytes %.1f MB\n", step_bytes / 1e6);
    for (int T : {1, 26}) {
        v4f *adj, *node, *obs;
        CK(hipMalloc(&adj, (size_t)T * tiles * adj4 * 16)); CK(hipMalloc(&node, (size_t)T * tiles * node4 * 16)); CK(hipMalloc(&obs, (size_t)T * tiles * obs4 * 16));
        for (int streamers : {3, 4}) for (int nt = 0; nt < 2; ++nt) for (long long gap : {0LL, 100LL, 400LL, 800LL, 1200LL}) {
            const int K = 520;
            auto launch = [&] {
                if (nt) tile_store<true><<<tiles, 256>>>(adj, node, obs, K, T, (size_t)tiles * adj4, (size_t)tiles * node4, (size_t)tiles * obs4, adj4, node4, obs4, streamers, gap);
                else tile_store<false><<<tiles, 256>>>(adj, node, obs, K, T, (size_t)tiles * adj4, (size_t)tiles * node4, (size_t)tiles * obs4, adj4, node4, obs4, streamers, gap);
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("T=%2d streamers %d %s gap %5lld ticks: %6.2f us per step  %.2f TB/s\n", T, streamers, nt ? "nt   " : "plain", gap, ms / K * 1e3, step_bytes / (ms / K * 1e-3) / 1e12);
        }
        CK(hipFree(adj)); CK(hipFree(node)); CK(hipFree(obs));
    }
    return 0;
}
