/* A consumer of include/gmpe.h written in plain C: no Python, no torch — what a binding in another host language (cgo, JNI, N-API) would do.
 * It reads a gmpe_config POD from a file (python: open(path, "wb").write(bytes(gmpe.make_config(...))) — the struct is plain data), allocates the outputs with
 * the HIP runtime, resets, runs `steps` closed-loop steps with a fixed action pattern (action of env n, agent a at step k = (7 n + 3 a + k) % n_actions) and then
 * after a second reset the same number of steps as ONE launch of the rollout kernel, and prints checksums that tests/test_gpu_c_consumer.py compares with the Python engine's.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_abi_consumer.c -o c_abi_consumer \
 *       -Lcontracts-marl-aam-corridors_amd -lgmpe -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/contracts-marl-aam-corridors_amd -Wl,-rpath,/opt/rocm/lib
 *   ./c_abi_consumer cfg.bin 6
 *
 * Reference side being replaced: one worker process per env stepping MultiAgentGraphEnv (onpolicy/envs/env_wrappers.py:843-905, 959-1037). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "gmpe.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define GMPE_OK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, gmpe_last_error()); return 3; } } while (0)

static double sum_f32(const float* dev, size_t n) {
    float* h = (float*)malloc(n * sizeof(float));
    double s = 0;
    if (hipMemcpy(h, dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { free(h); return -1e300; }
    for (size_t q = 0; q < n; ++q) s += (double)h[q];
    free(h);
    return s;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <config.bin> <steps>\n", argv[0]); return 1; }
    gmpe_config cfg;
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(&cfg, sizeof cfg, 1, f) != 1) { fprintf(stderr, "cannot read a gmpe_config (%zu bytes) from %s\n", sizeof cfg, argv[1]); return 1; }
    fclose(f);
    const int K = atoi(argv[2]);
    if (gmpe_abi_version() != GMPE_ABI_VERSION || cfg.abi_version != GMPE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    const size_t N = (size_t)cfg.num_envs, A = (size_t)cfg.num_agents, E = (size_t)gmpe_num_entities(&cfg), D = (size_t)gmpe_obs_dim(&cfg), F = (size_t)gmpe_node_feats(&cfg);

    gmpe_handle* h = NULL;
    GMPE_OK(gmpe_create(&cfg, 0, &h));
    HIP_OK(hipSetDevice(0));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    gmpe_outputs out;
    memset(&out, 0, sizeof out);
    HIP_OK(hipMalloc((void**)&out.obs, N * A * D * 4));
    HIP_OK(hipMalloc((void**)&out.agent_id, N * A * 4));
    HIP_OK(hipMalloc((void**)&out.node_obs, N * A * E * F * 4));
    HIP_OK(hipMalloc((void**)&out.adj, N * E * E * 4));                  /* one E x E matrix per env: adj_compact */
    HIP_OK(hipMalloc((void**)&out.reward, N * A * 4));
    HIP_OK(hipMalloc((void**)&out.done, N * A));
    out.adj_compact = 1;

    /* K action sets, resident on the device */
    int32_t* acts_h = (int32_t*)malloc((size_t)K * N * A * 4);
    for (int k = 0; k < K; ++k)
        for (size_t n = 0; n < N; ++n)
            for (size_t a = 0; a < A; ++a) acts_h[((size_t)k * N + n) * A + a] = (int32_t)((7 * n + 3 * a + (size_t)k) % (size_t)cfg.n_actions);
    int32_t* acts = NULL;
    HIP_OK(hipMalloc((void**)&acts, (size_t)K * N * A * 4));
    HIP_OK(hipMemcpy(acts, acts_h, (size_t)K * N * A * 4, hipMemcpyHostToDevice));

    /* closed loop: one launch per step */
    GMPE_OK(gmpe_reset(h, NULL, &out, st));
    HIP_OK(hipStreamSynchronize(st));
    printf("reset obs %.9e node %.9e adj %.9e\n", sum_f32(out.obs, N * A * D), sum_f32(out.node_obs, N * A * E * F), sum_f32(out.adj, N * E * E));
    double rsum = 0;
    for (int k = 0; k < K; ++k) {
        GMPE_OK(gmpe_step(h, acts + (size_t)k * N * A, &out, st));
        HIP_OK(hipStreamSynchronize(st));
        rsum += sum_f32(out.reward, N * A);
    }
    printf("loop obs %.9e node %.9e adj %.9e reward %.9e\n", sum_f32(out.obs, N * A * D), sum_f32(out.node_obs, N * A * E * F), sum_f32(out.adj, N * E * E), rsum);

    /* a second episode (the env streams continue from their counters): K steps as ONE launch of the persistent rollout kernel, every step overwriting the same buffers */
    GMPE_OK(gmpe_reset(h, NULL, &out, st));
    gmpe_rollout plan;
    memset(&plan, 0, sizeof plan);
    plan.num_steps = K; plan.num_action_sets = K; plan.num_slots = 1; plan.first_slot = 0;
    GMPE_OK(gmpe_rollout_steps(h, acts, &plan, &out, st));
    HIP_OK(hipStreamSynchronize(st));
    printf("rollout obs %.9e node %.9e adj %.9e\n", sum_f32(out.obs, N * A * D), sum_f32(out.node_obs, N * A * E * F), sum_f32(out.adj, N * E * E));

    gmpe_tuning t;
    GMPE_OK(gmpe_get_tuning(h, &t));
    printf("tuning G %d block %d roll %d\n", t.G, t.block, t.roll);
    GMPE_OK(gmpe_destroy(h));
    hipFree(out.obs); hipFree(out.agent_id); hipFree(out.node_obs); hipFree(out.adj); hipFree(out.reward); hipFree(out.done); hipFree(acts);
    free(acts_h);
    return 0;
}
