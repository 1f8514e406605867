// tools/expandbw.hip — diagnostic (not product): store patterns for the big-E adjacency expansion [N,E,E] -> [N,A,E,E]
// (k_adj_expand in gmpe_step.hip), timed next to a plain fill of the same bytes. Picks the pattern the library ships.
// build: hipcc -O3 --offload-arch=gfx950 tools/expandbw.hip -o gpurun_out/expandbw    run: gpurun_out/expandbw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));

// A: one lane per SOURCE float4, A strided stores (copies are nq float4 apart)
template <int U, bool NT>
__global__ __launch_bounds__(256) void expandA(const v4f* __restrict__ src, v4f* __restrict__ dst, uint32_t total4, uint32_t nq, int A) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total4) return;
    const long long n = t / nq; const int m = (int)(t - (uint32_t)n * nq);
    const v4f val = src[t];
    v4f* d = dst + n * (long long)A * nq + m;
    int a = 0;
    for (; a + U <= A; a += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(val, d + (long long)(a + u) * nq); else d[(long long)(a + u) * nq] = val; }
    }
    for (; a < A; ++a) { if (NT) __builtin_nontemporal_store(val, d + (long long)a * nq); else d[(long long)a * nq] = val; }
}
// B: output-contiguous. blockIdx.x covers the A*nq float4 of ONE env's output, blockIdx.y strides over envs; a wave's store is 1 KB
// contiguous and consecutive workgroups write consecutive 4 KB, like a fill. The source float4 is re-read per copy (L2 hits).
template <bool NT>
__global__ __launch_bounds__(256) void expandB(const v4f* __restrict__ src, v4f* __restrict__ dst, int N, uint32_t nq, int A, uint32_t m_nq) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x, per = (uint32_t)A * nq;
    if (j >= per) return;
    const uint32_t a = __umulhi(j, m_nq), m = j - a * nq;              // exact: j * nq < 2^32
    (void)a;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const v4f val = src[(size_t)n * nq + m];
        if (NT) __builtin_nontemporal_store(val, dst + (size_t)n * per + j); else dst[(size_t)n * per + j] = val;
    }
}
// C: like B but each lane keeps its source float4 for UA consecutive copies of the same env (fewer loads): lane -> (m), loop a in a chunk
template <int UA, bool NT>
__global__ __launch_bounds__(256) void expandC(const v4f* __restrict__ src, v4f* __restrict__ dst, int N, uint32_t nq, int A) {
    // blockIdx.x: chunk of 256 float4 inside the matrix; blockIdx.y: env stride; blockIdx.z: group of UA copies
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= nq) return;
    const int a0 = blockIdx.z * UA;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const v4f val = src[(size_t)n * nq + m];
        v4f* d = dst + ((size_t)n * A + a0) * nq + m;
#pragma unroll
        for (int u = 0; u < UA; ++u) if (a0 + u < A) { if (NT) __builtin_nontemporal_store(val, d + (size_t)u * nq); else d[(size_t)u * nq] = val; }
    }
}
// X: like B, but XCD-aware: workgroups are dealt to the 8 XCDs round-robin by linear id, so workgroup L works for XCD L % 8; all workgroups
// of one env are given to ONE XCD (env = 8 * (L / 8 / Bx) + L % 8), whose L2 then fetches the env's source matrix once instead of up to 8 times.
template <bool NT>
__global__ __launch_bounds__(256) void expandX(const v4f* __restrict__ src, v4f* __restrict__ dst, int N, uint32_t nq, int A, uint32_t Bx) {
    const uint32_t L = blockIdx.x, xcd = L & 7u, idx = L >> 3;
    const uint32_t el = idx / Bx, jb = idx - el * Bx;
    const uint32_t n = el * 8u + xcd;
    const uint32_t j = jb * 256u + threadIdx.x, per = (uint32_t)A * nq;
    if (n >= (uint32_t)N || j >= per) return;
    const uint32_t m = j % nq;
    const v4f val = src[(size_t)n * nq + m];
    if (NT) __builtin_nontemporal_store(val, dst + (size_t)n * per + j); else dst[(size_t)n * per + j] = val;
}
// W: no A-fold re-read of the source — a workgroup owns one env (or 1/P of its matrix), loads its R float4 per thread ONCE into
// registers and streams the A copies in output order (per copy a contiguous nq/P * 16 B block per workgroup).
template <int R, bool NT>
__global__ __launch_bounds__(256) void expandW(const v4f* __restrict__ src, v4f* __restrict__ dst, int N, uint32_t nq, int A, int P) {
    const uint32_t n = blockIdx.x / P, part = blockIdx.x - n * P, per_part = nq / P;
    if (n >= (uint32_t)N) return;
    v4f val[R];
    const uint32_t m0 = part * per_part + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) { const uint32_t m = m0 + r * 256u; val[r] = (r * 256u + threadIdx.x < per_part) ? src[(size_t)n * nq + m] : v4f{0, 0, 0, 0}; }
    v4f* d = dst + (size_t)n * A * nq + m0;
    for (int a = 0; a < A; ++a) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r * 256u + threadIdx.x < per_part) { if (NT) __builtin_nontemporal_store(val[r], d + (size_t)a * nq + r * 256u); else d[(size_t)a * nq + r * 256u] = val[r]; }
    }
}
__global__ void fill4(v4f* __restrict__ dst, size_t n4, float v) {
    const v4f x = {v, v, v, v};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(x, dst + i);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(0x100000000ull / d) + 1u; }

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Shape { const char* name; int N, A, E; } shapes[] = {{"c5 chunk 256x64x128", 256, 64, 128}, {"c5 shard 2048x64x128", 2048, 64, 128}, {"c5 8192x64x128 (src 512 MB)", 8192, 64, 128}, {"c4 8192x32x72", 8192, 32, 72}, {"c4 chunk 745x32x72", 745, 32, 72}};
    for (const Shape& s : shapes) {
        const uint32_t nq = (uint32_t)s.E * s.E / 4, total4 = (uint32_t)s.N * nq;
        const size_t out4 = (size_t)total4 * s.A;
        v4f *src, *dst; CK(hipMalloc(&src, (size_t)total4 * 16)); CK(hipMalloc(&dst, out4 * 16));
        CK(hipMemset(src, 0, (size_t)total4 * 16));
        const int reps = out4 * 16 > (1ull << 30) ? 6 : 30;
        auto run = [&](const char* label, auto launch) -> int {
            for (int w = 0; w < 2; ++w) launch();
            CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double per = ms / reps * 1e-3;
            printf("%-22s %-28s %9.1f us  write %.2f TB/s\n", s.name, label, per * 1e6, out4 * 16.0 / per / 1e12);
            return 0;
        };
        run("fill_nt grid 65536", [&] { fill4<<<65536, 256>>>(dst, out4, 1.f); });
        { const uint32_t Bx = ((uint32_t)s.A * nq + 255) / 256; const uint32_t gX = 8u * ((s.N + 7) / 8) * Bx;
          run("X nt (XCD-aware)", [&] { expandX<true><<<gX, 256>>>(src, dst, s.N, nq, s.A, Bx); });
          run("X plain (XCD-aware)", [&] { expandX<false><<<gX, 256>>>(src, dst, s.N, nq, s.A, Bx); }); }
        run("A U=4 nt", [&] { expandA<4, true><<<(total4 + 255) / 256, 256>>>(src, dst, total4, nq, s.A); });
        run("A U=8 nt", [&] { expandA<8, true><<<(total4 + 255) / 256, 256>>>(src, dst, total4, nq, s.A); });
        run("A U=4 plain", [&] { expandA<4, false><<<(total4 + 255) / 256, 256>>>(src, dst, total4, nq, s.A); });
        for (int P : {1, 2, 4, 8}) {                                      // W: registers hold the matrix part, no re-read
            char lab[64];
            const uint32_t per_part = nq / P;
            if (nq % P) continue;
            const int R = (int)((per_part + 255) / 256);
            snprintf(lab, sizeof lab, "W nt P=%d (R=%d)", P, R);
            const dim3 gW((uint32_t)s.N * P);
            if (R <= 2) run(lab, [&] { expandW<2, true><<<gW, 256>>>(src, dst, s.N, nq, s.A, P); });
            else if (R <= 4) run(lab, [&] { expandW<4, true><<<gW, 256>>>(src, dst, s.N, nq, s.A, P); });
            else if (R <= 8) run(lab, [&] { expandW<8, true><<<gW, 256>>>(src, dst, s.N, nq, s.A, P); });
            else if (R <= 16) run(lab, [&] { expandW<16, true><<<gW, 256>>>(src, dst, s.N, nq, s.A, P); });
        }
        for (int gy : {256, 1024}) {
            char lab[64];
            const dim3 gB(((uint32_t)s.A * nq + 255) / 256, gy < s.N ? gy : s.N);
            snprintf(lab, sizeof lab, "B nt gy=%d", gy);
            run(lab, [&] { expandB<true><<<gB, 256>>>(src, dst, s.N, nq, s.A, magic_of(nq)); });
            snprintf(lab, sizeof lab, "B plain gy=%d", gy);
            run(lab, [&] { expandB<false><<<gB, 256>>>(src, dst, s.N, nq, s.A, magic_of(nq)); });
            const dim3 gC((nq + 255) / 256, gy < s.N ? gy : s.N, (s.A + 3) / 4);
            snprintf(lab, sizeof lab, "C UA=4 nt gy=%d", gy);
            run(lab, [&] { expandC<4, true><<<gC, 256>>>(src, dst, s.N, nq, s.A); });
            const dim3 gC8((nq + 255) / 256, gy < s.N ? gy : s.N, (s.A + 7) / 8);
            snprintf(lab, sizeof lab, "C UA=8 nt gy=%d", gy);
            run(lab, [&] { expandC<8, true><<<gC8, 256>>>(src, dst, s.N, nq, s.A); });
        }
        CK(hipFree(src)); CK(hipFree(dst));
    }
    return 0;
}
