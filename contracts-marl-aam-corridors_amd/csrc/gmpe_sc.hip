// gmpe_sc.hip — one scenario variant of the fused kernel per translation unit (compiled with -DGMPE_SC=<variant>, in
// parallel): tile shapes BLOCK in {64,128,256} x exact-size instantiations AP in {0,3,10}, plus the steady-state instantiation
// <256, 10, SC, FL = 1> (run-time flags folded) that the C2/C3-shaped workloads run, plus the rollout instantiations
// <64, 0, SC, 2>, <256, {0, 10}, SC, 2> (FL = 2: K steps inside one launch, gmpe_rollout_steps), and the compile-time-G variants of the 4096 x 10 tile shapes:
// <256, 10, SC, 2, GC = 4 | 6> and <256, 10, SC, 1, GC = 4>.
#include "gmpe_kernel.h"

#ifndef GMPE_SC
#error "compile with -DGMPE_SC=<scenario variant>"
#endif

#ifndef GMPE_PART
#define GMPE_PART 0
#endif

namespace gmpe {

// The two steady-state step kernels (FL = 1: one launch per step, the closed-loop shape) live in a translation unit of their own (-DGMPE_PART=1) that is compiled with the max-ILP
// machine scheduler (-mllvm -amdgpu-sched-strategy=max-ilp): a step launch is one dependent chain per tile, and scheduling for ILP instead of occupancy shortens it (c2 closed loop
// 24.1 -> 23.6 us per step, c3 24.5 -> 24.2) while the rollout kernels, which interleave chains of several tiles, are 0.5 % faster with the default strategy (profiles/r04_notes.md).
#if GMPE_PART == 1
template __global__ void k_env<256, 10, GMPE_SC, 1>(const KParams);
template __global__ void k_env<256, 10, GMPE_SC, 1, 4>(const KParams);
#else
extern template __global__ void k_env<256, 10, GMPE_SC, 1>(const KParams);
extern template __global__ void k_env<256, 10, GMPE_SC, 1, 4>(const KParams);

template <int SC>
void launch_env(int block, int ap, int fl, dim3 grid, size_t lds, hipStream_t st, const KParams& p) {
    if (fl == 2) {
        // persistent rollout kernel: BLOCK 64 (run-time sizes) or 256 (run-time sizes, or exact-size for A = L = 10). Register budgets
        // (__launch_bounds__ in gmpe_kernel.h), measured per scenario on one box (profiles/README.md): navigation_graph exact-size at four
        // tiles per CU (G = 4; spills 5 dwords, 17.0 us per step vs 18.3 at three tiles / G = 6 without a spill, 19.9 run-time sizes); the
        // kinematic scenarios exact-size at three tiles per CU (G = 6, no spill: July 15.8 us vs 18.0 at four tiles with a 14-dword spill).
        if (block == 64) hipLaunchKernelGGL((k_env<64, 0, SC, 2>), grid, dim3(64), lds, st, p);
        else if (ap == 10 && p.G == 4) hipLaunchKernelGGL((k_env<256, 10, SC, 2, 4>), grid, dim3(256), lds, st, p);      // envs per tile folded too (the 4096 x 10 shapes)
        else if (ap == 10 && p.G == 6) hipLaunchKernelGGL((k_env<256, 10, SC, 2, 6>), grid, dim3(256), lds, st, p);
        else if (ap == 10) hipLaunchKernelGGL((k_env<256, 10, SC, 2>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((k_env<256, 0, SC, 2>), grid, dim3(256), lds, st, p);
        return;
    }
    if (fl && block == 256 && ap == 10) {
        if (p.G == 4) hipLaunchKernelGGL((k_env<256, 10, SC, 1, 4>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((k_env<256, 10, SC, 1>), grid, dim3(256), lds, st, p);
        return;
    }
#define LAUNCH_ENV(B) do { \
        if (ap == 10) hipLaunchKernelGGL((k_env<B, 10, SC, 0>), grid, dim3(B), lds, st, p); \
        else if (ap == 3) hipLaunchKernelGGL((k_env<B, 3, SC, 0>), grid, dim3(B), lds, st, p); \
        else hipLaunchKernelGGL((k_env<B, 0, SC, 0>), grid, dim3(B), lds, st, p); \
    } while (0)
    switch (block) {
        case 64: LAUNCH_ENV(64); break;
        case 128: LAUNCH_ENV(128); break;
        default: LAUNCH_ENV(256); break;
    }
#undef LAUNCH_ENV
}

// opt in to > 64 KiB of dynamic LDS (gfx950: 160 KiB per CU)
template <int SC>
hipError_t set_max_lds(int lds) {
#define FN(B, P) reinterpret_cast<const void*>(&k_env<B, P, SC, 0>)
    const void* fns[16] = {FN(64, 0), FN(64, 3), FN(64, 10), FN(128, 0), FN(128, 3), FN(128, 10), FN(256, 0), FN(256, 3), FN(256, 10),
                           reinterpret_cast<const void*>(&k_env<256, 10, SC, 1>),
                           reinterpret_cast<const void*>(&k_env<64, 0, SC, 2>), reinterpret_cast<const void*>(&k_env<256, 0, SC, 2>),
                           reinterpret_cast<const void*>(&k_env<256, 10, SC, 2>),
                           reinterpret_cast<const void*>(&k_env<256, 10, SC, 2, 4>), reinterpret_cast<const void*>(&k_env<256, 10, SC, 2, 6>),
                           reinterpret_cast<const void*>(&k_env<256, 10, SC, 1, 4>)};
#undef FN
    hipError_t e = hipSuccess;
    for (int q = 0; q < 16 && e == hipSuccess; ++q) e = hipFuncSetAttribute(fns[q], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
}

// resident workgroups per CU of the instantiation launch_env would pick (registers + LDS), 0 on error
template <int SC>
int max_tiles_per_cu(int block, int ap, size_t lds, int roll, int g) {
#define FN(B, P) reinterpret_cast<const void*>(&k_env<B, P, SC, 0>)
#define PICK(B) (ap == 10 ? FN(B, 10) : (ap == 3 ? FN(B, 3) : FN(B, 0)))
    if (roll) block = block == 64 ? 64 : 256;
    const void* roll10 = g == 4 ? reinterpret_cast<const void*>(&k_env<256, 10, SC, 2, 4>) : (g == 6 ? reinterpret_cast<const void*>(&k_env<256, 10, SC, 2, 6>) : reinterpret_cast<const void*>(&k_env<256, 10, SC, 2>));
    const void* step10 = g == 4 ? reinterpret_cast<const void*>(&k_env<256, 10, SC, 1, 4>) : reinterpret_cast<const void*>(&k_env<256, 10, SC, 1>);
    const void* fn = roll ? (block == 64 ? reinterpret_cast<const void*>(&k_env<64, 0, SC, 2>)
                                         : (ap == 10 ? roll10 : reinterpret_cast<const void*>(&k_env<256, 0, SC, 2>)))
                          : block == 64 ? PICK(64) : (block == 128 ? PICK(128) : (ap == 10 ? step10 : PICK(256)));
#undef PICK
#undef FN
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb;
}

template void launch_env<GMPE_SC>(int, int, int, dim3, size_t, hipStream_t, const KParams&);
template int max_tiles_per_cu<GMPE_SC>(int, int, size_t, int, int);
template hipError_t set_max_lds<GMPE_SC>(int);
#endif  // GMPE_PART

}  // namespace gmpe
