// gmpe_kernel.h — the fused GraphMPE step / reset kernel template (see gmpe_step.hip for the overview).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "gmpe_device.h"

namespace gmpe {

enum { MODE_STEP = 0, MODE_RESET = 1 };
// Scenario variant = compile-time parameter of the kernel: each variant's observation / reward / reset code is compiled
// into its own instantiation (gmpe_sc.hip, one translation unit per variant), so adding a scenario never costs the
// others registers. navigation_graph has a wall-less variant: the wall contact code is the largest register consumer.
enum { SC_NAV = 0, SC_NAV_WALLS = 1, SC_JULY = 2, SC_ROT = 3, SC_TWO = 4, SC_THREE = 5, SC_COUNT = 6 };
__host__ __device__ constexpr bool sc_kinematic(int sc) { return sc >= SC_JULY; }
__host__ __device__ constexpr bool sc_rotfam(int sc) { return sc >= SC_ROT; }                     // rotated-frame float32 features, F = 7
__host__ __device__ constexpr bool sc_phasefam(int sc) { return sc == SC_TWO || sc == SC_THREE; }  // D = 15, random tube length

struct KParams {
    gmpe_config c;
    DevState s;
    gmpe_outputs o;
    const int32_t* act;       // [N,A] or nullptr
    const float* onehot;      // [N,A,n_actions] or nullptr
    const uint8_t* mask;      // reset mask or nullptr
    const double* ovr;        // safety-filter hook slot: [N,A,2] controls integrated instead of the decoded action, or nullptr
    const uint8_t* ovr_use;   // [N,A] where to use them (nullptr: everywhere)
    int mode;
    int env_lo, env_hi;       // env range of this launch [lo, hi): the whole batch, or one chunk of the split big-E pipeline
    int A, L, O, E, D, F;     // F = node features per row (8, rot_inv: 7)
    int G;                    // envs per workgroup (G*A <= 64)
    int spec;                 // 1: multi-wave tiles specialise (wave 0 reward/info, waves 1.. graph stores)
    int rowpairs;             // 1: rollouts of exact-size tiles run distance_force_pass (agent-row pairs only, landmark block cached, navigation_graph: + next-step forces)
    int nfuse;                // > 0: doubles per env of the separate pair-force buffer (2*A*A) — rollouts of exact-size navigation_graph tiles compute the NEXT
                              // step's contact forces inside the distance pass (gmpe_create decides; 0: the force pass aliases the fp32 matrix)
    int nt;                   // 1: nontemporal graph stores (outputs per launch exceed the 256 MiB Infinity Cache)
    int ablate;               // diagnostic build only (-DGMPE_DIAG): timing-only ablations; the shipped library ignores it
    // rollout (FL == 2 instantiations, gmpe_rollout_steps): K steps inside one launch, state carried in LDS / registers
    int K, S;                 // steps, action sets (step k reads action set k % S of act [S,N,A])
    int num_slots, first_slot;   // outputs of step k go to slot (first_slot + k) % num_slots
    long long st_obs, st_id, st_node, st_adj, st_rew, st_done, st_info, st_mask, st_tab;   // slot strides in elements
    int TW; uint32_t m_TW;    // doubles per env of the entity table output (gmpe_entity_table_width) and its division magic
    float* masks;             // optional [slots][N,A] GraphReplayBuffer masks / active_masks of the step (graph_buffer.py:223-251)
    float* active;
    // magic multipliers for exact unsigned division by run-time constants (q < 2^22): floor(q/d) = umulhi(q, m)
    uint32_t m_E, m_AE, m_EE, m_nq, m_2E, m_pe, m_AD, m_A, m_L, m_O, m_C, m_AC, m_AEE, m_W, m_Sx, m_FW;
    unsigned long long* stamps;   // diagnostic build only (-DGMPE_STAMPS): [grid][16] s_memtime per phase
};
__host__ __device__ inline uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(0x100000000ull / d) + 1u; }
__device__ __forceinline__ int fdiv(int q, int d, uint32_t m) {
    if (__builtin_constant_p(d)) return d <= 1 ? q : (int)((uint32_t)q / (uint32_t)d);   // exact-size instantiations: divisor known at compile time
    return d <= 1 ? q : (int)__umulhi((uint32_t)q, m);
}

// Timing-only ablations (wrong results by construction) exist in the diagnostic build alone: -DGMPE_DIAG (Makefile target `diag`).
#ifdef GMPE_DIAG
#define GMPE_ABL(p) ((p).ablate)
#else
#define GMPE_ABL(p) 0
#endif

// ---------------------------------------------------------------- LDS carve (dynamic, 16-B aligned)
// A workgroup owns G consecutive environments (G*A <= 64: every agent of every env is one lane of
// wave 0 for the sequential-semantics passes); all BLOCK threads share the O(E^2) distance pass and
// the streaming stores. Arrays below hold G envs back to back.
struct Lds {
    double *ex, *ey;                  // [G][E]  entity positions (agents: post-integration)
    double *s2, *s3;                  // [G][A]  theta/speed or vx/vy BEFORE this step's reward loop
    double *n2, *n3;                  // [G][A]  ... AFTER it (reset_velocity on goal reach)
    double *vox, *voy, *vnx, *vny;    // [G][A]  p_vel before / after
    double *serr;                     // [G][A]  spacing error of this step (…_july.py:1168-1180)
    double *cn, *sn;                  // [G][A]  cos / sin of the post-reward heading (rot_inv node features)
    double *rew;                      // [G][A]
    double *tube;                     // [G][12]
    double *Dm;                       // [G][A][E] fp64 agent->entity distances (rows of cached_dist_mag)
    double *fw;                       // [G][A][2*NW] wall contact forces (x, y per wall), walls variant only
    double *F2;                       // [2][G][A][A] pair forces (x block, y block) that outlive the fp32 matrix (NF > 0: fused rollouts, see distance_force_pass)
    double *cntd;                     // [G][A][2] goal_min_time, prev_proj + [G] delta_spacing: staged by the loader lanes; + [G] int64 RNG counter after a reset (rollouts)
    int *s_old, *newf, *gt;           // [G][A]  status before, newly-reached flag, goal_tracker (final)
    int *dtg_o, *dtg_n, *trq_o, *trq_n, *sv_o, *sv_n;   // [G][A] info counters old/new
    int *flags;                       // [G][4]  0: reset this env, 1: heading draws this step, 2: env has masked nodes, 3: env active
    int *moff;                        // [G][E]  adjacency mask per node (done agent / reached landmark)
    int *cnt;                         // [G][A][9] info counters staged by the loader lanes (read by wave 0 in section 3)
    int *ptab;                        // [A(A-1)/2] agent pairs (a<<8 | k), a < k, shared by the G envs of the tile
    float *obs;                       // [G][A*D] staging
    float *M;                         // [G][E*E] masked distance matrix, fp32
};
__host__ __device__ inline size_t lds_bytes(int G, int A, int E, int D, int NW, int NF = 0) {
    const size_t EE4 = ((size_t)E * E + 3) / 4 * 4, AD4 = ((size_t)A * D + 3) / 4 * 4;
    size_t d = (size_t)G * (2 * E + 14 * A + 14 + (size_t)A * E + (size_t)A * 2 * NW + (size_t)NF);   // doubles
    size_t f = (size_t)G * (EE4 + AD4);                             // floats
    size_t i = (size_t)G * (18 * A + 4 + E) + (size_t)A * (A - 1) / 2 + 1;   // ints
    return d * 8 + 16 + f * 4 + ((i * 4 + 15) / 16) * 16 + 32;
}
__device__ inline Lds carve(char* base, int G, int A, int E, int D, int NW, int NF = 0) {
    Lds l;
    double* d = reinterpret_cast<double*>(base);
    l.ex = d; d += G * E; l.ey = d; d += G * E;
    l.s2 = d; d += G * A; l.s3 = d; d += G * A; l.n2 = d; d += G * A; l.n3 = d; d += G * A;
    l.vox = d; d += G * A; l.voy = d; d += G * A; l.vnx = d; d += G * A; l.vny = d; d += G * A;
    l.serr = d; d += G * A; l.cn = d; d += G * A; l.sn = d; d += G * A; l.rew = d; d += G * A; l.tube = d; d += G * 12;
    l.Dm = d; d += (size_t)G * A * E;
    l.fw = d; d += (size_t)G * A * 2 * NW;
    l.F2 = d; d += (size_t)G * NF;
    l.cntd = d; d += (size_t)G * (2 * A + 2);
    if ((uintptr_t)d & 15) d += 1;
    float* f = reinterpret_cast<float*>(d);
    const size_t EE4 = ((size_t)E * E + 3) / 4 * 4, AD4 = ((size_t)A * D + 3) / 4 * 4;
    l.M = f; f += (size_t)G * EE4;
    l.obs = f; f += (size_t)G * AD4;
    int* i = reinterpret_cast<int*>(f);
    l.s_old = i; i += G * A; l.newf = i; i += G * A; l.gt = i; i += G * A;
    l.dtg_o = i; i += G * A; l.dtg_n = i; i += G * A; l.trq_o = i; i += G * A; l.trq_n = i; i += G * A;
    l.sv_o = i; i += G * A; l.sv_n = i; i += G * A; l.flags = i; i += G * 4; l.moff = i; i += G * E; l.cnt = i; i += G * A * 9; l.ptab = i;
    return l;
}
// view of env g inside the workgroup tile
__device__ inline Lds env_view(const Lds& l, int g, int A, int E, int D) {
    Lds v;
    const size_t EE4 = ((size_t)E * E + 3) / 4 * 4, AD4 = ((size_t)A * D + 3) / 4 * 4;
    v.ex = l.ex + g * E; v.ey = l.ey + g * E;
    v.s2 = l.s2 + g * A; v.s3 = l.s3 + g * A; v.n2 = l.n2 + g * A; v.n3 = l.n3 + g * A;
    v.vox = l.vox + g * A; v.voy = l.voy + g * A; v.vnx = l.vnx + g * A; v.vny = l.vny + g * A;
    v.serr = l.serr + g * A; v.cn = l.cn + g * A; v.sn = l.sn + g * A; v.rew = l.rew + g * A; v.tube = l.tube + g * 12;
    v.Dm = l.Dm + (size_t)g * A * E; v.fw = l.fw; v.F2 = l.F2; v.cntd = l.cntd; v.cnt = l.cnt;      // fw / cnt / cntd are indexed with the tile-level agent slot
    v.s_old = l.s_old + g * A; v.newf = l.newf + g * A; v.gt = l.gt + g * A;
    v.dtg_o = l.dtg_o + g * A; v.dtg_n = l.dtg_n + g * A; v.trq_o = l.trq_o + g * A; v.trq_n = l.trq_n + g * A;
    v.sv_o = l.sv_o + g * A; v.sv_n = l.sv_n + g * A; v.flags = l.flags + g * 4; v.moff = l.moff + g * E; v.ptab = l.ptab;
    v.obs = l.obs + g * AD4; v.M = l.M + g * EE4;
    return v;
}

// Streaming 16-byte store with the nontemporal hint (the observations are written once and read by another kernel).
// Written as inline asm: with the builtin inside `if (p.nt) ... else plain store` the optimiser merges the two stores and
// drops the hint. The asm statement carries its own wait state (cdna_hip_programming.md §5.7 item 2); a store has no
// result to wait for, and the kernel issues no load after these stores.
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4* dst, const float4& v) {
    v4f_t x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dst), "v"(x) : "memory");
}

template <int SC>
__device__ __forceinline__ void vel_of(double a2, double a3, double& vx, double& vy) {
    if (sc_kinematic(SC)) { double sn, cs; sincos(a2, &sn, &cs); vx = a3 * cs; vy = a3 * sn; }      // core.py:281-286
    else { vx = a2; vy = a3; }                                       // core.py:191-193
}

__device__ __forceinline__ bool wall_band_hit(const KParams& p, double px, double py, double size) {
    for (int w = 0; w < p.c.num_walls; ++w) {
        const gmpe_wall& wl = p.c.walls[w];
        const double band = 1.5 * size;
        const double perp = wl.orient == 0 ? py : px, prll = wl.orient == 0 ? px : py;
        if (wl.axis_pos - band <= perp && perp <= wl.axis_pos + band && wl.end0 - band <= prll && prll <= wl.end1 + band)
            return true;
    }
    return false;
}
// Scenario.is_obstacle_collision (…_july.py:864-890) for agent i at its current position, distances taken from the shared fp64 rows
__device__ __forceinline__ bool obstacle_collision_ego(const KParams& p, const Lds& l, int i) {
    const double* row = l.Dm + (size_t)i * p.E + p.A + p.L;
    for (int o = 0; o < p.O; ++o)
        if (row[o] < 2.0 * (p.c.entity_size + p.c.entity_size)) return true;
    return wall_band_hit(p, l.ex[i], l.ey[i], p.c.entity_size);
}

// _set_action (environment.py:336-475)
template <int SC>
__device__ __forceinline__ void decode_action(const gmpe_config& c, int idx, double& u0, double& u1) {
    if (!sc_kinematic(SC)) {
        if (c.n_actions == 5) {
            u0 = (idx == 1 ? 1.0 : 0.0) - (idx == 2 ? 1.0 : 0.0);
            u1 = (idx == 3 ? 1.0 : 0.0) - (idx == 4 ? 1.0 : 0.0);
        } else {                                                     // action_map 382-392
            const double d = 0.71;
            u0 = (idx == 1 ? -1.0 : idx == 5 ? 1.0 : (idx == 2 || idx == 8) ? -d : (idx == 4 || idx == 6) ? d : 0.0);
            u1 = (idx == 3 ? -1.0 : idx == 7 ? 1.0 : (idx == 2 || idx == 4) ? -d : (idx == 6 || idx == 8) ? d : 0.0);
        }
    } else {
        const int wi = idx / 5, ai = idx - wi * 5;
        u0 = wi == 0 ? c.ang_rate_opt[0] : wi == 1 ? c.ang_rate_opt[1] : wi == 2 ? c.ang_rate_opt[2] : wi == 3 ? c.ang_rate_opt[3] : c.ang_rate_opt[4];
        u1 = ai == 0 ? c.accel_opt[0] : ai == 1 ? c.accel_opt[1] : ai == 2 ? c.accel_opt[2] : ai == 3 ? c.accel_opt[3] : c.accel_opt[4];
    }
    u0 *= c.sensitivity; u1 *= c.sensitivity;
}

// Scenario.observation (…_july.py:1337-1463) for ego i into the LDS staging row; `phase` = value of
// the first get_agent_phase call of the step. Uses the PRE-reward own velocity (vox/voy).
template <int AP, int SC>
__device__ __forceinline__ void write_obs(const KParams& p, const Lds& l, int i, double vx, double vy, int phase) {
    float* o = l.obs + (size_t)i * ((AP > 0 && (SC == SC_NAV || SC == SC_JULY)) ? (SC == SC_JULY ? 19 : 13) : p.D);
    const double px = l.ex[i], py = l.ey[i];
    const double gx = l.ex[(AP ? AP : p.A) + i] - px, gy = l.ey[(AP ? AP : p.A) + i] - py;
    o[0] = (float)px; o[1] = (float)py; o[2] = (float)vx; o[3] = (float)vy;
    o[4] = (float)gx; o[5] = (float)gy; o[6] = 0.0f; o[7] = (float)gx; o[8] = (float)gy;
    // stable two-smallest of the other agents (1398-1417): strict '<' keeps the first of equal distances
    const double INF = __builtin_huge_val();
    int b1 = -1, b2 = -1; double d1 = INF, d2 = INF;
    const double* row = l.Dm + (size_t)i * ((AP > 0 && (SC == SC_NAV || SC == SC_JULY)) ? 2 * AP : p.E);
    if (AP) {
        double rv[AP ? AP : 1];
#pragma unroll
        for (int k = 0; k < AP; ++k) rv[k] = row[k < (AP ? AP : p.A) ? k : 0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < AP; ++k) {
            const double d = (k < (AP ? AP : p.A) && k != i) ? rv[k] : INF;
            const bool lt1 = d < d1, lt2 = d < d2;
            d2 = lt1 ? d1 : (lt2 ? d : d2); b2 = lt1 ? b1 : (lt2 ? k : b2);
            d1 = lt1 ? d : d1; b1 = lt1 ? k : b1;
        }
    } else {
        for (int k = 0; k < (AP ? AP : p.A); ++k) {
            const double d = k != i ? row[k] : INF;
            const bool lt1 = d < d1, lt2 = d < d2;
            d2 = lt1 ? d1 : (lt2 ? d : d2); b2 = lt1 ? b1 : (lt2 ? k : b2);
            d1 = lt1 ? d : d1; b1 = lt1 ? k : b1;
        }
    }
    o[9] = b1 >= 0 ? (float)(l.ex[b1] - px) : 0.f; o[10] = b1 >= 0 ? (float)(l.ey[b1] - py) : 0.f;
    o[11] = b2 >= 0 ? (float)(l.ex[b2] - px) : 0.f; o[12] = b2 >= 0 ? (float)(l.ey[b2] - py) : 0.f;
    if (SC == SC_JULY) {
        o[13] = (float)(l.tube[T_ENTX] - px); o[14] = (float)(l.tube[T_ENTY] - py);
        o[15] = (float)(l.tube[T_EXX] - px); o[16] = (float)(l.tube[T_EXY] - py);
        o[17] = (float)l.tube[T_WIDTH]; o[18] = (float)phase;
    }
}

// Scenario.observation of the rot_inv family (…rot_inv.py:1453-1548): float32 = [cos th, sin th, speed, goal (rotated), two
// nearest neighbours (rel. vector cast to float32, then rotated in float64), s/L, y/half_w (clipped), exit-gate distance / L,
// phase] = 13; two_phase_graph.py:1152-1236 / three_phase_graph.py put [cos, sin](heading error to the corridor axis) before the
// phase (D = 15), skip finished agents among the neighbours, and two_phase replaces the goal vector by float32(exit) - pos.
// Uses the PRE-reward heading (the reward of this agent runs after its observation); agent k counts as finished for ego i
// iff s_old[k] or (new[k] and k < i) — the ordered-visibility rule.
template <int AP, int SC>
__device__ __forceinline__ void write_obs_rot(const KParams& p, const Lds& l, int i, int phase) {
    float* o = l.obs + (size_t)i * p.D;
    const double px = l.ex[i], py = l.ey[i], th = l.s2[i];
    double sn, cs; sincos(th, &sn, &cs);
    const double INF = __builtin_huge_val();
    int b1 = -1, b2 = -1; double d1 = INF, d2 = INF;
    const double* row = l.Dm + (size_t)i * p.E;
    for (int k = 0; k < p.A; ++k) {
        bool skip = k == i;
        if (sc_phasefam(SC)) skip = skip || l.s_old[k] || (l.newf[k] && k < i);
        const double d = !skip ? row[k] : INF;
        const bool lt1 = d < d1, lt2 = d < d2;
        d2 = lt1 ? d1 : (lt2 ? d : d2); b2 = lt1 ? b1 : (lt2 ? k : b2);
        d1 = lt1 ? d : d1; b1 = lt1 ? k : b1;
    }
    double gx, gy;
    if (SC == SC_TWO) rot2(cs, sn, (double)(float)l.tube[T_EXX] - px, (double)(float)l.tube[T_EXY] - py, gx, gy);
    else rot2(cs, sn, l.ex[p.A + i] - px, l.ey[p.A + i] - py, gx, gy);
    double n1x = 0, n1y = 0, n2x = 0, n2y = 0;
    if (b1 >= 0) rot2(cs, sn, (double)(float)(l.ex[b1] - px), (double)(float)(l.ey[b1] - py), n1x, n1y);
    if (b2 >= 0) rot2(cs, sn, (double)(float)(l.ex[b2] - px), (double)(float)(l.ey[b2] - py), n2x, n2y);
    const double L = l.tube[T_L], hw = l.tube[T_HALFW];
    double s, yy; tube_sy(l.tube, px, py, s, yy);
    o[0] = (float)cs; o[1] = (float)sn; o[2] = (float)l.s3[i];
    o[3] = (float)gx; o[4] = (float)gy; o[5] = (float)n1x; o[6] = (float)n1y; o[7] = (float)n2x; o[8] = (float)n2y;
    o[9] = (float)clipd(s / L, -2.0, 2.0); o[10] = (float)clipd(yy / (hw + 1e-9), -2.0, 2.0);
    o[11] = (float)(exit_gate_distance(s, yy, L, hw) / (L + 1e-9));
    if (sc_phasefam(SC)) {
        double hs, hc; sincos(heading_error_signed(l.tube, th), &hs, &hc);
        o[12] = (float)hc; o[13] = (float)hs; o[14] = (float)phase;
    } else o[12] = (float)phase;
}

// Draw window of a resetting env: its A agent lanes evaluate A consecutive draws of the env's stream at once — one Philox evaluation (or one tape load) per
// lane — into the env's LDS slots, instead of every lane evaluating every draw of the sequential placement loop (the lanes of an env run that loop in lock
// step, so they all need the same draw at the same time). win_draw(k) serves draw k and refills the window when k runs past it; values, order and the sticky
// tape-exhausted flag are exactly draw_at's (a draw past the tape only counts when it is consumed).
__device__ __forceinline__ double win_draw(const KParams& p, double* buf, int64_t& base, int W, int n, int i, int64_t k, int& err) {
    if (k >= base + W) {                                                     // env-uniform (k and base are)
        base = k;
        const int64_t kk = k + i;
        double val;
        if (p.s.tape) val = kk < p.s.tape_len ? p.s.tape[(size_t)n * p.s.tape_len + kk] : __builtin_nan("");
        else val = philox_uniform(p.c.seed, (uint32_t)(p.c.env_id_base + n), (uint64_t)kk);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");               // earlier reads of the old window are done
        buf[i] = val;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // LDS operations of one wave execute in order
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    double v = buf[(int)(k - base)];
    if (v != v) { err |= 1; v = 0.5; }                                       // tape exhausted (draw_at's rule)
    return v;
}

// ---- batched placement (exact-size rollout kernels; reset_world_coop<SC, true>) ----
// sqrt(dx^2 + dy^2) < thr — the rejection test of the placement loops (…_july.py:895-904) — decided on the squared distance unless it lies within a relative
// 1e-12 of thr^2, where the exact comparison runs: with a correctly rounded sqrt (relative error 1.1e-16) the two outer cases cannot disagree with it.
__device__ __forceinline__ bool closer_than(double dx, double dy, double thr) {
    const double d2 = dx * dx + dy * dy, t2 = thr * thr;
    if (d2 < t2 * (1.0 - 1e-12)) return true;
    if (d2 > t2 * (1.0 + 1e-12)) return false;
    return sqrt(d2) < thr;
}
// One draw of env n's stream: the tape value (0.5 past its end: draw_at's rule) or Philox. The sticky tape-exhausted bit is set by the caller from the final
// counter — exactly the draws [ctr_in, ctr_out) of a reset are consumed, whatever was evaluated speculatively.
__device__ __forceinline__ double draw_raw(const KParams& p, int n, int64_t k) {
    if (p.s.tape) return k < p.s.tape_len ? p.s.tape[(size_t)n * p.s.tape_len + k] : 0.5;
    return philox_uniform(p.c.seed, (uint32_t)(p.c.env_id_base + n), (uint64_t)k);
}
__device__ __forceinline__ void wave_lds_sync() {                           // LDS operations of one wave execute in order: the fences pin the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Draws of a resetting env for the batched placement: the tile's streaming waves evaluate the first `nd` draws of the env's stream from its reset counter on
// (k_env: while wave 0 computes the terminal step's rewards / info) into the env's fp32-matrix region, which is dead between the step's distance pass and the reset's;
// a placement round then READS its draws instead of evaluating Philox inside its chain. Draws past the buffer (a reset that rejects unusually often) are evaluated in place.
struct DrawBuf { const double* d; int64_t base; int nd; };
__device__ __forceinline__ double buf_get(const KParams& p, const DrawBuf& w, int n, int64_t k) {
    const int64_t o = k - w.base;
    return o < w.nd ? w.d[o] : draw_raw(p, n, k);
}
// navigation_graph placement of one entity class (M entities at ex/ey[base ...], pairwise at least `thr` apart, optionally clear of obstacles / wall bands) by
// greedy rejection in stream order — attempt t uses draws (ctr + 2t, ctr + 2t + 1) and is accepted iff it is clear of everything accepted before it, or it is the
// GMPE_MAX_TRIES-th consecutive failure for its entity (oracle/gmpe_oracle.c restates the loop literally). The candidates do not depend on what was accepted, so
// the A lanes of the env evaluate A consecutive attempts at once: lane i tests attempt i against the entities placed by earlier batches and against the earlier
// attempts of its batch (a bit mask), and every lane replays the greedy decisions of the batch on those masks — a handful of bit operations per attempt instead
// of a dependent LDS / fp64 chain per attempt.
__device__ __forceinline__ void place_uniform(const KParams& p, const Lds& l, const DrawBuf& win, int n, int i, bool mine, unsigned long long emask, int M, int base, double thr,
                                              bool check_static, bool is_agent, int64_t& ctr, int& err) {
    const gmpe_config& c = p.c;
    const int A = p.A, O = p.O, o0 = p.A + p.L;
    const double ws = c.world_size, size = c.entity_size, lo = -ws / 2, hi = ws / 2;
    const int lane0 = emask ? __ffsll((long long)emask) - 1 : 0;
    double* cx = l.vox; double* cy = l.voy;                                 // batch scratch: rows of the env that the re-initialisation after the placement rewrites
    unsigned long long* cm = reinterpret_cast<unsigned long long*>(l.vnx);
    int k = 0, tries = 0;
    while (__ballot(mine && k < M)) {
        if (mine && k < M) {
            const double px = 0.8 * (lo + (hi - lo) * buf_get(p, win, n, ctr + 2 * i));
            const double py = 0.8 * (lo + (hi - lo) * buf_get(p, win, n, ctr + 2 * i + 1));
            bool bad0 = false;
            if (check_static) {
                bad0 = wall_band_hit(p, px, py, size);
                for (int o = 0; o < O; ++o) bad0 = bad0 || closer_than(l.ex[o0 + o] - px, l.ey[o0 + o] - py, 2.0 * (size + size));
            }
            for (int q = 0; q < k; ++q) bad0 = bad0 || closer_than(l.ex[base + q] - px, l.ey[base + q] - py, thr);
            cx[i] = px; cy[i] = py;
            wave_lds_sync();
            unsigned long long cl = 0ull;
            for (int j = 0; j < i; ++j) if (closer_than(cx[j] - px, cy[j] - py, thr)) cl |= 1ull << j;
            cm[i] = cl;
            wave_lds_sync();
            const unsigned long long bad = (__ballot(bad0) & emask) >> lane0;   // bit j: attempt j fails against the entities of earlier batches / statics
            unsigned long long acc = 0ull;
            int kk = k, t = tries, used = 0, slot = -1;
            for (int j = 0; j < A && kk < M; ++j) {                          // env-uniform: every lane of the env replays the same decisions
                const bool b = ((bad >> j) & 1ull) || (cm[j] & acc) != 0ull;
                used = j + 1;
                if (b && ++t < GMPE_MAX_TRIES) continue;
                if (b) err |= 2;
                if (j == i) slot = kk;
                acc |= 1ull << j; ++kk; t = 0;
            }
            wave_lds_sync();                                                 // everybody has read the batch before the next one overwrites it
            if (slot >= 0) {
                l.ex[base + slot] = px; l.ey[base + slot] = py;
                if (is_agent) { l.n2[slot] = 0.0; l.n3[slot] = 0.0; }
            }
            wave_lds_sync();
            ctr += 2 * used; k = kk; tries = t;
        }
    }
}

// Reset of the envs of a tile by wave 0 (reset_world: …_july.py:339-420, 440-515, 518-613, custom_scenarios/utils.py:165-193;
// navigation_graph: DESIGN.md). The reference places entities one after another with rejection sampling — inherently sequential
// in the entity index, but each attempt's collision test against the already placed entities is not: every agent lane of the env
// computes the SAME candidate (same draws of the env's stream, same arithmetic: no shuffles) and tests it against the entities it
// OWNS (agent i; obstacles / landmarks i, i+A, ...: a lane only ever re-reads its own LDS writes), and one ballot over the env's
// lanes decides. Draw order and count are exactly the sequential ones (oracle/gmpe_oracle.c restates the loop literally), the
// rejection loop is bounded. One lane per env doing all of it serially cost 70 us per reset step at c2 and 4-5 ms at c4
// (profiles/README.md): a chain of ~300-cycle fp64 sqrt per placed entity and attempt on a single lane.
// All lanes of wave 0 call this (wave-uniform); `mine` = lane belongs to an env that resets. Writes positions / headings to LDS
// (ex, ey, n2, n3, tube) and landmark / obstacle / tube records to HBM.
// (Round 4 tried computing the NEXT episode's placement ahead of the reset — exact, since a reset is a pure function of the env's draw counter — on wave 3 of a
// steady rollout step: the second copy of this loop in the step loop cost more registers than the shorter reset step gave back; profiles/r04_placement_ahead_experiment.patch.)
// BATCH (exact-size rollout kernels): the attempts themselves run in parallel. Kinematic scenarios — the candidate of agent k depends on k (its slot on the line) and
// an accepted agent consumes one more draw (its heading), so the A lanes of the env evaluate A consecutive attempts of the CURRENT agent at once and the first clear
// one (or the GMPE_MAX_TRIES-th failure) wins; navigation_graph — candidates are independent of what was accepted, so a batch of A attempts is resolved greedily in
// stream order (place_uniform). Draws are read from the env's prefilled buffer (DrawBuf). Same placements, draw order and counters (the GPU tests compare placements and RNG counters with the
// oracle's literal loops). Used by the exact-size ROLLOUT kernels of navigation_graph and July (k_env below): all-env reset step 32-44 -> 26-31 us at c2, 48-59 -> 40-45 at c3,
// rollouts -1...-2 % (profiles/r04_ab_batched_placement_abk.log). Not by the step kernels (round 3: at their register cap they lost more than the reset gained) and not by the
// rot_inv family, whose wider jitter rejects little — its rounds stay ~A and each batched round costs more (reset step 30-34 -> 41-43 us).
template <int SC, bool BATCH = false>
__device__ __forceinline__ void reset_world_coop(const KParams& p, const Lds& l, int n, int i, bool mine, unsigned long long emask, int64_t& ctr, int& err) {
    const gmpe_config& c = p.c;
    const double ws = c.world_size, size = c.entity_size;
    const int A = p.A, L = p.L, O = p.O;
    const int o0 = A + L;
    int64_t wbase = -(int64_t)A - 1;                                        // empty window; slots = the env's spacing-error row (consumed before a reset)
    const int64_t ctr_in = ctr;
    const DrawBuf win = {reinterpret_cast<const double*>(l.M), ctr, BATCH ? (p.E * p.E + 3) / 4 * 2 : 0};   // BATCH: the env's prefilled draws (k_env)
    const int lane0 = emask ? __ffsll((long long)emask) - 1 : 0;
#define WDRAW(k) (BATCH ? buf_get(p, win, n, (k)) : win_draw(p, l.serr, wbase, A, n, i, (k), err))
    if (sc_kinematic(SC)) {
        double entx = 0, enty = 0, exx = 0, exy = 0, sa = 0, ca = 1;
        if (mine) {
            (void)WDRAW(ctr++);                              // wall_length draw, unused (:368)
            const double a = 3 * size * 2.5, b = ws * 0.15;
            const double width = a > b ? a : b;
            const double angle = -M_PI / 2 + (M_PI / 2 - (-M_PI / 2)) * WDRAW(ctr++);
            double tl = ws * 0.8;
            if (sc_phasefam(SC)) tl += -ws * 0.3 + (ws * 0.1 - (-ws * 0.3)) * WDRAW(ctr++);   // two_phase_graph.py:506
            ca = cos(angle); sa = sin(angle);
            const double be = tl / 4, bx = -tl / 4;
            entx = ca * 0 + sa * be; enty = -sa * 0 + ca * be;
            exx = ca * 0 + sa * bx; exy = -sa * 0 + ca * bx;
            const double dx = exx - entx, dy = exy - enty;
            const double Lt = sqrt(dx * dx + dy * dy) + 1e-9;
            const double ex = dx / Lt, ey = dy / Lt;
            if (i == 0) {
                double* t = l.tube;
                t[T_ANGLE] = angle; t[T_ENTX] = entx; t[T_ENTY] = enty; t[T_EXX] = exx; t[T_EXY] = exy;
                t[T_EX] = ex; t[T_EY] = ey; t[T_NX] = (double)(float)(-ey); t[T_NY] = (double)(float)ex;
                t[T_L] = Lt; t[T_HALFW] = width * 0.5; t[T_WIDTH] = width;
                for (int q = 0; q < GMPE_TUBE_STRIDE; ++q) p.s.tube[(size_t)n * GMPE_TUBE_STRIDE + q] = t[q];
            }
        }
        int k = 0, tries = 0;
        double mx = 0, my = 0;                                              // agent i's accepted position (this lane owns agent i)
        if constexpr (BATCH) while (__ballot(mine && k < A)) {
            if (mine && k < A) {
                // lane i evaluates attempt i of agent k: draws (ctr + 2i, ctr + 2i + 1); the accepted attempt's heading draw follows them
                const double u0 = buf_get(p, win, n, ctr + 2 * i), u1 = buf_get(p, win, n, ctr + 2 * i + 1);
                constexpr bool rot = sc_rotfam(SC);             // rot_inv.py:463, 469
                const double jf = rot ? 0.3 : 0.2;
                const double jx = jf * (-ws + (ws - (-ws)) * u0), jy = jf * (-ws + (ws - (-ws)) * u1);
                const double dfe = rot ? (ws + k) / 3 : (ws + k) / 5;
                const double px = entx + dfe * sa + jx, py = enty + dfe * ca + jy;
                bool bad_i = wall_band_hit(p, px, py, size);
                for (int o = 0; o < O; ++o) bad_i = bad_i || closer_than(l.ex[o0 + o] - px, l.ey[o0 + o] - py, 2.0 * (size + size));
                for (int q = 0; q < k; ++q) bad_i = bad_i || closer_than(l.ex[q] - px, l.ey[q] - py, c.sep_dist);
                const unsigned long long all = emask >> lane0;
                const unsigned long long good = ~((__ballot(bad_i) & emask) >> lane0) & all;
                const int jg = good ? __ffsll((long long)good) - 1 : A;      // first clear attempt of the batch
                const int jfo = GMPE_MAX_TRIES - 1 - tries;                  // the attempt that is accepted even if it fails (the reference would spin forever)
                int js;
                if (jfo < A && jfo < jg) { js = jfo; err |= 2; }
                else if (jg < A) js = jg;
                else { tries += A; ctr += 2 * A; continue; }
                ctr += 2 * (js + 1);
                const double th = 0.0 + (2 * M_PI - 0.0) * buf_get(p, win, n, ctr); ++ctr;
                if (i == js) { l.ex[k] = px; l.ey[k] = py; l.n2[k] = th; l.n3[k] = c.v_min; }
                wave_lds_sync();
                ++k; tries = 0;
            }
        }
        else while (__ballot(mine && k < A)) {
            if (mine && k < A) {
                const double u0 = WDRAW(ctr), u1 = WDRAW(ctr + 1);
                ctr += 2;
                constexpr bool rot = sc_rotfam(SC);             // rot_inv.py:463, 469
                const double jf = rot ? 0.3 : 0.2;
                const double jx = jf * (-ws + (ws - (-ws)) * u0), jy = jf * (-ws + (ws - (-ws)) * u1);
                const double dfe = rot ? (ws + k) / 3 : (ws + k) / 5;
                const double px = entx + dfe * sa + jx, py = enty + dfe * ca + jy;
                bool bad_i = wall_band_hit(p, px, py, size);
                for (int o = i; o < O; o += A) bad_i = bad_i || norm2(l.ex[o0 + o] - px, l.ey[o0 + o] - py) < 2.0 * (size + size);
                bad_i = bad_i || (i < k && norm2(mx - px, my - py) < c.sep_dist);
                const bool bad = (__ballot(bad_i) & emask) != 0ull;
                if (bad && ++tries < GMPE_MAX_TRIES) continue;
                if (bad) err |= 2;
                const double th = 0.0 + (2 * M_PI - 0.0) * WDRAW(ctr++);
                if (i == k) { mx = px; my = py; l.ex[k] = px; l.ey[k] = py; l.n2[k] = th; l.n3[k] = c.v_min; }
                ++k; tries = 0;
            }
        }
        if (mine) {
            // landmarks by args.formation_type (…_july.py:492-497; rot_inv.py:493-498, two_phase_graph.py:464-469, three_phase_graph.py:459-464);
            // lane i owns landmarks i, i+A, ... ; the landmarks' reset_velocity() draws nothing
            if (c.formation_type == GMPE_FORMATION_LINE) {
                // np.linspace((-ws/2, -ws/2), (ws/2, -ws/2), L) (utils.py:77-130): the zero y step sends BOTH coordinates through linspace's
                // (i / div) * delta + start branch; the last row is the stop value itself
                const int div = L - 1;
                const double s0 = -ws / 2, e0 = ws / 2, ddx = e0 - s0, ddy = s0 - s0;
                for (int q = i; q < L; q += A) {
                    const double f = div > 0 ? (double)q / (double)div : (double)q;
                    double lx = f * ddx + s0, ly = f * ddy + s0;
                    if (L > 1 && q == L - 1) { lx = e0; ly = s0; }
                    bool hit = wall_band_hit(p, lx, ly, size);                  // the reference raises ValueError here (utils.py:113-114)
                    for (int o = 0; o < O; ++o) hit = hit || norm2(l.ex[o0 + o] - lx, l.ey[o0 + o] - ly) < 2.0 * (size + size);
                    if (hit) err |= 2;
                    l.ex[A + q] = lx; l.ey[A + q] = ly;
                }
            } else if (c.formation_type == GMPE_FORMATION_CIRCLE) {
                // set_landmarks_in_circle (utils.py:231-267): centre (0, exit_y + ws/5), radius ws/3
                const double cy = exy + ws / 5, radius = ws / 3, angle_step = 2 * M_PI / L;
                for (int q = i; q < L; q += A) {
                    const double ang = q * angle_step;
                    l.ex[A + q] = 0.0 + radius * cos(ang); l.ey[A + q] = cy + radius * sin(ang);
                }
            } else {
                const double rel = -ws / 3;
                const double rx = ca * 0.0 + sa * rel, ry = -sa * 0.0 + ca * rel;
                for (int q = i; q < L; q += A) { l.ex[A + q] = exx + rx; l.ey[A + q] = exy + ry; }
            }
        }
    } else if constexpr (BATCH) {
        place_uniform(p, l, win, n, i, mine, emask, O, o0, 2.0 * (size + size), false, false, ctr, err);   // obstacles: >= 2*(size+size) apart
        place_uniform(p, l, win, n, i, mine, emask, A, 0, c.sep_dist, true, true, ctr, err);               // agents: clear of obstacles / wall bands, >= sep_dist apart
        place_uniform(p, l, win, n, i, mine, emask, L, A, c.sep_dist, true, false, ctr, err);              // goals (landmarks): likewise
        if (mine) for (int o = i; o < O; o += A) {
            p.s.obstacles[((size_t)n * O + o) * 2] = l.ex[o0 + o];
            p.s.obstacles[((size_t)n * O + o) * 2 + 1] = l.ey[o0 + o];
        }
    } else {
        const double lo = -ws / 2, hi = ws / 2;
        int k = 0, tries = 0;
        // obstacles: >= 2*(size+size) apart; lane i owns obstacles i, i+A, ...
        while (__ballot(mine && k < O)) {
            if (mine && k < O) {
                const double px = 0.8 * (lo + (hi - lo) * WDRAW(ctr));
                const double py = 0.8 * (lo + (hi - lo) * WDRAW(ctr + 1));
                ctr += 2;
                bool bad_i = false;
                for (int q = i; q < k; q += A) bad_i = bad_i || norm2(l.ex[o0 + q] - px, l.ey[o0 + q] - py) < 2.0 * (size + size);
                const bool bad = (__ballot(bad_i) & emask) != 0ull;
                if (bad && ++tries < GMPE_MAX_TRIES) continue;
                if (bad) err |= 2;
                if (k % A == i) { l.ex[o0 + k] = px; l.ey[o0 + k] = py; }
                ++k; tries = 0;
            }
        }
        // agents: clear of obstacles / wall bands, >= sep_dist apart
        k = 0; tries = 0;
        double mx = 0, my = 0;
        while (__ballot(mine && k < A)) {
            if (mine && k < A) {
                const double px = 0.8 * (lo + (hi - lo) * WDRAW(ctr));
                const double py = 0.8 * (lo + (hi - lo) * WDRAW(ctr + 1));
                ctr += 2;
                bool bad_i = wall_band_hit(p, px, py, size);
                for (int o = i; o < O; o += A) bad_i = bad_i || norm2(l.ex[o0 + o] - px, l.ey[o0 + o] - py) < 2.0 * (size + size);
                bad_i = bad_i || (i < k && norm2(mx - px, my - py) < c.sep_dist);
                const bool bad = (__ballot(bad_i) & emask) != 0ull;
                if (bad && ++tries < GMPE_MAX_TRIES) continue;
                if (bad) err |= 2;
                if (i == k) { mx = px; my = py; l.ex[k] = px; l.ey[k] = py; l.n2[k] = 0.0; l.n3[k] = 0.0; }
                ++k; tries = 0;
            }
        }
        // goals (landmarks): clear of obstacles / wall bands, >= sep_dist apart; lane i owns landmarks i, i+A, ...
        k = 0; tries = 0;
        while (__ballot(mine && k < L)) {
            if (mine && k < L) {
                const double px = 0.8 * (lo + (hi - lo) * WDRAW(ctr));
                const double py = 0.8 * (lo + (hi - lo) * WDRAW(ctr + 1));
                ctr += 2;
                bool bad_i = wall_band_hit(p, px, py, size);
                for (int o = i; o < O; o += A) bad_i = bad_i || norm2(l.ex[o0 + o] - px, l.ey[o0 + o] - py) < 2.0 * (size + size);
                for (int r = i; r < k; r += A) bad_i = bad_i || norm2(l.ex[A + r] - px, l.ey[A + r] - py) < c.sep_dist;
                const bool bad = (__ballot(bad_i) & emask) != 0ull;
                if (bad && ++tries < GMPE_MAX_TRIES) continue;
                if (bad) err |= 2;
                if (k % A == i) { l.ex[A + k] = px; l.ey[A + k] = py; }
                ++k; tries = 0;
            }
        }
        if (mine) for (int o = i; o < O; o += A) {
            p.s.obstacles[((size_t)n * O + o) * 2] = l.ex[o0 + o];
            p.s.obstacles[((size_t)n * O + o) * 2 + 1] = l.ey[o0 + o];
        }
    }
    if (mine) for (int q = i; q < L; q += A) {
        p.s.landmarks[((size_t)n * L + q) * 2] = l.ex[A + q];
        p.s.landmarks[((size_t)n * L + q) * 2 + 1] = l.ey[A + q];
    }
    if (BATCH && mine && p.s.tape && ctr > ctr_in && ctr > p.s.tape_len) err |= 1;    // a consumed draw lay past the tape (draw_at's sticky bit)
}
#undef WDRAW


// Post-move distance pass for every env of the tile (World.calculate_distances, core.py:600-624:
// delta taken as pos[min]-pos[max], so the matrix is exactly symmetric). Writes the fp64 agent rows
// Dm[g][r][c] (r < A) AND the whole unmasked fp32 matrix M, static (landmark / obstacle) block included.
// U independent pairs of one lane (q0, q0+BLOCK, ...), interleaved: the pass is a chain of dependent LDS reads + an fp64 sqrt
template <int BLOCK, int U, int CA>
__device__ __forceinline__ void distance_trip(const KParams& p, const Lds& l, int q0, int total, int W, int dv, bool even, bool only_reset) {
    const int A = CA ? CA : p.A, E = CA ? 2 * CA : p.E, AE = A * E, EE4 = (E * E + 3) / 4 * 4;
    double ds[U]; int gs[U], rs[U], cs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = q0 + u * BLOCK;
        const bool live = q < total;
        const int qq = live ? q : 0;
        const int g = fdiv(qq, W, p.m_W), w = qq - g * W;
        const int a = fdiv(w, dv, p.m_Sx), b = w - a * dv;
        int r, cc;
        if (even) { const bool up = b >= a; r = up ? a : E - 1 - a; cc = up ? b + 1 : E - a + b; }
        else { int c0 = a + 1 + b; c0 = c0 >= E ? c0 - E : c0; r = a < c0 ? a : c0; cc = a < c0 ? c0 : a; }
        const double dx = l.ex[g * E + r] - l.ex[g * E + cc], dy = l.ey[g * E + r] - l.ey[g * E + cc];   // pos[min] - pos[max] (core.py:600-624)
        ds[u] = (GMPE_ABL(p) & 16) ? dx * dx + dy * dy : sqrt(dx * dx + dy * dy);   // 16: diagnostic build only
        gs[u] = (live && !(only_reset && !l.flags[g * 4 + 0])) ? g : -1; rs[u] = r; cs[u] = cc;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (gs[u] < 0) continue;
        const int r = rs[u], cc = cs[u];
        float* Mg = l.M + (size_t)gs[u] * EE4;
        const float df = (float)ds[u];
        Mg[r * E + cc] = df; Mg[cc * E + r] = df;
        if (r < A) {
            double* Dg = l.Dm + (size_t)gs[u] * AE;
            Dg[r * E + cc] = ds[u];
            if (cc < A) Dg[cc * E + r] = ds[u];
        }
    }
}
template <int BLOCK, int CA>
__device__ __forceinline__ void distance_pass(const KParams& p, const Lds& l, int G, int tid, bool only_reset) {
    // One lane per UNORDERED entity pair (r < c) — agent-agent, agent-static and static-static alike.
    // Pair index without a table: E even: w = a*(E-1)+b, a < E/2: b >= a -> (a, b+1), else the folded row (E-1-a, E-a+b);
    // E odd: w = a*((E-1)/2)+b -> (a, a+1+b mod E), the circulant enumeration. Each pair appears exactly once.
    // A wave takes its remaining pairs (up to 5 per lane) in ONE interleaved trip, sized wave-uniformly so that no dead
    // entry is computed (C2/C3 tile of 4 envs: 760 pairs on 256 lanes = 3 per lane).
    const int A = CA ? CA : p.A, E = CA ? 2 * CA : p.E, AE = A * E, EE4 = (E * E + 3) / 4 * 4;
    const int W = E * (E - 1) / 2;
    const bool even = (E & 1) == 0;
    const int dv = even ? E - 1 : (E - 1) / 2;
    const int total = G * W;
    const int wave_base = tid & ~63;
    for (int base = 0; base + wave_base < total; base += 5 * BLOCK) {
        const int left = total - (base + wave_base);                    // wave-uniform
        const int nu = (left + BLOCK - 1) / BLOCK;
        const int q0 = base + tid;
        if (nu >= 5) distance_trip<BLOCK, 5, CA>(p, l, q0, total, W, dv, even, only_reset);
        else if (nu == 4) distance_trip<BLOCK, 4, CA>(p, l, q0, total, W, dv, even, only_reset);
        else if (nu == 3) distance_trip<BLOCK, 3, CA>(p, l, q0, total, W, dv, even, only_reset);
        else if (nu == 2) distance_trip<BLOCK, 2, CA>(p, l, q0, total, W, dv, even, only_reset);
        else distance_trip<BLOCK, 1, CA>(p, l, q0, total, W, dv, even, only_reset);
    }
    for (int q = tid; q < G * E; q += BLOCK) {                           // diagonal
        const int g = fdiv(q, E, p.m_E), r = q - g * E;
        if (only_reset && !l.flags[g * 4 + 0]) continue;
        l.M[(size_t)g * EE4 + r * E + r] = 0.0f;
        if (r < A) l.Dm[(size_t)g * AE + r * E + r] = 0.0;
    }
}

#define GMPE_NSTAMPS 32
#ifdef GMPE_STAMPS
#define STAMP(k) do { if (tid == 0 && p.stamps) p.stamps[(size_t)blockIdx.x * GMPE_NSTAMPS + (k)] = __builtin_readcyclecounter(); } while (0)
#define STAMP_T(k, t) do { if (tid == (t) && p.stamps) p.stamps[(size_t)blockIdx.x * GMPE_NSTAMPS + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP_T(k, t) do { } while (0)
#endif
// Rollouts of exact-size navigation_graph tiles (A = L = CA, no obstacles, no walls): the distance pass of step k ALSO computes the contact forces of step
// k + 1 (get_entity_collision_force, core.py:872-906) — the positions after step k's move are the positions step k + 1's force pass would read, the pair
// (a, k > a) and its delta pos[a] - pos[k] are the same, so the values are bit-identical — and it visits only the pairs that change: agent x agent and agent x
// landmark (145 of an env's 190); the landmark x landmark block of the fp32 matrix is built once per launch / reset (`statics`) and stays (its mask only grows
// between resets). One fp64 sqrt per agent pair instead of two, 24 % fewer pairs per step, and the separate force pass with its barrier disappears from the
// step. Pair order across the tile: all envs' agent pairs first (their softplus code then runs in the first trip of the first waves only), then the
// agent x landmark pairs, then the statics. The forces go to a buffer of their own (F2): they outlive the fp32 matrix the classic force pass aliases.
template <int BLOCK, int CA, bool FORCES>
__device__ __forceinline__ void distance_force_pass(const KParams& p, const Lds& l, int G, int Gv, int tid, bool only_reset, bool statics) {
    constexpr int A = CA, E = 2 * CA, AE = A * E, EE4 = (E * E + 3) / 4 * 4, NP = A * (A - 1) / 2, NL = A * A;
    const gmpe_config& c = p.c;
    double* Fx = l.F2; double* Fy = Fx + (size_t)G * A * A;
    const int n1 = Gv * NP, n2 = n1 + Gv * NL, total = n2 + (statics ? Gv * NP : 0);
    const int wave_base = tid & ~63;
    for (int base = 0; base + wave_base < total; base += 3 * BLOCK) {
        double ds[3], dxs[3], dys[3]; int gs[3], rs[3], cs[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = base + u * BLOCK + tid;
            const bool live = q < total;
            const int qq = live ? q : 0;
            int g, r, cc;
            if (qq < n1) { g = qq / NP; const int pk = l.ptab[qq - g * NP]; r = pk >> 8; cc = pk & 255; }                 // agent pair (a < k)
            else if (qq < n2) { const int t = qq - n1; g = t / NL; const int w = t - g * NL; r = w / A; cc = A + (w - r * A); }   // agent x landmark
            else { const int t = qq - n2; g = t / NP; const int pk = l.ptab[t - g * NP]; r = A + (pk >> 8); cc = A + (pk & 255); }   // landmark pair (L = A)
            const double dx = l.ex[g * E + r] - l.ex[g * E + cc], dy = l.ey[g * E + r] - l.ey[g * E + cc];   // pos[min] - pos[max] (core.py:600-624)
            dxs[u] = dx; dys[u] = dy; ds[u] = sqrt(dx * dx + dy * dy);
            gs[u] = (live && !(only_reset && !l.flags[g * 4 + 0])) ? g : -1; rs[u] = r; cs[u] = cc;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (gs[u] < 0) continue;
            const int r = rs[u], cc = cs[u];
            float* Mg = l.M + (size_t)gs[u] * EE4;
            const float df = (float)ds[u];
            Mg[r * E + cc] = df; Mg[cc * E + r] = df;
            if (r < A) {
                double* Dg = l.Dm + (size_t)gs[u] * AE;
                Dg[r * E + cc] = ds[u];
                if (cc < A) Dg[cc * E + r] = ds[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (!FORCES) break;                                               // kinematic scenarios: no contact forces, only the pair set shrinks
            if (base + u * BLOCK + wave_base >= n1) continue;                 // wave-uniform: this trip of this wave holds no agent pair
            if (gs[u] < 0 || cs[u] >= A) continue;
            // d_min: multiagent/core.py:880 COLLISION_DISTANCE; classic MPE (onpolicy/envs/mpe/core.py:276, 282) size_a + size_b
            const double dmin = c.contact_family ? c.agent_size + c.agent_size : c.sep_dist;
            const double pen = logaddexp0(-(ds[u] - dmin) / c.contact_margin) * c.contact_margin;
            const int slot = gs[u] * A * A + rs[u] * A + cs[u];
            Fx[slot] = c.contact_force * dxs[u] / ds[u] * pen; Fy[slot] = c.contact_force * dys[u] / ds[u] * pen;
        }
    }
    for (int q = tid; q < G * E; q += BLOCK) {                           // diagonal
        const int g = q / E, r = q - g * E;
        if (only_reset && !l.flags[g * 4 + 0]) continue;
        if (r >= A && !statics) continue;
        l.M[(size_t)g * EE4 + r * E + r] = 0.0f;
        if (r < A) l.Dm[(size_t)g * AE + r * E + r] = 0.0;
    }
}

#define SWEEP(var, n) _Pragma("unroll") for (int var = 0; var < (AP ? AP : (n)); ++var)
// Graph outputs of a tile: optional adjacency mask pass (block-wide, with its barrier), then the adj and node_obs
// stores executed by threads t0, t0+nthr, ... (all BLOCK threads, or only the streaming waves of a specialised tile).
template <int BLOCK, int AP, int SC, int FL>
__device__ __forceinline__ void stream_graph_fn(const KParams& p, const gmpe_outputs& o, const Lds& l, const int Gv, const int n0, const int tid,
                                                const int t0, const int nthr, const bool do_mask, const int any_mask) {
    constexpr bool CT = AP > 0 && SC != SC_NAV_WALLS;
    const int A = CT ? AP : p.A, L = CT ? AP : p.L, E = CT ? 2 * AP : p.E, EE = E * E, EE4 = (EE + 3) / 4 * 4;
    const int abl = FL ? 0 : GMPE_ABL(p);
    const bool nt = FL == 1 ? false : p.nt != 0;                         // FL = 1: the steady-state instantiation (step, no ablation, ordinary stores); rollouts: by slot volume
    // ---- 6. adjacency mask (…_july.py:1627-1648): rows/cols of done agents and reached landmarks -> 0.
    // Only tiles that contain such an entity pay for this pass.
    if (do_mask && any_mask && !(abl & 4)) {
        for (int q = tid; q < Gv * EE; q += BLOCK) {
            const int gg = fdiv(q, EE, p.m_EE), rc = q - gg * EE;
            if (!l.flags[gg * 4 + 2]) continue;
            const int r = fdiv(rc, E, p.m_E), cc = rc - r * E;
            if (l.moff[gg * E + r] | l.moff[gg * E + cc]) l.M[(size_t)gg * EE4 + rc] = 0.0f;
        }
    }
    if (do_mask) __syncthreads();
    

    // ---- 7. stream the observations out (16-byte stores wherever the row length allows)
    if (o.adj && !(abl & 1)) {
        const bool vec = (EE & 3) == 0;
        if (o.adj_compact) {
            float* dst = o.adj + (size_t)n0 * EE;
            if (vec) {
                const int nq = EE / 4;
                for (int q = t0; q < Gv * nq; q += nthr) {
                    const int gg = fdiv(q, nq, p.m_nq), m = q - gg * nq;
                    if (l.flags[gg * 4 + 3]) reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(l.M + (size_t)gg * EE4)[m];
                }
            } else for (int q = t0; q < Gv * EE; q += nthr) { const int gg = fdiv(q, EE, p.m_EE); if (l.flags[gg * 4 + 3]) dst[q] = l.M[(size_t)gg * EE4 + (q - gg * EE)]; }
        } else {
            float* dst = o.adj + (size_t)n0 * A * EE;
#ifdef GMPE_NTORDER_KNOB
            if (vec && nt && p.ablate != 200) {                              // experiment build (tools/ntorder.py): GMPE_ABLATE=200 forces the lane-keeps-a-float4 order
#else
            if (vec && nt) {
#endif
                // big graphs (nontemporal stores): OUTPUT order — the tile writes each env's [A,E,E] block front to back, so the
                // workgroups in flight stream whole 0.6-4 MB blocks like a fill (tools/expandbw.hip: 6.9 vs 5.7 TB/s for the
                // lane-keeps-a-float4 order below, whose A copies open A write fronts per tile). Also at small E: decided inside one process on one set
                // of slot buffers (tools/ntorder.py: c2 19.4 vs 19.75 us per step, c3 18.4 vs 19.1) — across processes slot rollouts spread +-5 %
                const int nq = EE / 4, per = A * nq;
                for (int gg = 0; gg < Gv; ++gg) {
                    if (!l.flags[gg * 4 + 3]) continue;
                    const float4* M4 = reinterpret_cast<const float4*>(l.M + (size_t)gg * EE4);
                    float4* d4 = reinterpret_cast<float4*>(dst) + (size_t)gg * per;
                    for (int q = t0; q < per; q += nthr) {
                        const int a = fdiv(q, nq, p.m_nq), m = q - a * nq;      // exact: q * nq < A * nq^2 < 2^32 (checked by gmpe_create)
                        nt_store4(&d4[q], M4[m]);
                    }
                }
            } else if (vec) {
                const int nq = EE / 4;
                // each lane keeps one float4 of the env's matrix and stores it to the A ego copies
                for (int q = t0; q < Gv * nq; q += nthr) {
                    const int gg = fdiv(q, nq, p.m_nq), m = q - gg * nq;
                    if (!l.flags[gg * 4 + 3]) continue;
                    const float4 val = reinterpret_cast<const float4*>(l.M + (size_t)gg * EE4)[m];
                    float4* d4 = reinterpret_cast<float4*>(dst) + (size_t)gg * A * nq + m;
                    SWEEP(a, A) { if (!AP || a < A) { if (nt) nt_store4(&d4[(size_t)a * nq], val); else d4[(size_t)a * nq] = val; } }
                }
            } else {
                const int AEE = A * EE;
                for (int q = t0; q < Gv * AEE; q += nthr) {
                    const int gg = q / AEE, rem = q - gg * AEE, rc = rem % EE;    // rare path (E*E % 4 != 0): plain division
                    if (l.flags[gg * 4 + 3]) dst[q] = l.M[(size_t)gg * EE4 + rc];
                }
            }
        }
    }
    
    STAMP_T(16, 64); STAMP_T(19, 192);                                     // diagnostic build: adjacency part issued (waves 1 / 3)
    if (o.entity_table) {
        // entity table (include/gmpe.h gmpe_outputs.entity_table): the fp64 per-entity state every node_obs row below is a pure function of — what a rank ships
        // to the learner instead of the [A,E,F] rows (gmpe_expand_node_obs rebuilds them there bit for bit)
        const int W = p.TW;
        double* dst = o.entity_table + (size_t)n0 * W;
        for (int q = t0; q < Gv * W; q += nthr) {
            const int gg = fdiv(q, W, p.m_TW), w = q - gg * W;
            if (!l.flags[gg * 4 + 3]) continue;
            double val;
            const int wm = W - (E + 31) / 32;                                // first of the adjacency-mask words (32 node bits per double: exact)
            if (w < 2 * E) val = w < E ? l.ex[gg * E + w] : l.ey[gg * E + (w - E)];
            else if (w >= wm) {
                // this step's adjacency mask (…_july.py:1627-1648: done agents, reached landmarks), so that the learner can rebuild the E x E matrix from the positions
                unsigned bits = 0;
                const int k0 = (w - wm) * 32;
                for (int b = 0; b < 32 && k0 + b < E; ++b) bits |= (l.moff[gg * E + k0 + b] ? 1u : 0u) << b;
                val = (double)bits;
            }
            else if (SC == SC_TWO && w >= wm - 2) val = l.tube[gg * GMPE_TUBE_STRIDE + (w == wm - 2 ? T_EXX : T_EXY)];
            else {
                const int r = w - 2 * E, blk = fdiv(r, A, p.m_A), a = gg * A + (r - blk * A);
                val = blk == 0 ? l.vox[a] : blk == 1 ? l.voy[a] : blk == 2 ? l.vnx[a] : blk == 3 ? l.vny[a] : blk == 4 ? l.cn[a] : l.sn[a];
            }
            dst[q] = val;
        }
    }
    if (sc_rotfam(SC) && o.node_obs && !(abl & 2) && p.c.graph_feat_type != 1) {
        // rot_inv node row (…rot_inv.py:1690-1766): 7 float32 = [rel_vel, rel_pos, rel_goal (all rotated by the ego heading), type].
        // Positions / velocities are rounded to float32 FIRST, differenced in float32, rotated in float64, rounded again.
        // A lane owns one (env, entity) and walks the egos; rows are 28 B, so the stores are scalar — and ordinary: L2 merges the seven dwords of a row into whole lines,
        // with the nontemporal hint they reach HBM as partial writes (26-slot rot_inv rollout 19.3 -> 76 us per step, profiles/r04_notes.md).
        float* base = o.node_obs + (size_t)n0 * A * E * 7;
        for (int sidx = t0; sidx < Gv * E; sidx += nthr) {
            const int gg = fdiv(sidx, E, p.m_E), k = sidx - gg * E;
            if (!l.flags[gg * 4 + 3]) continue;
            const int ab = gg * A, eb = gg * E;
            const bool kag = k < A;
            const int kk = kag ? k : 0;
            const float kx = (float)l.ex[eb + k], ky = (float)l.ey[eb + k];
            const float kvox = kag ? (float)l.vox[ab + kk] : 0.0f, kvoy = kag ? (float)l.voy[ab + kk] : 0.0f;
            const float kvnx = kag ? (float)l.vnx[ab + kk] : 0.0f, kvny = kag ? (float)l.vny[ab + kk] : 0.0f;
            // goal node feature of an agent: its landmark; in two_phase_graph.py:1405 the corridor exit
            const float gxk = SC == SC_TWO ? (float)l.tube[gg * GMPE_TUBE_STRIDE + T_EXX] : (kag ? (float)l.ex[eb + A + kk] : 0.0f);
            const float gyk = SC == SC_TWO ? (float)l.tube[gg * GMPE_TUBE_STRIDE + T_EXY] : (kag ? (float)l.ey[eb + A + kk] : 0.0f);
            const float typ = kag ? 0.0f : (k < A + L ? 1.0f : 2.0f);
            for (int ei = 0; ei < A; ++ei) {
                const float apx = (float)l.ex[eb + ei], apy = (float)l.ey[eb + ei];
                const float avx = (float)l.vnx[ab + ei], avy = (float)l.vny[ab + ei];   // vn == vo unless the ego reached its goal in this step (section 2)
                const double cs = l.cn[ab + ei], sn = l.sn[ab + ei];          // ego heading AFTER its own reward
                const bool post = k <= ei;                                    // vn[k] == vo[k] unless k reached its goal in this step
                const float rvx = (post ? kvnx : kvox) - avx, rvy = (post ? kvny : kvoy) - avy;
                const float rpx = kx - apx, rpy = ky - apy;
                double o0, o1, o2, o3, o4, o5;
                rot2(cs, sn, (double)rvx, (double)rvy, o0, o1);
                rot2(cs, sn, (double)rpx, (double)rpy, o2, o3);
                if (kag) rot2(cs, sn, (double)(gxk - apx), (double)(gyk - apy), o4, o5); else { o4 = o2; o5 = o3; }
                float* dst = base + ((size_t)(gg * A + ei) * E + k) * 7;
                dst[0] = (float)o0; dst[1] = (float)o1; dst[2] = (float)o2; dst[3] = (float)o3;
                dst[4] = (float)o4; dst[5] = (float)o5; dst[6] = typ;
            }
        }
    }
    if (o.node_obs && !(abl & 2) && p.c.graph_feat_type == 1) {
        // graph_feat_type 'global' (_get_entity_feat_global, …_july.py:1672-1691; the same function in the rot_inv family's files): row (ego, entity k) = 7 floats [vel, pos, goal, type] in
        // world coordinates — the same for every ego except that ego i sees agent k's re-drawn velocity iff k reached its goal in this
        // step and k <= i (ordered-visibility rule). A lane owns one (env, entity) and walks the egos; 28-byte rows: scalar stores.
        float* base = o.node_obs + (size_t)n0 * A * E * 7;
        for (int sidx = t0; sidx < Gv * E; sidx += nthr) {
            const int gg = fdiv(sidx, E, p.m_E), k = sidx - gg * E;
            if (!l.flags[gg * 4 + 3]) continue;
            const int ab = gg * A, eb = gg * E;
            const bool kag = k < A;
            const int kk = kag ? k : 0;
            const float kx = (float)l.ex[eb + k], ky = (float)l.ey[eb + k];
            const float kvox = kag ? (float)l.vox[ab + kk] : 0.0f, kvoy = kag ? (float)l.voy[ab + kk] : 0.0f;
            const float kvnx = kag ? (float)l.vnx[ab + kk] : 0.0f, kvny = kag ? (float)l.vny[ab + kk] : 0.0f;
            const float gx = kag ? (float)l.ex[eb + A + kk] : kx, gy = kag ? (float)l.ey[eb + A + kk] : ky;
            const float typ = kag ? 0.0f : (k < A + L ? 1.0f : 2.0f);
            for (int ei = 0; ei < A; ++ei) {
                const bool post = k <= ei;                                    // vn[k] == vo[k] unless k reached its goal in this step (section 2)
                float* dst = base + ((size_t)(gg * A + ei) * E + k) * 7;
                dst[0] = post ? kvnx : kvox; dst[1] = post ? kvny : kvoy; dst[2] = kx; dst[3] = ky; dst[4] = gx; dst[5] = gy; dst[6] = typ;
            }
        }
    }
    if (!sc_rotfam(SC) && o.node_obs && !(abl & 2) && p.c.graph_feat_type != 1) {
        // node row (ego, entity k) = 2 float4: [rel_vel, rel_pos] and [rel_goal, occupied, type].
        // A lane owns one (env, entity, half) slot, keeps that entity's data in registers and walks the egos.
        float4* base = reinterpret_cast<float4*>(o.node_obs + (size_t)n0 * A * E * GMPE_NODE_FEATS);
        const int E2 = 2 * E;
        for (int sidx = t0; sidx < Gv * E2; sidx += nthr) {
            const int gg = fdiv(sidx, E2, p.m_2E), rem = sidx - gg * E2;
            if (!l.flags[gg * 4 + 3]) continue;
            const int k = rem >> 1, half = rem & 1;
            const int ab = gg * A, eb = gg * E;
            const double kx = l.ex[eb + k], ky = l.ey[eb + k];
            const bool kag = k < A;
            const int kk = kag ? k : 0;
            // velocities: section 2 leaves vn == vo unless the agent reached its goal in this step, so "the ego's own velocity after its reward" is
            // vn[ego], and "agent k's velocity as ego sees it" (re-drawn iff k reached the goal and k <= ego) is k <= ego ? vn[k] : vo[k]
            const double kvox = kag ? l.vox[ab + kk] : 0.0, kvoy = kag ? l.voy[ab + kk] : 0.0;
            const double kvnx = kag ? l.vnx[ab + kk] : 0.0, kvny = kag ? l.vny[ab + kk] : 0.0;
            const double gx = kag ? l.ex[eb + A + kk] : kx, gy = kag ? l.ey[eb + A + kk] : ky;
            const float occ = kag ? 0.0f : 1.0f, typ = kag ? 0.0f : (k < A + L ? 1.0f : 2.0f);
            float4* dst = base + (size_t)gg * A * E2 + rem;
            SWEEP(ei, A) {
                const bool ok = !AP || ei < A;
                const int ec = ok ? ei : 0;
                const double px = l.ex[eb + ec], py = l.ey[eb + ec];
                float4 val;
                if (half == 0) {
                    const double evx = l.vnx[ab + ec], evy = l.vny[ab + ec];
                    const bool post = k <= ec;
                    val = make_float4((float)((post ? kvnx : kvox) - evx), (float)((post ? kvny : kvoy) - evy), (float)(kx - px), (float)(ky - py));
                } else {
                    val = make_float4((float)(gx - px), (float)(gy - py), occ, typ);
                }
                if (ok) { if (nt) nt_store4(&dst[(size_t)ec * E2], val); else dst[(size_t)ec * E2] = val; }
            }
        }
    }
}

#ifndef GMPE_BATCHED_PLACEMENT
#define GMPE_BATCHED_PLACEMENT 1    /* exact-size navigation_graph / July rollout kernels place a resetting env's entities in batches of A attempts (reset_world_coop<SC, true>) */
#endif
#ifndef GMPE_MIN_WAVES
#define GMPE_MIN_WAVES 1
#endif
#ifndef GMPE_MIN_WAVES_NOWALLS
#define GMPE_MIN_WAVES_NOWALLS 1
#endif

// ---------------------------------------------------------------- the fused kernel
// AP > 0: compile-time bound (A, L <= AP) for the per-agent sweeps, so they unroll fully and every LDS read of a
// sweep is issued before the first use (the sweeps are latency-bound: one wave per SIMD, ~100-cycle LDS reads).
// Out-of-range iterations read a clamped index and are masked in the arithmetic. AP == 0: run-time bounds.
// SC: scenario variant (above). Only SC_NAV_WALLS compiles the wall-contact code (asin / cos / softplus inside the agent lane's dynamics) out: it is
// the single largest consumer of registers (187 -> 135 VGPRs), i.e. 2 -> 3 waves per SIMD for wall-less worlds.
// FL = 1: the steady-state instantiation — step mode, wave specialisation on, ordinary stores, no ablation — with those run-time
// flags folded (selected by the host when they hold; everything else takes FL = 0).
// Register budgets of the step kernels (launch bounds below): four waves per SIMD for the exact-size ones except the rot_inv family, which needs ~131-137 VGPRs with run-time G (three waves); with
// GC = 4 rot_inv and three_phase fit 128 (one dword / nothing spilled) and run four tiles per CU — closed loop 24.5 -> 23.1 and 25.1 -> 23.3 us per step — while two_phase would spill ten dwords (25.1 -> 28.3) and stays at three.
// GC > 0: envs per tile known at compile time too (round 4; the exact-size rollout / steady-state instantiations at the tile shapes gmpe_create picks for 4096 x 10:
// G = 4 and 6). Every LDS array base of the carve and every `g * E` then folds into an instruction offset instead of living in one of ~34 SGPRs or being recomputed
// per use: navigation_graph's rollout kernel 128 -> 110 VGPRs with no scratch left, SGPR spills 130 -> 86, 5 % fewer instructions (profiles/r04_notes.md).
template <int BLOCK, int AP, int SC, int FL, int GC = 0>
__global__ __launch_bounds__(BLOCK, (SC == SC_NAV_WALLS ? GMPE_MIN_WAVES : (FL == 2 && BLOCK > 64 ? ((sc_rotfam(SC) || (AP > 0 && sc_kinematic(SC))) ? 3 : 4) : (AP > 0 && BLOCK > 64 ? ((sc_rotfam(SC) && !(GC > 0 && SC != SC_TWO)) ? 3 : 4) : GMPE_MIN_WAVES_NOWALLS)))) void k_env(const KParams p_arg) {
    const KParams& p = p_arg;
    constexpr bool WALLS = SC == SC_NAV_WALLS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {   // Touch every 64-byte line of the 1.1 KB kernarg segment at once: the compiler fetches kernel parameters right before each
        // use and waits for them one by one; after this batch those scalar loads hit the scalar cache.
        const int __attribute__((address_space(4)))* ka = (const int __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        int acc = 0;
#pragma unroll
        for (int q = 0; q < (int)((sizeof(KParams) + 63) / 64); ++q) acc ^= ka[q * 16];
        asm volatile("" :: "s"(acc));
    }
    const int tid = threadIdx.x;
    // exact-size instantiations (the host selects AP only when A == L == AP, O == 0) know A, L, E, D at compile time. The rot_inv
    // family needs ~130-150 VGPRs for that and is compiled for 3 waves per SIMD (the tile-shape search then packs 6 envs per tile);
    // the walls variant would lose a wave and keeps run-time sizes.
    constexpr bool CT = AP > 0 && SC != SC_NAV_WALLS;     // host: AP > 0 only if A == L == AP and O == 0
    const int A = CT ? AP : p.A, L = CT ? AP : p.L, O = CT ? 0 : p.O, E = CT ? 2 * AP : p.E, D = CT ? (SC == SC_JULY ? 19 : (sc_phasefam(SC) ? 15 : 13)) : p.D, G = GC > 0 ? GC : p.G, N = p.c.num_envs;
    const gmpe_config& c = p.c;
    const Lds l = carve(smem, G, A, E, D, SC == SC_NAV_WALLS ? p.c.num_walls : 0, p.nfuse);
    const int n0 = p.env_lo + blockIdx.x * G;
    const int Gv = min(G, p.env_hi - n0);                               // envs actually present in this tile
    constexpr bool july = SC == SC_JULY, rotinv = SC == SC_ROT, rotfam = sc_rotfam(SC), two = SC == SC_TWO, three = SC == SC_THREE;
    constexpr int PV = two ? 1 : (three ? 2 : 0);                        // phase FSM variant (gmpe_device.h)
    const bool step = FL ? true : p.mode == MODE_STEP;
    constexpr bool kin = sc_kinematic(SC);
    constexpr bool BATCHK = FL == 2 && AP > 0 && BLOCK > 64 && !sc_rotfam(SC) && SC != SC_NAV_WALLS && GMPE_BATCHED_PLACEMENT;   // batched placement from prefilled draws (reset_world_coop<SC, true>)
    const int EE = E * E, EE4 = (EE + 3) / 4 * 4, AD4 = (A * D + 3) / 4 * 4;
    const double INF = __builtin_huge_val();

    // agent lane mapping: lane tid of wave 0 = (env g, agent i)
    const int g = fdiv(tid, A, p.m_A), i = tid - g * A;
    const int n = n0 + g;
    bool ag = tid < Gv * A;
    if (ag && !step && p.mask && !p.mask[n]) ag = false;               // explicit reset: masked-out env
    const size_t na = (size_t)n * A + i;
    const Lds v = env_view(l, ag ? g : 0, A, E, D);

    // ---- per-agent registers
    int prev_phase = 0, phase_reached = 0, cooldown = 0;
    double p_dist = 0, tim = 0;
    int cur_step = 0, act_idx = 0;
    int err = 0;
    int64_t ctr0 = 0;

    STAMP(0);
    // ---- 0. load state. Every global read of the step is ISSUED here before any of them is consumed
    // (first iteration of each staging loop hoisted into registers), so the tile pays one HBM round trip.
    const bool t_ok = tid < Gv * GMPE_TUBE_STRIDE, l_ok = tid < Gv * L, o_ok = tid < Gv * O;
    double tube0 = 0, lmx0 = 0, lmy0 = 0, obx0 = 0, oby0 = 0;
    if (t_ok) tube0 = p.s.tube[(size_t)n0 * GMPE_TUBE_STRIDE + tid];
    if (l_ok) { lmx0 = p.s.landmarks[((size_t)n0 * L + tid) * 2]; lmy0 = p.s.landmarks[((size_t)n0 * L + tid) * 2 + 1]; }
    if (o_ok) { obx0 = p.s.obstacles[((size_t)n0 * O + tid) * 2]; oby0 = p.s.obstacles[((size_t)n0 * O + tid) * 2 + 1]; }
    // Info counters (read only by sections 3+4) are fetched by "loader" lanes — the first lanes of wave 1 in multi-wave tiles — and
    // staged in LDS: wave 0, whose instruction stream is the tile's critical path, issues 13 fewer loads and holds 14 fewer registers.
    const int lt = BLOCK > 64 ? tid - 64 : tid;
    const bool loader = step && lt >= 0 && lt < Gv * A;
    int c9[9] = {-1, -1, -1, -1, 0, 0, 0, 0, 0}; double cd0 = 0, cd1 = 0, cds = 0;
    if (loader) {
        const size_t nl = (size_t)n0 * A + lt;
        c9[0] = p.s.times_required[nl]; c9[1] = p.s.dists_to_goal[nl]; c9[2] = p.s.dist_left[nl]; c9[3] = p.s.goal_reached[nl];
        c9[4] = p.s.n_agent_coll[nl]; c9[5] = p.s.n_obst_coll[nl]; c9[6] = p.s.spacing_viol[nl]; c9[7] = p.s.steps_in_corr[nl];
        c9[8] = p.s.conformance[nl];
        cd0 = p.s.goal_min_time[nl];
        if (SC == SC_ROT) cd1 = p.s.prev_proj[nl];
        const int lg = fdiv(lt, A, p.m_A);
        if (lt == lg * A) cds = p.s.delta_spacing[n0 + lg];
    }
    double x0 = 0, y0 = 0, a20 = 0, a30 = 0; int st0 = 0, gt0 = -1;
    if (ag) {
        prev_phase = p.s.prev_phase[na];
        cur_step = p.s.current_step[n];
        ctr0 = p.s.rng_ctr[n];
        if (step) {
            x0 = p.s.x[na]; y0 = p.s.y[na]; a20 = p.s.s2[na]; a30 = p.s.s3[na];
            st0 = p.s.status[na]; gt0 = p.s.goal_tracker[na];
            phase_reached = p.s.phase_reached[na]; cooldown = p.s.cooldown[na];
            p_dist = p.s.p_dist[na]; tim = p.s.time[na];
            if (p.act) act_idx = p.act[na];
            else {                                                      // np.argmax: first maximum
                const float* oh = p.onehot + na * c.n_actions;
                float best = oh[0];
                for (int q = 1; q < c.n_actions; ++q) { const float x = oh[q]; if (x > best) { best = x; act_idx = q; } }
            }
        }
    }
    for (int q = tid; q < G * E; q += BLOCK) l.moff[q] = 0;
    for (int q = tid; q < A * A; q += BLOCK) {                          // agent-pair table (a < k), row-major triangular order
        const int a = fdiv(q, A, p.m_A), k = q - a * A;
        if (k > a) l.ptab[a * A - a * (a + 1) / 2 + (k - a - 1)] = (a << 8) | k;
    }
    if (tid < G) {
        const int nn = n0 + tid;
        const bool active = nn < p.env_hi && (step || !p.mask || p.mask[nn]);
        l.flags[tid * 4 + 0] = (!step && active); l.flags[tid * 4 + 1] = 0; l.flags[tid * 4 + 2] = 0; l.flags[tid * 4 + 3] = active;
    }
    if (t_ok) l.tube[tid] = tube0;
    for (int q = tid + BLOCK; q < Gv * GMPE_TUBE_STRIDE; q += BLOCK) l.tube[q] = p.s.tube[(size_t)n0 * GMPE_TUBE_STRIDE + q];
    if (l_ok) { const int gg = fdiv(tid, L, p.m_L), k = tid - gg * L; l.ex[gg * E + A + k] = lmx0; l.ey[gg * E + A + k] = lmy0; }
    for (int q = tid + BLOCK; q < Gv * L; q += BLOCK) {
        const int gg = fdiv(q, L, p.m_L), k = q - gg * L;
        l.ex[gg * E + A + k] = p.s.landmarks[((size_t)n0 * L + q) * 2]; l.ey[gg * E + A + k] = p.s.landmarks[((size_t)n0 * L + q) * 2 + 1];
    }
    if (o_ok) { const int gg = fdiv(tid, O, p.m_O), k = tid - gg * O; l.ex[gg * E + A + L + k] = obx0; l.ey[gg * E + A + L + k] = oby0; }
    for (int q = tid + BLOCK; q < Gv * O; q += BLOCK) {
        const int gg = fdiv(q, O, p.m_O), k = q - gg * O;
        l.ex[gg * E + A + L + k] = p.s.obstacles[((size_t)n0 * O + q) * 2]; l.ey[gg * E + A + L + k] = p.s.obstacles[((size_t)n0 * O + q) * 2 + 1];
    }
    if (loader) {
#pragma unroll
        for (int k = 0; k < 9; ++k) l.cnt[lt * 9 + k] = c9[k];
        l.cntd[lt * 2] = cd0; l.cntd[lt * 2 + 1] = cd1;
        const int lg = fdiv(lt, A, p.m_A);
        if (lt == lg * A) l.cntd[(size_t)G * A * 2 + lg] = cds;
    }
    if (ag && step) {
        v.ex[i] = x0; v.ey[i] = y0; v.s2[i] = a20; v.s3[i] = a30; v.s_old[i] = st0; v.gt[i] = gt0;
        act_idx = act_idx < 0 ? 0 : (act_idx >= c.n_actions ? c.n_actions - 1 : act_idx);
    }
    __syncthreads();
    STAMP(1);

    // FL == 2 (gmpe_rollout_steps): the K steps of an open-loop rollout run inside this launch. A tile's step k+1 depends only on the
    // tile's own state, which stays in LDS / registers between steps (no state reload, write-back after the last step only); the
    // graph stores of step k drain while the tile — and the other tiles of the CU, which drift out of phase — run step k+1's
    // latency chain. Every other instantiation runs the body once.
    constexpr bool ROLL = FL == 2;
    // the instantiations that can run distance_force_pass: exact-size rollout tiles (A = L = AP, no obstacles); navigation_graph fuses the next step's force pass into it
    constexpr bool FUSE_OK = ROLL && SC != SC_NAV_WALLS && AP > 0 && BLOCK > 64;
    const bool FUSE = FUSE_OK && p.rowpairs != 0 && (kin || p.nfuse > 0);
    const int K = ROLL ? p.K : 1;
    int slot = ROLL ? p.first_slot : 0, aset = 0;
    const int tid_o = tid; const bool ag_o = ag;
    int kk = 0;
    do {                                                                    // `while (ROLL && ...)`: no loop at all in the other instantiations
        // Rollouts hide the lane index from the optimiser once per step: everything below is a function of it, and loop-invariant
        // code motion would otherwise hoist a whole step's worth of index / address arithmetic out of the loop and keep it in
        // registers (256 VGPRs instead of ~125).
        int tid_k = tid_o;
        const KParams __attribute__((address_space(4)))* pk = (const KParams __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        if (ROLL) { asm volatile("" : "+v"(tid_k)); asm volatile("" : "+s"(pk)); }   // same for the kernel parameters: uniform fp64 expressions of the config live in VGPR pairs
        const int tid = tid_k;
        const KParams& p = ROLL ? *(const KParams*)pk : p_arg;              // the kernel's only argument sits at offset 0 of the kernarg segment
        const gmpe_config& c = p.c;
        const int g = fdiv(tid, A, p.m_A), i = tid - g * A;
        const int n = n0 + g;
        const bool ag = ROLL ? tid < Gv * A : ag_o;
        const size_t na = (size_t)n * A + i;
        const Lds v = env_view(l, ag ? g : 0, A, E, D);
        const unsigned long long emask = ag ? ((A >= 64 ? ~0ull : ((1ull << A) - 1ull)) << (g * A)) : 0ull;
        gmpe_outputs out_k;                                                 // rollouts: this step's slot of every output
        const gmpe_outputs& out = ROLL ? out_k : p.o;
        int act_next = 0;
        if (ROLL) {
            // this step's output slot, and the next step's actions (issued now, consumed at the top of the next iteration)
            out_k = p.o;
            if (out_k.obs) out_k.obs += (size_t)slot * p.st_obs;
            if (out_k.agent_id) out_k.agent_id += (size_t)slot * p.st_id;
            if (out_k.node_obs) out_k.node_obs += (size_t)slot * p.st_node;
            if (out_k.adj) out_k.adj += (size_t)slot * p.st_adj;
            if (out_k.reward) out_k.reward += (size_t)slot * p.st_rew;
            if (out_k.done) out_k.done += (size_t)slot * p.st_done;
            if (out_k.info) out_k.info += (size_t)slot * p.st_info;
            if (out_k.entity_table) out_k.entity_table += (size_t)slot * p.st_tab;
            const int anext = aset + 1 == p.S ? 0 : aset + 1;
            if (ag && kk + 1 < K) act_next = p.act[(size_t)anext * N * A + na];
            aset = anext;
        }
        int ph1 = 0, cp = 0, prevA = 0, ndraw = 0;
        bool goal_branch = true, done = false, all_done = false;
        double dgoal = 0, rew = 0;
        int trq = -1, dtg = -1, dleft = -1, greached = -1, nac = 0, noc = 0, sv = 0, sic = 0, conf = 0;   // info counters: staged in LDS, live inside one step only
        double gmt = 0, dsp0 = 0, pproj = 0;
        if (step) {
            cur_step += 1;
            const int C = A + O;                                            // colliders: agents + obstacles (landmarks collide=False)
            double* Fx = FUSE ? l.F2 : reinterpret_cast<double*>(l.M);      // pair forces alias the (not yet built) fp32 matrix, except in fused rollouts
            double* Fy = Fx + (size_t)G * A * C;
            if (!kin && !(FUSE && kk > 0)) {                                // fused rollouts: the previous step's distance pass left this step's forces in F2
                // ---- 1a. contact forces, one PAIR per lane (get_entity_collision_force core.py:872-906): pair (a, k>a)
                // computed once with delta = pos[a]-pos[k]; side a gets +F, side k gets -F (summed in 1b).
                const int NP = A * (A - 1) / 2, W = NP + A * O;            // valid pairs only: agent pairs (a<k) + agent x obstacle
                // One pair per lane and trip; only the waves that still own a pair run another trip (C2: 270 pairs on 256 lanes —
                // the 14-pair tail costs one wave, not four). Branch-free: nearly every wave holds a pair inside the softplus
                // range; far pairs get pen = log1p(exp(-large)) = 0 like in the reference.
                for (int q = tid; q < Gv * W; q += BLOCK) {
                    if (!FL && (GMPE_ABL(p) & 8)) { Fx[q] = 0.0; Fy[q] = 0.0; continue; }   // diagnostic build only
                    const int gg = fdiv(q, W, p.m_FW), w = q - gg * W;
                    const bool apair = w < NP;
                    const int pk = l.ptab[apair ? w : 0];
                    const int t = apair ? 0 : w - NP;
                    const int ao = fdiv(t, O, p.m_O);
                    const int a = apair ? (pk >> 8) : ao, kk = apair ? (pk & 255) : A + (t - ao * O);
                    const int k = kk < A ? kk : L + kk;                     // entity index of collider kk
                    const double dx = l.ex[gg * E + a] - l.ex[gg * E + k], dy = l.ey[gg * E + a] - l.ey[gg * E + k];
                    const double dist = sqrt(dx * dx + dy * dy);
                    // d_min: multiagent/core.py:880 COLLISION_DISTANCE; classic MPE (onpolicy/envs/mpe/core.py:276, 282) size_a + size_b
                    const double dmin = c.contact_family ? c.agent_size + (kk < A ? c.agent_size : c.collider_size) : c.sep_dist;
                    const double pen = logaddexp0(-(dist - dmin) / c.contact_margin) * c.contact_margin;
                    const int slot = gg * A * C + a * C + kk;
                    Fx[slot] = c.contact_force * dx / dist * pen; Fy[slot] = c.contact_force * dy / dist * pen;
                }
                if (WALLS) {
                    // wall contact forces (get_wall_collision_force core.py:909-964), one (agent, wall) per lane: the asin / cos /
                    // softplus code lives here instead of in the agent lane's dynamics (187 -> 141 VGPRs for the walls variant)
                    const int NW = c.num_walls;
                    for (int q = tid; q < Gv * A * NW; q += BLOCK) {
                        const int slot = q / NW, w = q - slot * NW;         // slot = tile-level agent index (gg*A + a)
                        const int gg = fdiv(slot, A, p.m_A), a = slot - gg * A;
                        double wx = 0.0, wy = 0.0;
                        if (!wall_force(c.walls[w], l.ex[gg * E + a], l.ey[gg * E + a], c.contact_family ? c.agent_size : c.entity_size, c.wall_contact_force, c.wall_contact_margin, wx, wy)) { wx = 0.0; wy = 0.0; }
                        l.fw[(size_t)slot * 2 * NW + 2 * w] = wx; l.fw[(size_t)slot * 2 * NW + 2 * w + 1] = wy;   // None -> +0.0: x + 0.0 == x
                    }
                }
                __syncthreads();
            }
            STAMP(2);
            // ---- 1b. action decode + dynamics
            double nx = 0, ny = 0, nv2 = 0, nv3 = 0;
            if (ag) {
                double u0, u1; decode_action<SC>(c, act_idx, u0, u1);
                if (p.ovr && (!p.ovr_use || p.ovr_use[na])) { u0 = p.ovr[na * 2]; u1 = p.ovr[na * 2 + 1]; }   // safety-filter hook slot (core.py:692-736)
                nx = v.ex[i]; ny = v.ey[i]; nv2 = v.s2[i]; nv3 = v.s3[i];
                if (kin) {
                    if (!v.s_old[i]) {                                      // update_agent_state core.py:819-826
                        const double dt = c.dt, th0 = nv2, v0 = nv3;
                        const double th1 = th0 + u0 * dt, v1 = v0 + u1 * dt;
                        if (u0 != 0.0) {
                            double s0, c0, s1, c1; sincos(th0, &s0, &c0); sincos(th1, &s1, &c1);
                            nx += (v1 * s1 - v0 * s0) / u0 + u1 * (c1 - c0) / (u0 * u0);
                            ny += (-v1 * c1 + v0 * c0) / u0 + u1 * (s1 - s0) / (u0 * u0);
                        } else {
                            const double d = (v0 + 0.5 * u1 * dt) * dt;
                            double s0, c0; sincos(th0, &s0, &c0);
                            nx += d * c0; ny += d * s0;
                        }
                        double vv = v1;
                        if (vv > c.v_max) vv = c.v_max;
                        if (vv < c.v_min) vv = c.v_min;
                        nv2 = th1; nv3 = vv;
                        p_dist += vv * dt; tim += dt;
                    }
                } else {
                    // force path core.py:766-845: accumulate in the reference's order for this agent — other
                    // entities by ascending index (side b below its own index, side a above), then walls.
                    const bool classic = c.contact_family != 0;             // onpolicy/envs/mpe/core.py constant family (gmpe_config)
                    const double fsc = classic ? c.action_force_scale : 1.0; // apply_action_force: mass * accel (or mass)
                    double sx = fsc * u0, sy = fsc * u1;
                    const double* fxg = Fx + (size_t)g * A * C; const double* fyg = Fy + (size_t)g * A * C;
                    const bool ego_live = classic || v.s_old[i] == 0;       // done side gets no agent-agent force (899-900); classic MPE has no such rule
                    SWEEP(k, A) {
                        const bool ok = !AP || k < A;
                        const int kc = ok ? k : 0;
                        const bool below = kc < i;
                        const int idx = below ? kc * C + i : i * C + kc;
                        const double fx = fxg[idx], fy = fyg[idx];
                        const bool use = ok && ego_live && kc != i && (fx != 0.0 || fy != 0.0);
                        sx = use ? ((below ? -fx : fx) + sx) : sx;
                        sy = use ? ((below ? -fy : fy) + sy) : sy;
                    }
                    for (int o = 0; o < O; ++o) {                           // immovable obstacles push regardless of status
                        const double fx = fxg[i * C + A + o], fy = fyg[i * C + A + o];
                        if (fx != 0.0 || fy != 0.0) { sx = fx + sx; sy = fy + sy; }
                    }
                    if (WALLS) {
                        const double* fwi = l.fw + (size_t)(g * A + i) * 2 * c.num_walls;
                        for (int w = 0; w < c.num_walls; ++w) { sx = sx + fwi[2 * w]; sy = sy + fwi[2 * w + 1]; }
                    }
                    double vx = nv2 * (1 - c.damping), vy = nv3 * (1 - c.damping);
                    if (classic) { vx += (sx / c.agent_mass) * c.dt; vy += (sy / c.agent_mass) * c.dt; }
                    else { vx += (sx / 1.0) * c.dt; vy += (sy / 1.0) * c.dt; }
                    if (c.max_speed > 0) {
                        const double sp = sqrt(vx * vx + vy * vy);
                        if (sp > c.max_speed) { vx = vx / sp * c.max_speed; vy = vy / sp * c.max_speed; }
                    }
                    nv2 = vx; nv3 = vy;
                    nx += vx * c.dt; ny += vy * c.dt;
                    const double ax = vx * c.dt, ay = vy * c.dt;
                    p_dist += sqrt(ax * ax + ay * ay); tim += c.dt;
                }
                v.ex[i] = nx; v.ey[i] = ny; v.s2[i] = nv2; v.s3[i] = nv3;   // nobody reads positions between 1a and here
            }
            __syncthreads();
            STAMP(3);
            if constexpr (FUSE_OK) {
                if (FUSE) distance_force_pass<BLOCK, AP, !kin>(p, l, G, Gv, tid, false, kk == 0);   // + the next step's contact forces
                else distance_pass<BLOCK, (CT ? AP : 0)>(p, l, Gv, tid, false);
            } else distance_pass<BLOCK, (CT ? AP : 0)>(p, l, Gv, tid, false);             // post-move rows: obs, reward, info, adj all read these
            __syncthreads();
            STAMP(4);

            // ---- 2. phase FSM + who newly reaches the goal (depends only on own data: SURVEY §8a)
            prevA = prev_phase;
            if (ag) {
                const double px = v.ex[i], py = v.ey[i];
                double vx, vy; vel_of<SC>(v.s2[i], v.s3[i], vx, vy);
                v.vox[i] = vx; v.voy[i] = vy;
                if (july) {
                    ph1 = phase_eval(v.tube, px, py, prev_phase, prevA);     // observation's call (:1447)
                    if (cooldown > 0) cooldown -= 1;
                    int prevB;
                    cp = phase_eval(v.tube, px, py, prevA, prevB);           // reward's call (:1113)
                    if (cooldown > 0) cooldown -= 1;
                    prevA = prevB;
                    goal_branch = (cp == 2 && phase_reached != 0);
                } else if (rotfam) {
                    // rot_inv.py:675-739: the query mutates only the cooldown, so observation's and reward's calls agree;
                    // phase 2 is only returned with phase_reached >= 1, hence the goal block runs iff cp == 2 (:1281-1297).
                    // three_phase has no demotion either (:1110); in two_phase the agent finishes at the 1 -> 2 transition.
                    ph1 = phase_eval_rot<PV>(v.tube, px, py, prev_phase, phase_reached);
                    if (cooldown > 0) cooldown -= 1;
                    if (cooldown > 0) cooldown -= 1;
                    cp = ph1;
                    goal_branch = (cp == 2) && (rotinv ? phase_reached >= 1 : true);
                }
                dgoal = v.Dm[(size_t)i * E + A + i];
                if (two) v.newf[i] = cp == 2 && prev_phase == 1 && phase_reached == 1 && !v.s_old[i];   // two_phase_graph.py:1023-1044
                else v.newf[i] = goal_branch && dgoal < c.goal_thresh && !v.s_old[i];
            }
            // rank of each newly-reached agent among its env's: heading re-draws follow agent order (core.py:328)
            if (tid < 64) {
                const unsigned long long bal = __ballot(ag && v.newf[i]);
                if (ag) {
                    if (v.newf[i]) {
                        const int rank = __popcll(bal & emask & ((1ull << tid) - 1ull));
                        if (kin) { v.n2[i] = 0.0 + (2 * M_PI - 0.0) * draw_at(c, p.s, n, ctr0 + rank, err); v.n3[i] = c.v_min; }
                        else { v.n2[i] = 0.0; v.n3[i] = 0.0; }
                        if (!two && !three) v.gt[i] = i;                          // the phase-graph files never write goal_tracker (three_phase_graph.py:1119)
                        double vx, vy; vel_of<SC>(v.n2[i], v.n3[i], vx, vy);
                        v.vnx[i] = vx; v.vny[i] = vy;
                    } else { v.n2[i] = v.s2[i]; v.n3[i] = v.s3[i]; v.vnx[i] = v.vox[i]; v.vny[i] = v.voy[i]; }
                    if (rotfam) { double sn_, cn_; sincos(v.n2[i], &sn_, &cn_); v.cn[i] = cn_; v.sn[i] = sn_; }
                    ndraw = kin ? __popcll(bal & emask) : 0;                  // draws consumed (DI reset_velocity draws none)
                    if (i == 0) v.flags[1] = ndraw;
                    // done flag (_get_done environment.py:264-271) and this step's adjacency mask (…_july.py:1627-1648:
                    // done agents, reached landmarks) depend only on status / goal_tracker: known before the rewards
                    const bool st_now = v.s_old[i] || v.newf[i];
                    done = st_now || cur_step >= c.episode_length;
                    v.moff[i] = st_now ? 1 : 0; v.moff[A + i] = (v.gt[i] == i) ? 1 : 0;
                }
                // all agents of an env done -> the worker resets it (env_wrappers.py:865-870)
                const unsigned long long dbal = __ballot(ag && done);
                const unsigned long long mbal = __ballot(ag && (v.moff[i] | v.moff[A + i]));
                all_done = ag && ((dbal & emask) == emask);
                if (ag && i == 0) { v.flags[0] = all_done; v.flags[2] = (mbal & emask) != 0ull; }
                if (BATCHK && all_done && i == 0) reinterpret_cast<long long*>(l.cntd + (size_t)G * A * 2 + G)[g] = ctr0 + ndraw;   // where the reset's draws start (read by the streaming waves below)
            }
            __syncthreads();
            STAMP(5);
        }


        int any_reset = 0, any_mask = 0;
        for (int gg = 0; gg < Gv; ++gg) { any_reset |= l.flags[gg * 4 + 0]; any_mask |= l.flags[gg * 4 + 2]; }   // block-uniform
        const int abl = FL ? 0 : GMPE_ABL(p);

        // Common case (no env of the tile resets): issue the 22 KB/env of graph stores FIRST, then do the reward /
        // info arithmetic under the HBM write drain. Tiles with a reset need the terminal reward/info before the
        // reset overwrites the LDS state, so they keep the reference's order.
        const bool early = step && !any_reset;
        // Multi-wave tiles specialise: wave 0 (all agent lanes) does reward / info / write-back while waves 1.. stream the
        // graph observations, so the ~7 us of per-agent arithmetic runs beside the store issue instead of after it.
        const bool spec = early && BLOCK > 64 && (FL ? true : p.spec != 0);
        if (early && !spec) stream_graph_fn<BLOCK, AP, SC, FL>(p, out, l, Gv, n0, tid, tid, BLOCK, true, any_mask);
        if constexpr (BATCHK) if (any_reset && tid >= 64) {
            // batched placement (reset_world_coop<SC, true>): the streaming waves, idle in a step with a reset, evaluate the resetting envs' draws while wave 0 computes the
            // terminal rewards / info; the barrier in front of section 5 publishes them
            const int ND = EE4 / 2;
            for (int q = tid - 64; q < Gv * ND; q += BLOCK - 64) {
                const int gg = q / ND, d = q - gg * ND;
                if (!l.flags[gg * 4 + 0]) continue;
                const int64_t c0 = reinterpret_cast<const long long*>(l.cntd + (size_t)G * A * 2 + G)[gg];
                reinterpret_cast<double*>(l.M + (size_t)gg * EE4)[d] = draw_raw(p, n0 + gg, c0 + d);
            }
        }
        if (spec) {
            if (any_mask && !(abl & 4)) {
                for (int q = tid; q < Gv * EE; q += BLOCK) {
                    const int gg = fdiv(q, EE, p.m_EE), rc = q - gg * EE;
                    if (!l.flags[gg * 4 + 2]) continue;
                    const int r = fdiv(rc, E, p.m_E), cc = rc - r * E;
                    if (l.moff[gg * E + r] | l.moff[gg * E + cc]) l.M[(size_t)gg * EE4 + rc] = 0.0f;
                }
            }
            __syncthreads();
        }
        if (step) {
            if (!spec || tid < 64) {
                // ---- sections 3+4 (obs, reward, info, write-back). In specialised tiles only wave 0 gets here and the
                // block barrier between the two sections is replaced by wave-local ordering.
                const bool block_sync = !spec;
                // ---- 3. obs, reward, done (ego i sees agent k done iff s_old[k] || (new[k] && k < i))
                rew = 0;
                if (ag) {
                    {   // staged info counters (LDS reads: unaffected by the HBM write drain that has started)
                        const int* ci = l.cnt + (size_t)(g * A + i) * 9;
                        trq = ci[0]; dtg = ci[1]; dleft = ci[2]; greached = ci[3]; nac = ci[4]; noc = ci[5]; sv = ci[6]; sic = ci[7]; conf = ci[8];
                        gmt = l.cntd[(size_t)(g * A + i) * 2]; if (rotinv) pproj = l.cntd[(size_t)(g * A + i) * 2 + 1];
                        dsp0 = l.cntd[(size_t)G * A * 2 + g];
                    }
                    const double px = v.ex[i], py = v.ey[i];
                    const double* row = v.Dm + (size_t)i * E;
                    if (rotfam) write_obs_rot<AP, SC>(p, v, i, ph1); else write_obs<AP, SC>(p, v, i, v.vox[i], v.voy[i], ph1);
                    STAMP(13);
                    // collision block (…_july.py:1117-1124) and info_callback's collision count (:780-786) in one sweep
                    int ncol_r = 0, ncol_i = 0;
                    const bool me_old = v.s_old[i] != 0, me_new = v.newf[i] != 0;
                    if (AP) {                                                  // all LDS reads of the sweep first, then the arithmetic
                        double rv[AP ? AP : 1]; int so[AP ? AP : 1], nf[AP ? AP : 1];
        #pragma unroll
                        for (int a = 0; a < AP; ++a) { const int ac = a < A ? a : 0; rv[a] = row[ac]; so[a] = v.s_old[ac]; nf[a] = v.newf[ac]; }
                        __builtin_amdgcn_sched_barrier(0);
        #pragma unroll
                        for (int a = 0; a < AP; ++a) {
                            const bool close = a < A && rv[a] < c.sep_dist && a != i;
                            ncol_r += (close && !so[a] && !(nf[a] && a < i)) ? 1 : 0;
                            ncol_i += (close && !so[a] && !(nf[a] && a <= i)) ? 1 : 0;
                        }
                    } else {
                        for (int a = 0; a < A; ++a) {
                            const bool close = row[a] < c.sep_dist && a != i;
                            const int so = v.s_old[a], nf = v.newf[a];
                            ncol_r += (close && !so && !(nf && a < i)) ? 1 : 0;
                            ncol_i += (close && !so && !(nf && a <= i)) ? 1 : 0;
                        }
                    }
                    if (me_old) { ncol_r = 0; }
                    if (me_old || me_new) ncol_i = 0;
                    // reward term per contact: 4 x collision_rew (…_july.py:1117-1124, rot_inv.py:1134-1139); three_phase_graph.py:965-970: 1 x;
                    // two_phase_graph.py: none (block commented out)
                    if (three) { for (int q = 0; q < ncol_r; ++q) rew -= c.collision_rew; }
                    else if (!two) for (int q = 0; q < ncol_r; ++q) rew -= c.collision_rew * 4;
                    nac += ncol_i;
                    STAMP(14);
                    const bool obst_hit = obstacle_collision_ego(p, v, i);
                    if (obst_hit) { if (!two && !three) rew -= c.collision_rew * 3; noc += 1; }
                    double serr = 0;
                    if (july) {
                        const double tdx = v.tube[T_EXX] - v.tube[T_ENTX], tdy = v.tube[T_EXY] - v.tube[T_ENTY];
                        const double tlen = sqrt(tdx * tdx + tdy * tdy);
                        if (cp == 2 && cp > prevA + 1) rew -= c.goal_rew * 3;
                        const double ux = tdx / tlen, uy = tdy / tlen;
                        const double qx = px - v.tube[T_ENTX], qy = py - v.tube[T_ENTY];
                        const double proj = qx * ux + qy * uy;
                        if (cp == prevA + 1 && phase_reached == cp - 1) {
                            if (cp == 1) {
                                const double edist = norm2(qx - proj * tdx, qy - proj * tdy);   // un-normalised (:1154)
                                if (0 <= proj && proj < 0.1 * tlen && edist < 0.2 * tlen) rew += c.goal_rew * 3;
                            } else if (cp == 2) rew += c.goal_rew * 3;
                        }
                        if (cp == 0) rew -= norm2(v.tube[T_ENTX] - px, v.tube[T_ENTY] - py);
                        else if (cp == 1) {
                            double hx, hy; sincos(v.s2[i], &hy, &hx);
                            int front = -1, back = -1; double fproj = 0, bproj = 0;
                            for (int k = 0; k < A; ++k) {                       // 1136-1143, first wins ties
                                if (k == i) continue;
                                const double pj = (v.ex[k] - px) * hx + (v.ey[k] - py) * hy;
                                if (pj > 0) { if (front < 0 || pj < fproj) { front = k; fproj = pj; } }
                                else { if (back < 0 || pj > bproj) { back = k; bproj = pj; } }
                            }
                            if (front >= 0) { const double df = row[front] - c.sep_dist; serr += df < 0 ? fabs(df) : 0; }
                            if (back >= 0) { const double df = row[back] - c.sep_dist; serr += df < 0 ? fabs(df) : 0; }
                            if (serr > 0) sv += 1;
                            rew -= serr * c.formation_rew;
                            rew -= norm2(v.tube[T_EXX] - px, v.tube[T_EXY] - py);
                            sic += 1;
                        } else if (cp == 2 && phase_reached == 0) cp = 0;
                        else {
                            if (dgoal < c.goal_thresh) { if (me_new) rew += c.goal_rew * 5; }
                            else rew -= dgoal;
                        }
                        if (phase_reached == 1 && cp == 0) conf += 1;
                        if (cp > phase_reached) phase_reached = cp;
                        if (cp < prevA) rew -= c.collision_rew * 3;
                        if (cp < phase_reached) rew -= c.collision_rew;
                        prev_phase = cp;
                    } else if (rotfam) {
                        // Scenario.reward, rot_inv.py:1122-1338; two_phase_graph.py:955-1142; three_phase_graph.py:957-1160
                        const double Lt = v.tube[T_L], hw = v.tube[T_HALFW];
                        double ts, ty; tube_sy(v.tube, px, py, ts, ty);
                        if (cp == 2 && cp > prev_phase + 1) rew -= c.goal_rew;
                        if (cp == prev_phase + 1 && phase_reached == cp - 1) {
                            if (cp == 1 && in_entrance_gate(ts, ty, Lt, hw) && cooldown == 0) {
                                rew += c.goal_rew;
                                // rot_inv: episode_length/10 as a float into an int32 array (:1200, :228); phase graphs: episode_length
                                cooldown = (two || three) ? c.episode_length : (int)((double)c.episode_length / 10);
                                phase_reached = 1;
                            } else if (cp == 2) {
                                rew += c.goal_rew; phase_reached = 2;
                                if (two && me_new) rew += c.goal_rew * 5;                               // finished at the exit gate (:1040-1044)
                            }
                        }
                        double herr = 0;
                        if (two || three) herr = fabs(heading_error_signed(v.tube, v.s2[i]));           // pre-reward heading
                        if (cp == 0) {
                            const double de = entrance_gate_distance(ts, ty, hw);
                            rew -= de;
                            if ((two || three) && de < c.world_size * 0.1) rew -= herr * c.formation_rew * 0.5;
                        } else if (cp == 1) {
                            double hx, hy; sincos(v.s2[i], &hy, &hx);
                            int front = -1, back = -1; double fproj = 0, bproj = 0;
                            for (int k = 0; k < A; ++k) {
                                if (k == i) continue;
                                const double pj = (v.ex[k] - px) * hx + (v.ey[k] - py) * hy;
                                if (pj > 0) { if (front < 0 || pj < fproj) { front = k; fproj = pj; } }
                                else { if (back < 0 || pj > bproj) { back = k; bproj = pj; } }
                            }
                            if (front >= 0) { const double df = row[front] - c.sep_dist; serr += df < 0 ? fabs(df) : 0; }
                            if (back >= 0) { const double df = row[back] - c.sep_dist; serr += df < 0 ? fabs(df) : 0; }
                            if (serr > 0) sv += 1;
                            rew -= serr * c.formation_rew;
                            rew -= exit_gate_distance(ts, ty, Lt, hw);
                            if (rotinv) {
                                const double tdx = v.tube[T_EXX] - v.tube[T_ENTX], tdy = v.tube[T_EXY] - v.tube[T_ENTY];
                                const double tlen = sqrt(tdx * tdx + tdy * tdy);
                                const double proj = (px - v.tube[T_ENTX]) * (tdx / tlen) + (py - v.tube[T_ENTY]) * (tdy / tlen);
                                const double gain = c.goal_rew / (c.world_size * 0.8 * 10);         // :522
                                const double dproj = proj - pproj;
                                rew += gain * (dproj > -0.05 ? dproj : -0.05);
                                pproj = (double)(float)proj;                                       // float32 array (:374)
                            } else rew -= herr * c.formation_rew * 0.1;
                            sic += 1;
                        } else if (rotinv && cp == 2 && phase_reached == 0) cp = 0;
                        else if (cp == 2 && !two) {
                            if (dgoal < c.goal_thresh) { if (me_new) rew += c.goal_rew * 5; }
                            else rew -= dgoal;
                        }
                        if (phase_reached == 1 && cp == 0) conf += 1;
                        if (cp > phase_reached) phase_reached = cp;
                        if (cp < prev_phase) rew -= c.collision_rew;
                        if (cp < phase_reached) rew -= c.collision_rew;
                        prev_phase = cp;
                        if (in_tube_rect(ts, ty, Lt, hw) && cp != 1 && !(three && in_exit_gate(ts, ty, Lt, hw, 0.02))) rew -= c.collision_rew;
                        if (ts > Lt && phase_reached < 1) rew -= c.goal_rew;
                    } else {
                        if (dgoal < c.goal_thresh) { if (me_new) rew += c.goal_rew * 5; }
                        else rew -= dgoal;
                    }
                    rew = clipd(rew, -4 * c.collision_rew, c.goal_rew * 5);
                    if (!rotfam) rew = clipd(rew, c.min_reward, c.max_reward);                      // rot_inv.py:1338 clips once
                    v.serr[i] = two ? 0.0 : serr; v.rew[i] = rew;           // two_phase_graph.py never appends to delta_spacing

                    STAMP(15);
                    // ---- info counters that depend on own data only (…_july.py:744-773)
                    v.dtg_o[i] = dtg; v.trq_o[i] = trq;
                    int nearest = 0; double dmin = INF;
                    SWEEP(q, L) { const bool ok = !AP || q < L; const double d = ok ? row[A + (ok ? q : 0)] : INF; const bool lt = d < dmin; dmin = lt ? d : dmin; nearest = lt ? q : nearest; }
                    const double thr = c.goal_thresh;
                    const int tnow = (int)((double)cur_step * c.dt);
                    if (dmin < thr && (nearest != greached && greached != -1)) { greached = nearest; dleft = (int)dmin; }
                    if (dmin < thr && trq == -1) { trq = tnow; dtg = (int)p_dist; dleft = (int)dmin; greached = nearest; }
                    if (trq == -1) { dtg = (int)p_dist; dleft = (int)dmin; }
                    if (dmin > thr && trq != -1) { dtg = (int)p_dist; trq = tnow; dleft = (int)dmin; }
                    if (dmin < thr && nearest == greached) { dleft = (int)dmin; greached = nearest; }
                    v.dtg_n[i] = dtg; v.trq_n[i] = trq;
                    v.sv_n[i] = sv; v.sv_o[i] = sv - (serr > 0 ? 1 : 0);
                }
                if (block_sync) __syncthreads();
                else {                                                          // every agent lane lives in wave 0: LDS ops of one wave are
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // executed in order, the fences only pin the compiler's order
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                STAMP(6);

                // ---- 4. info (sequential view: agents j<=i already updated, j>i not yet), outputs, write-back
                if (ag) {
                    double rsum = 0;
                    if (c.collaborative) for (int a = 0; a < A; ++a) rsum += v.rew[a];
                    if (out.reward) out.reward[na] = (float)(c.collaborative ? rsum : rew);
                    if (out.done) out.done[na] = done ? 1 : 0;
                    if (ROLL) {                                                 // GraphReplayBuffer.insert's mask rules (graph_buffer.py:223-251; graph_mpe_runner.py:85-90, 395-405)
                        if (p.masks) p.masks[(size_t)slot * p.st_mask + na] = done ? 0.0f : 1.0f;
                        if (p.active) p.active[(size_t)slot * p.st_mask + na] = (done && !all_done) ? 0.0f : 1.0f;
                    }
                    if (out.info) {
                        // the counters are small integers: their sums and sums of squares are exact in fp64 in any order
                        double sd = 0, st = 0, sdd = 0, stt = 0; int ssv = 0;
                        SWEEP(a, A) {
                            const bool ok = !AP || a < A;
                            const int ac = ok ? a : 0;
                            const bool nw = ac <= i;
                            const int dn = v.dtg_n[ac], d_o = v.dtg_o[ac], tn = v.trq_n[ac], to = v.trq_o[ac], svn = v.sv_n[ac], svo = v.sv_o[ac];
                            const double dd = ok ? (double)(nw ? dn : d_o) : 0.0, tt = ok ? (double)(nw ? tn : to) : 0.0;
                            sd += dd; st += tt; sdd += dd * dd; stt += tt * tt;
                            ssv += ok ? (nw ? svn : svo) : 0;
                        }
                        double dsp = dsp0;
                        if (july || rotfam) for (int a = 0; a <= i; ++a) dsp += v.serr[a];   // same order as the list append (:1180)
                        const double dm = sd / A, tm = st / A;
                        // population variance = (A*sum(x^2) - sum(x)^2) / A^2, numerator exact
                        const double dvn = (double)A * sdd - sd * sd, tvn = (double)A * stt - st * st;
                        const double ds = sqrt(dvn) / A, ts = sqrt(tvn) / A;
                        float* o = out.info + na * GMPE_INFO_KEYS;
                        o[0] = (float)rew; o[1] = (float)dleft; o[2] = (float)trq; o[3] = (float)nac; o[4] = (float)noc;
                        o[5] = (float)dm; o[6] = (float)ds; o[7] = (float)(dm / (ds + 0.0001)); o[8] = (float)dtg;
                        o[9] = (float)trq; o[10] = (float)tm; o[11] = (float)ts; o[12] = (float)(tm / (ts + 0.0001));
                        o[13] = (float)((double)conf / c.episode_length);
                        o[14] = (float)(dsp / (ssv != 0 ? (double)ssv : 1.0));
                        o[15] = (float)((double)sv / (sic != 0 ? sic : 1));
                        o[16] = (float)gmt;
                        o[17] = (float)phase_reached;
                    }
                    if (ROLL && !all_done && kk + 1 < K) {                      // rollout: carry the counters in their LDS staging slots
                        int* ci = l.cnt + (size_t)(g * A + i) * 9;
                        ci[0] = trq; ci[1] = dtg; ci[2] = dleft; ci[3] = greached; ci[4] = nac; ci[5] = noc; ci[6] = sv; ci[7] = sic; ci[8] = conf;
                        if (rotinv) l.cntd[(size_t)(g * A + i) * 2 + 1] = pproj;
                        if (i == 0) {
                            double dsp = dsp0;
                            if (july || rotfam) for (int a = 0; a < A; ++a) dsp += v.serr[a];
                            l.cntd[(size_t)G * A * 2 + g] = dsp;                // read by every lane of the env at the top of section 3: same wave, program order
                        }
                    } else if (!all_done) {                                     // persist the stepped state
                        if (i == 0) {
                            double dsp = dsp0;
                            if (july || rotfam) for (int a = 0; a < A; ++a) dsp += v.serr[a];
                            p.s.delta_spacing[n] = dsp;
                            p.s.rng_ctr[n] = ctr0 + v.flags[1];
                            p.s.current_step[n] = cur_step;
                        }
                        p.s.x[na] = v.ex[i]; p.s.y[na] = v.ey[i]; p.s.s2[na] = v.n2[i]; p.s.s3[na] = v.n3[i];
                        p.s.status[na] = (uint8_t)(v.s_old[i] || v.newf[i]);
                        p.s.prev_phase[na] = prev_phase; p.s.phase_reached[na] = phase_reached; p.s.cooldown[na] = cooldown;
                        p.s.goal_tracker[na] = v.gt[i]; p.s.p_dist[na] = p_dist; p.s.time[na] = tim;
                        p.s.times_required[na] = trq; p.s.dists_to_goal[na] = dtg; p.s.dist_left[na] = dleft;
                        p.s.goal_reached[na] = greached; p.s.n_agent_coll[na] = nac; p.s.n_obst_coll[na] = noc;
                        p.s.spacing_viol[na] = sv; p.s.steps_in_corr[na] = sic; p.s.conformance[na] = conf;
                        if (rotinv) p.s.prev_proj[na] = pproj;
                    }
                }
                STAMP(7);
                if (spec && !ROLL) {       // wave-local: the rows were written by this wave's own lanes
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (out.obs) {
                        float* dst = out.obs + (size_t)n0 * A * D;
                        const int AD = A * D;
                        for (int q = tid; q < Gv * AD; q += 64) { const int gg = fdiv(q, AD, p.m_AD); if (l.flags[gg * 4 + 3]) dst[q] = l.obs[(size_t)gg * AD4 + (q - gg * AD)]; }
                    }
                    if (out.agent_id) for (int q = tid; q < Gv * A; q += 64) { const int gg = fdiv(q, A, p.m_A); if (l.flags[gg * 4 + 3]) out.agent_id[(size_t)n0 * A + q] = q - gg * A; }
                }
            }
            else {
                STAMP_T(9, 64); STAMP_T(17, 192);                                // waves 1 / 3, right before / after the issue of their share of the graph stores
                stream_graph_fn<BLOCK, AP, SC, FL>(p, out, l, Gv, n0, tid, tid - 64, BLOCK - 64, false, any_mask);
                STAMP_T(10, 64); STAMP_T(18, 192);
            }
            __syncthreads();
        }
        {
            // ---- 5. reset (explicit, or the worker's auto-reset when every agent of the env is done)
            if (any_reset) {
                const bool mine = ag && v.flags[0];
                if (tid < 64) {                                                   // wave 0 holds every agent lane: placement is a wave-cooperative loop
                    int64_t ctr = ctr0 + ((mine && step) ? v.flags[1] : 0);      // this step's heading re-draws come first
                    STAMP(20);
                    reset_world_coop<SC, BATCHK>(p, v, n, i, mine, emask, ctr, err);
                    STAMP(21);
                    if (mine && i == 0) {
                        if (ROLL) reinterpret_cast<long long*>(l.cntd + (size_t)G * A * 2 + G)[g] = ctr;
                        p.s.rng_ctr[n] = ctr;
                        p.s.current_step[n] = 0;
                        p.s.delta_spacing[n] = 0.0;
                        v.flags[2] = 0;
                    }
                }
                __syncthreads();
                if (mine) {
                    v.s2[i] = v.n2[i]; v.s3[i] = v.n3[i];
                    double vx, vy; vel_of<SC>(v.n2[i], v.n3[i], vx, vy);
                    v.vox[i] = v.vnx[i] = vx; v.voy[i] = v.vny[i] = vy;
                    v.s_old[i] = 0; v.newf[i] = 0; v.gt[i] = -1; v.moff[i] = 0; v.moff[A + i] = 0;
                    int prevA = prev_phase, ph = 0;
                    if (july) ph = phase_eval(v.tube, v.ex[i], v.ey[i], prev_phase, prevA);   // reset-time observation (:1447)
                    prev_phase = prevA;
                    if (rotfam) { ph = phase_eval_rot<PV>(v.tube, v.ex[i], v.ey[i], prev_phase, 0); double sn_, cn_; sincos(v.n2[i], &sn_, &cn_); v.cn[i] = cn_; v.sn[i] = sn_; p.s.prev_proj[na] = 0.0; }
                    const double dx = v.ex[i] - v.ex[A + i], dy = v.ey[i] - v.ey[A + i];
                    gmt = c.max_speed > 0 ? sqrt(dx * dx + dy * dy) / c.max_speed : 0.0;
                    p.s.x[na] = v.ex[i]; p.s.y[na] = v.ey[i]; p.s.s2[na] = v.n2[i]; p.s.s3[na] = v.n3[i];
                    p.s.status[na] = 0; p.s.prev_phase[na] = prev_phase; p.s.phase_reached[na] = 0; p.s.cooldown[na] = 0;
                    p.s.goal_tracker[na] = -1; p.s.p_dist[na] = 0.0; p.s.time[na] = 0.0;
                    p.s.times_required[na] = -1; p.s.dists_to_goal[na] = -1; p.s.dist_left[na] = -1;
                    p.s.goal_reached[na] = -1; p.s.n_agent_coll[na] = 0; p.s.n_obst_coll[na] = 0;
                    p.s.spacing_viol[na] = 0; p.s.steps_in_corr[na] = 0; p.s.conformance[na] = 0;
                    p.s.goal_min_time[na] = gmt;
                    ph1 = ph;
                }
                STAMP(22);
                __syncthreads();                                                // positions of all agents final
                if constexpr (FUSE_OK) {
                    if (FUSE) distance_force_pass<BLOCK, AP, !kin>(p, l, G, Gv, tid, true, true);   // the reset envs' rows, statics and next-step forces
                    else distance_pass<BLOCK, (CT ? AP : 0)>(p, l, Gv, tid, true);
                } else distance_pass<BLOCK, (CT ? AP : 0)>(p, l, Gv, tid, true);
                __syncthreads();
                STAMP(23);
                if (mine) { if (rotfam) write_obs_rot<AP, SC>(p, v, i, ph1); else write_obs<AP, SC>(p, v, i, v.vox[i], v.voy[i], ph1); }
                any_mask = 0;
                for (int gg = 0; gg < Gv; ++gg) any_mask |= l.flags[gg * 4 + 2];
            }
            STAMP(8);
        }
        if (ROLL && !early) {
            // A reset iteration stores with every wave, a steady-state one with waves 1.. (graph) and wave 0 (obs): when consecutive steps
            // share output addresses (one slot, or the slots wrap) two different waves write the same address in a row. Make the earlier
            // step's stores complete (vmcnt(0): acknowledged by L2) in every wave before any wave issues the later ones.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
        }
        if (!early) stream_graph_fn<BLOCK, AP, SC, FL>(p, out, l, Gv, n0, tid, tid, BLOCK, true, any_mask);
        // ---- small outputs: obs staging rows and agent ids. In a specialised tile wave 0 has already stored them
        // (it wrote the staging rows itself), so nobody waits behind the barrier for the streaming waves.
        STAMP(11);
        // (rollouts: wave 0's per-step work is the critical path of the whole kernel, so there the 3 us of obs stores — queued behind
        // the streaming waves' traffic — are taken off it and every thread stores a few elements after the step's closing barrier)
        if (!spec || ROLL) {
            if (out.obs) {
                float* dst = out.obs + (size_t)n0 * A * D;
                const int AD = A * D;
                for (int q = tid; q < Gv * AD; q += BLOCK) { const int gg = fdiv(q, AD, p.m_AD); if (l.flags[gg * 4 + 3]) dst[q] = l.obs[(size_t)gg * AD4 + (q - gg * AD)]; }
            }
            if (out.agent_id) for (int q = tid; q < Gv * A; q += BLOCK) { const int gg = fdiv(q, A, p.m_A); if (l.flags[gg * 4 + 3]) out.agent_id[(size_t)n0 * A + q] = q - gg * A; }   // get_id :1554
        }
        if (ROLL && kk + 1 < K) {
            // ---- carry the state into the next step (what the next launch would have re-read from HBM)
            if (!early) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (!spec || !early) __syncthreads();                               // the tail above read flags / staging rows with every thread
            if (ag) {
                if (all_done) {                                                 // this env was reset in this step (state already in LDS: section 5)
                    phase_reached = 0; cooldown = 0; p_dist = 0.0; tim = 0.0; cur_step = 0;
                    ctr0 = reinterpret_cast<const long long*>(l.cntd + (size_t)G * A * 2 + G)[g];
                    int* ci = l.cnt + (size_t)(g * A + i) * 9;
                    ci[0] = -1; ci[1] = -1; ci[2] = -1; ci[3] = -1; ci[4] = 0; ci[5] = 0; ci[6] = 0; ci[7] = 0; ci[8] = 0;
                    l.cntd[(size_t)(g * A + i) * 2] = gmt; l.cntd[(size_t)(g * A + i) * 2 + 1] = 0.0;
                    if (i == 0) l.cntd[(size_t)G * A * 2 + g] = 0.0;
                } else {
                    v.s2[i] = v.n2[i]; v.s3[i] = v.n3[i];
                    v.s_old[i] = (v.s_old[i] || v.newf[i]) ? 1 : 0;
                    ctr0 += ndraw;
                }
                const int na_ = c.n_actions;
                act_idx = act_next < 0 ? 0 : (act_next >= na_ ? na_ - 1 : act_next);
            }
            for (int q = tid; q < G * E; q += BLOCK) l.moff[q] = 0;
            if (tid < G) {
                const bool active = n0 + tid < p.env_hi;
                l.flags[tid * 4 + 0] = 0; l.flags[tid * 4 + 1] = 0; l.flags[tid * 4 + 2] = 0; l.flags[tid * 4 + 3] = active;
            }
            slot = slot + 1 == p.num_slots ? 0 : slot + 1;
            __syncthreads();
        }
    } while (ROLL && ++kk < K);
    if (ag && err) atomicOr(&p.s.error_flags[n], err);
    STAMP(12);
}


// Host entry points of one scenario variant; defined and explicitly instantiated in gmpe_sc.hip (-DGMPE_SC=k).
template <int SC> void launch_env(int block, int ap, int fl, dim3 grid, size_t lds, hipStream_t st, const KParams& p);
template <int SC> hipError_t set_max_lds(int lds);
template <int SC> int max_tiles_per_cu(int block, int ap, size_t lds, int roll, int g = 0);   // of the steady-state (FL = 1) instantiation where one exists; roll: of the rollout (FL = 2) one; g: envs per tile (selects the compile-time-G instantiation where one exists)

}  // namespace gmpe
