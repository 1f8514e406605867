// gmpe_step.hip — fused GraphMPE step / reset kernel for gfx950 (MI355X) + the C ABI of include/gmpe.h.
//
// One workgroup per environment. Everything the reference does for one env in one
// `MultiAgentGraphEnv.step` (multiagent/environment.py:1021-1063) + the worker's auto-reset
// (onpolicy/envs/env_wrappers.py:865-870) happens inside ONE launch:
//
//   load SoA state -> LDS | decode action + integrate (closed-form unicycle, or MPE soft-contact
//   forces) | phase FSM + goal reach (parallel restatement of the sequential agent loop, SURVEY.md
//   §8a "ordered-visibility rule") | reward / done / info | optional reset (serial rejection sampler
//   on lane 0) | E×E distance matrix in LDS | coalesced 16-byte stores of adj [A,E,E], node_obs
//   [A,E,8], obs [A,D].
//
// HBM-bound by construction: per env-step the kernel reads ~1 KB of state and writes
// 4·A·(E² + 8E + D + 2) bytes of fp32 observations; no MFMA (there is no contraction here).
// All geometry is fp64 with contraction OFF so thresholds see the reference's roundings; values are
// rounded to fp32 once, on store, like GraphReplayBuffer's float32 copy (graph_buffer.py:226).
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

struct KParams {
    gmpe_config c;
    DevState s;
    gmpe_outputs o;
    const int32_t* act;       // [N,A] or nullptr
    const float* onehot;      // [N,A,n_actions] or nullptr
    const uint8_t* mask;      // reset mask or nullptr
    int mode;
    int A, L, O, E, D, F;     // F = node features per row (8, rot_inv: 7)
    int G;                    // envs per workgroup (G*A <= 64)
    int spec;                 // 1: multi-wave tiles specialise (wave 0 reward/info, waves 1.. graph stores)
    int nt;                   // 1: nontemporal graph stores (outputs per launch exceed the 256 MiB Infinity Cache)
    int ablate;               // debug: timing-only builds of the kernel skip parts (GMPE_ABLATE)
    // magic multipliers for exact unsigned division by run-time constants (q < 2^22): floor(q/d) = umulhi(q, m)
    uint32_t m_E, m_AE, m_EE, m_nq, m_2E, m_pe, m_AD, m_A, m_L, m_O, m_S, m_SS, m_C, m_AC, m_AEE, m_W, m_Sx, m_FW;
    unsigned long long* stamps;   // diagnostic build only (-DGMPE_STAMPS): [grid][16] s_memtime per phase
};
__host__ __device__ inline uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(0x100000000ull / d) + 1u; }
__device__ __forceinline__ int fdiv(int q, int d, uint32_t m) { return d <= 1 ? q : (int)__umulhi((uint32_t)q, m); }

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
    int *s_old, *newf, *gt;           // [G][A]  status before, newly-reached flag, goal_tracker (final)
    int *dtg_o, *dtg_n, *trq_o, *trq_n, *sv_o, *sv_n;   // [G][A] info counters old/new
    int *flags;                       // [G][4]  0: reset this env, 1: heading draws this step, 2: env has masked nodes, 3: env active
    int *moff;                        // [G][E]  adjacency mask per node (done agent / reached landmark)
    int *ptab;                        // [A(A-1)/2] agent pairs (a<<8 | k), a < k, shared by the G envs of the tile
    float *obs;                       // [G][A*D] staging
    float *M;                         // [G][E*E] masked distance matrix, fp32
};
__host__ __device__ inline size_t lds_bytes(int G, int A, int E, int D) {
    const size_t EE4 = ((size_t)E * E + 3) / 4 * 4, AD4 = ((size_t)A * D + 3) / 4 * 4;
    size_t d = (size_t)G * (2 * E + 12 * A + 12 + (size_t)A * E);   // doubles
    size_t f = (size_t)G * (EE4 + AD4);                             // floats
    size_t i = (size_t)G * (9 * A + 4 + E) + (size_t)A * (A - 1) / 2 + 1;   // ints
    return d * 8 + 16 + f * 4 + ((i * 4 + 15) / 16) * 16 + 32;
}
__device__ inline Lds carve(char* base, int G, int A, int E, int D) {
    Lds l;
    double* d = reinterpret_cast<double*>(base);
    l.ex = d; d += G * E; l.ey = d; d += G * E;
    l.s2 = d; d += G * A; l.s3 = d; d += G * A; l.n2 = d; d += G * A; l.n3 = d; d += G * A;
    l.vox = d; d += G * A; l.voy = d; d += G * A; l.vnx = d; d += G * A; l.vny = d; d += G * A;
    l.serr = d; d += G * A; l.cn = d; d += G * A; l.sn = d; d += G * A; l.rew = d; d += G * A; l.tube = d; d += G * 12;
    l.Dm = d; d += (size_t)G * A * E;
    if ((uintptr_t)d & 15) d += 1;
    float* f = reinterpret_cast<float*>(d);
    const size_t EE4 = ((size_t)E * E + 3) / 4 * 4, AD4 = ((size_t)A * D + 3) / 4 * 4;
    l.M = f; f += (size_t)G * EE4;
    l.obs = f; f += (size_t)G * AD4;
    int* i = reinterpret_cast<int*>(f);
    l.s_old = i; i += G * A; l.newf = i; i += G * A; l.gt = i; i += G * A;
    l.dtg_o = i; i += G * A; l.dtg_n = i; i += G * A; l.trq_o = i; i += G * A; l.trq_n = i; i += G * A;
    l.sv_o = i; i += G * A; l.sv_n = i; i += G * A; l.flags = i; i += G * 4; l.moff = i; i += G * E; l.ptab = i;
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
    v.Dm = l.Dm + (size_t)g * A * E;
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

__device__ __forceinline__ bool kinematic(const gmpe_config& c) { return c.dynamics != GMPE_DYN_DOUBLE_INTEGRATOR; }

__device__ __forceinline__ void vel_of(const gmpe_config& c, double a2, double a3, double& vx, double& vy) {
    if (kinematic(c)) { double sn, cs; sincos(a2, &sn, &cs); vx = a3 * cs; vy = a3 * sn; }      // core.py:281-286
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
// Scenario.is_obstacle_collision (…_july.py:864-890) at an arbitrary point (reset placement)
__device__ __forceinline__ bool obstacle_collision(const KParams& p, const Lds& l, double px, double py, double size) {
    const int o0 = p.A + p.L;
    for (int o = 0; o < p.O; ++o)
        if (norm2(l.ex[o0 + o] - px, l.ey[o0 + o] - py) < 2.0 * (p.c.entity_size + size)) return true;
    return wall_band_hit(p, px, py, size);
}
// same test for agent i at its current position, distances taken from the shared fp64 rows
__device__ __forceinline__ bool obstacle_collision_ego(const KParams& p, const Lds& l, int i) {
    const double* row = l.Dm + (size_t)i * p.E + p.A + p.L;
    for (int o = 0; o < p.O; ++o)
        if (row[o] < 2.0 * (p.c.entity_size + p.c.entity_size)) return true;
    return wall_band_hit(p, l.ex[i], l.ey[i], p.c.entity_size);
}

// _set_action (environment.py:336-475)
__device__ __forceinline__ void decode_action(const gmpe_config& c, int idx, double& u0, double& u1) {
    if (c.dynamics == GMPE_DYN_DOUBLE_INTEGRATOR) {
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
template <int AP>
__device__ __forceinline__ void write_obs(const KParams& p, const Lds& l, int i, double vx, double vy, int phase) {
    float* o = l.obs + (size_t)i * p.D;
    const double px = l.ex[i], py = l.ey[i];
    const double gx = l.ex[p.A + i] - px, gy = l.ey[p.A + i] - py;
    o[0] = (float)px; o[1] = (float)py; o[2] = (float)vx; o[3] = (float)vy;
    o[4] = (float)gx; o[5] = (float)gy; o[6] = 0.0f; o[7] = (float)gx; o[8] = (float)gy;
    // stable two-smallest of the other agents (1398-1417): strict '<' keeps the first of equal distances
    const double INF = __builtin_huge_val();
    int b1 = -1, b2 = -1; double d1 = INF, d2 = INF;
    const double* row = l.Dm + (size_t)i * p.E;
    if (AP) {
        double rv[AP ? AP : 1];
#pragma unroll
        for (int k = 0; k < AP; ++k) rv[k] = row[k < p.A ? k : 0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < AP; ++k) {
            const double d = (k < p.A && k != i) ? rv[k] : INF;
            const bool lt1 = d < d1, lt2 = d < d2;
            d2 = lt1 ? d1 : (lt2 ? d : d2); b2 = lt1 ? b1 : (lt2 ? k : b2);
            d1 = lt1 ? d : d1; b1 = lt1 ? k : b1;
        }
    } else {
        for (int k = 0; k < p.A; ++k) {
            const double d = k != i ? row[k] : INF;
            const bool lt1 = d < d1, lt2 = d < d2;
            d2 = lt1 ? d1 : (lt2 ? d : d2); b2 = lt1 ? b1 : (lt2 ? k : b2);
            d1 = lt1 ? d : d1; b1 = lt1 ? k : b1;
        }
    }
    o[9] = b1 >= 0 ? (float)(l.ex[b1] - px) : 0.f; o[10] = b1 >= 0 ? (float)(l.ey[b1] - py) : 0.f;
    o[11] = b2 >= 0 ? (float)(l.ex[b2] - px) : 0.f; o[12] = b2 >= 0 ? (float)(l.ey[b2] - py) : 0.f;
    if (p.c.scenario == GMPE_SCENARIO_TUBE_JULY) {
        o[13] = (float)(l.tube[T_ENTX] - px); o[14] = (float)(l.tube[T_ENTY] - py);
        o[15] = (float)(l.tube[T_EXX] - px); o[16] = (float)(l.tube[T_EXY] - py);
        o[17] = (float)l.tube[T_WIDTH]; o[18] = (float)phase;
    }
}

// Scenario.observation of rot_inv (…rot_inv.py:1453-1548): 13 float32 = [cos th, sin th, speed, goal (rotated), two nearest
// neighbours (rel. vector cast to float32, then rotated in float64), s/L, y/half_w (clipped), exit-gate distance / L, phase].
// Uses the PRE-reward heading (the reward of this agent runs after its observation).
template <int AP>
__device__ __forceinline__ void write_obs_rot(const KParams& p, const Lds& l, int i, int phase) {
    float* o = l.obs + (size_t)i * p.D;
    const double px = l.ex[i], py = l.ey[i], th = l.s2[i];
    double sn, cs; sincos(th, &sn, &cs);
    const double INF = __builtin_huge_val();
    int b1 = -1, b2 = -1; double d1 = INF, d2 = INF;
    const double* row = l.Dm + (size_t)i * p.E;
    for (int k = 0; k < p.A; ++k) {
        const double d = k != i ? row[k] : INF;
        const bool lt1 = d < d1, lt2 = d < d2;
        d2 = lt1 ? d1 : (lt2 ? d : d2); b2 = lt1 ? b1 : (lt2 ? k : b2);
        d1 = lt1 ? d : d1; b1 = lt1 ? k : b1;
    }
    double gx, gy; rot2(cs, sn, l.ex[p.A + i] - px, l.ey[p.A + i] - py, gx, gy);
    double n1x = 0, n1y = 0, n2x = 0, n2y = 0;
    if (b1 >= 0) rot2(cs, sn, (double)(float)(l.ex[b1] - px), (double)(float)(l.ey[b1] - py), n1x, n1y);
    if (b2 >= 0) rot2(cs, sn, (double)(float)(l.ex[b2] - px), (double)(float)(l.ey[b2] - py), n2x, n2y);
    const double L = l.tube[T_L], hw = l.tube[T_HALFW];
    double s, yy; tube_sy(l.tube, px, py, s, yy);
    o[0] = (float)cs; o[1] = (float)sn; o[2] = (float)l.s3[i];
    o[3] = (float)gx; o[4] = (float)gy; o[5] = (float)n1x; o[6] = (float)n1y; o[7] = (float)n2x; o[8] = (float)n2y;
    o[9] = (float)clipd(s / L, -2.0, 2.0); o[10] = (float)clipd(yy / (hw + 1e-9), -2.0, 2.0);
    o[11] = (float)(exit_gate_distance(s, yy, L, hw) / (L + 1e-9)); o[12] = (float)phase;
}

// Serial reset of one env by ONE lane (reset_world: …_july.py:339-420, 440-515, 518-613,
// custom_scenarios/utils.py:165-193; navigation_graph: DESIGN.md). Writes positions / headings to
// LDS (ex, ey, n2, n3, tube) and landmark / obstacle / tube records to HBM. Bounded rejection loop.
__device__ __forceinline__ void reset_world_serial(const KParams& p, const Lds& l, int n, int64_t& ctr, int& err) {
    const gmpe_config& c = p.c;
    const double ws = c.world_size, size = c.entity_size;
    const int A = p.A, L = p.L, O = p.O;
    if (c.scenario != GMPE_SCENARIO_NAVIGATION_GRAPH) {
        (void)draw_at(c, p.s, n, ctr++, err);                              // wall_length draw, unused (:368)
        const double a = 3 * size * 2.5, b = ws * 0.15;
        const double width = a > b ? a : b;
        const double angle = -M_PI / 2 + (M_PI / 2 - (-M_PI / 2)) * draw_at(c, p.s, n, ctr++, err);
        const double tl = ws * 0.8;
        const double ca = cos(angle), sa = sin(angle);
        const double be = tl / 4, bx = -tl / 4;
        const double entx = ca * 0 + sa * be, enty = -sa * 0 + ca * be;
        const double exx = ca * 0 + sa * bx, exy = -sa * 0 + ca * bx;
        const double dx = exx - entx, dy = exy - enty;
        const double Lt = sqrt(dx * dx + dy * dy) + 1e-9;
        const double ex = dx / Lt, ey = dy / Lt;
        double* t = l.tube;
        t[T_ANGLE] = angle; t[T_ENTX] = entx; t[T_ENTY] = enty; t[T_EXX] = exx; t[T_EXY] = exy;
        t[T_EX] = ex; t[T_EY] = ey; t[T_NX] = (double)(float)(-ey); t[T_NY] = (double)(float)ex;
        t[T_L] = Lt; t[T_HALFW] = width * 0.5; t[T_WIDTH] = width;
        for (int q = 0; q < GMPE_TUBE_STRIDE; ++q) p.s.tube[(size_t)n * GMPE_TUBE_STRIDE + q] = t[q];
        int k = 0, tries = 0;
        while (k < A) {
            const double u0 = draw_at(c, p.s, n, ctr++, err), u1 = draw_at(c, p.s, n, ctr++, err);
            const bool rot = c.scenario == GMPE_SCENARIO_ROT_INV;              // rot_inv.py:463, 469
            const double jf = rot ? 0.3 : 0.2;
            const double jx = jf * (-ws + (ws - (-ws)) * u0), jy = jf * (-ws + (ws - (-ws)) * u1);
            const double dfe = rot ? (ws + k) / 3 : (ws + k) / 5;
            const double px = entx + dfe * sa + jx, py = enty + dfe * ca + jy;
            bool bad = obstacle_collision(p, l, px, py, size);
            for (int q = 0; q < k && !bad; ++q) bad = norm2(l.ex[q] - px, l.ey[q] - py) < c.sep_dist;
            if (bad && ++tries < GMPE_MAX_TRIES) continue;
            if (bad) err |= 2;
            l.ex[k] = px; l.ey[k] = py;
            l.n2[k] = 0.0 + (2 * M_PI - 0.0) * draw_at(c, p.s, n, ctr++, err);
            l.n3[k] = c.v_min;
            ++k; tries = 0;
        }
        const double rel = -ws / 3;
        const double rx = ca * 0.0 + sa * rel, ry = -sa * 0.0 + ca * rel;
        for (int q = 0; q < L; ++q) { l.ex[A + q] = exx + rx; l.ey[A + q] = exy + ry; }
    } else {
        const double lo = -ws / 2, hi = ws / 2;
        for (int o = 0, tries = 0; o < O;) {
            const double px = 0.8 * (lo + (hi - lo) * draw_at(c, p.s, n, ctr, err));
            const double py = 0.8 * (lo + (hi - lo) * draw_at(c, p.s, n, ctr + 1, err));
            ctr += 2;
            bool bad = false;
            for (int q = 0; q < o && !bad; ++q) bad = norm2(l.ex[A + L + q] - px, l.ey[A + L + q] - py) < 2.0 * (size + size);
            if (bad && ++tries < GMPE_MAX_TRIES) continue;
            if (bad) err |= 2;
            l.ex[A + L + o] = px; l.ey[A + L + o] = py; ++o; tries = 0;
        }
        for (int k = 0, tries = 0; k < A;) {
            const double px = 0.8 * (lo + (hi - lo) * draw_at(c, p.s, n, ctr, err));
            const double py = 0.8 * (lo + (hi - lo) * draw_at(c, p.s, n, ctr + 1, err));
            ctr += 2;
            bool bad = obstacle_collision(p, l, px, py, size);
            for (int q = 0; q < k && !bad; ++q) bad = norm2(l.ex[q] - px, l.ey[q] - py) < c.sep_dist;
            if (bad && ++tries < GMPE_MAX_TRIES) continue;
            if (bad) err |= 2;
            l.ex[k] = px; l.ey[k] = py; l.n2[k] = 0.0; l.n3[k] = 0.0; ++k; tries = 0;
        }
        for (int q = 0, tries = 0; q < L;) {
            const double px = 0.8 * (lo + (hi - lo) * draw_at(c, p.s, n, ctr, err));
            const double py = 0.8 * (lo + (hi - lo) * draw_at(c, p.s, n, ctr + 1, err));
            ctr += 2;
            bool bad = obstacle_collision(p, l, px, py, size);
            for (int r = 0; r < q && !bad; ++r) bad = norm2(l.ex[A + r] - px, l.ey[A + r] - py) < c.sep_dist;
            if (bad && ++tries < GMPE_MAX_TRIES) continue;
            if (bad) err |= 2;
            l.ex[A + q] = px; l.ey[A + q] = py; ++q; tries = 0;
        }
        for (int o = 0; o < O; ++o) {
            p.s.obstacles[((size_t)n * O + o) * 2] = l.ex[A + L + o];
            p.s.obstacles[((size_t)n * O + o) * 2 + 1] = l.ey[A + L + o];
        }
    }
    for (int q = 0; q < L; ++q) {
        p.s.landmarks[((size_t)n * L + q) * 2] = l.ex[A + q];
        p.s.landmarks[((size_t)n * L + q) * 2 + 1] = l.ey[A + q];
    }
}


// Post-move distance pass for every env of the tile (World.calculate_distances, core.py:600-624:
// delta taken as pos[min]-pos[max], so the matrix is exactly symmetric). Writes the fp64 agent rows
// Dm[g][r][c] AND the unmasked fp32 matrix M (agent rows + their mirrored columns); the static
// (landmark/obstacle) x (landmark/obstacle) block is filled by static_block().
template <int BLOCK>
__device__ __forceinline__ void distance_pass(const KParams& p, const Lds& l, int G, int tid, bool only_reset) {
    const int A = p.A, E = p.E, AE = A * E, EE4 = (E * E + 3) / 4 * 4;
    const int NP = A * (A - 1) / 2, S = E - A, AS = A * S, W = NP + AS;   // per env: agent pairs + agent x static entities
    const int total = G * W;
    for (int q0 = tid; q0 < total; q0 += 2 * BLOCK) {                   // two independent entries per trip
        double ds[2]; int gs[2], rs[2], cs[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = q0 + u * BLOCK;
            const bool live = q < total;
            const int qq = live ? q : tid;
            const int g = fdiv(qq, W, p.m_W), w = qq - g * W;
            const bool ap_ = w < NP;
            const int pk = l.ptab[ap_ ? w : 0];
            const int t = ap_ ? 0 : w - NP;
            const int ro = fdiv(t, S, p.m_Sx);
            const int r = ap_ ? (pk >> 8) : ro, cc = ap_ ? (pk & 255) : A + (t - ro * S);   // r < cc always
            const double dx = l.ex[g * E + r] - l.ex[g * E + cc], dy = l.ey[g * E + r] - l.ey[g * E + cc];
            ds[u] = sqrt(dx * dx + dy * dy);
            gs[u] = (live && !(only_reset && !l.flags[g * 4 + 0])) ? g : -1; rs[u] = r; cs[u] = cc;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (gs[u] < 0) continue;
            const int r = rs[u], cc = cs[u];
            double* Dg = l.Dm + (size_t)gs[u] * AE;
            float* Mg = l.M + (size_t)gs[u] * EE4;
            const float df = (float)ds[u];
            Dg[r * E + cc] = ds[u]; Mg[r * E + cc] = df; Mg[cc * E + r] = df;
            if (cc < A) Dg[cc * E + r] = ds[u];
        }
    }
    for (int q = tid; q < G * A; q += BLOCK) {                           // diagonal
        const int g = fdiv(q, A, p.m_A), r = q - g * A;
        if (only_reset && !l.flags[g * 4 + 0]) continue;
        l.Dm[(size_t)g * AE + r * E + r] = 0.0; l.M[(size_t)g * EE4 + r * E + r] = 0.0f;
    }
}
template <int BLOCK>
__device__ __forceinline__ void static_block(const KParams& p, const Lds& l, int G, int tid, bool only_reset) {
    const int A = p.A, E = p.E, S = p.L + p.O, SS = S * S, EE4 = (E * E + 3) / 4 * 4;
    for (int q = tid; q < G * SS; q += BLOCK) {
        const int g = fdiv(q, SS, p.m_SS), rc = q - g * SS, r3 = fdiv(rc, S, p.m_S), c3 = rc - r3 * S;
        if (only_reset && !l.flags[g * 4 + 0]) continue;
        const int r = A + r3, cc = A + c3;
        const int a = r < cc ? r : cc, b = r < cc ? cc : r;
        const double dx = l.ex[g * E + a] - l.ex[g * E + b], dy = l.ey[g * E + a] - l.ey[g * E + b];
        l.M[(size_t)g * EE4 + r * E + cc] = r != cc ? (float)sqrt(dx * dx + dy * dy) : 0.0f;
    }
}

#define SWEEP(var, n) _Pragma("unroll") for (int var = 0; var < (AP ? AP : (n)); ++var)
// Graph outputs of a tile: optional adjacency mask pass (block-wide, with its barrier), then the adj and node_obs
// stores executed by threads t0, t0+nthr, ... (all BLOCK threads, or only the streaming waves of a specialised tile).
template <int BLOCK, int AP>
__device__ __forceinline__ void stream_graph_fn(const KParams& p, const Lds& l, const int Gv, const int n0, const int tid,
                                                const int t0, const int nthr, const bool do_mask, const int any_mask) {
    const int A = p.A, L = p.L, E = p.E, EE = E * E, EE4 = (EE + 3) / 4 * 4;
    const int abl = p.ablate;
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
    if (p.o.adj && !(abl & 1)) {
        const bool vec = (EE & 3) == 0;
        if (p.o.adj_compact) {
            float* dst = p.o.adj + (size_t)n0 * EE;
            if (vec) {
                const int nq = EE / 4;
                for (int q = t0; q < Gv * nq; q += nthr) {
                    const int gg = fdiv(q, nq, p.m_nq), m = q - gg * nq;
                    if (l.flags[gg * 4 + 3]) reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(l.M + (size_t)gg * EE4)[m];
                }
            } else for (int q = t0; q < Gv * EE; q += nthr) { const int gg = fdiv(q, EE, p.m_EE); if (l.flags[gg * 4 + 3]) dst[q] = l.M[(size_t)gg * EE4 + (q - gg * EE)]; }
        } else {
            float* dst = p.o.adj + (size_t)n0 * A * EE;
            if (vec) {
                const int nq = EE / 4;
                // each lane keeps one float4 of the env's matrix and stores it to the A ego copies
                for (int q = t0; q < Gv * nq; q += nthr) {
                    const int gg = fdiv(q, nq, p.m_nq), m = q - gg * nq;
                    if (!l.flags[gg * 4 + 3]) continue;
                    const float4 val = reinterpret_cast<const float4*>(l.M + (size_t)gg * EE4)[m];
                    float4* d4 = reinterpret_cast<float4*>(dst) + (size_t)gg * A * nq + m;
                    SWEEP(a, A) { if (!AP || a < A) { if (p.nt) nt_store4(&d4[(size_t)a * nq], val); else d4[(size_t)a * nq] = val; } }
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
    
    if (p.o.node_obs && !(abl & 2) && p.F == 7) {
        // rot_inv node row (…rot_inv.py:1690-1766): 7 float32 = [rel_vel, rel_pos, rel_goal (all rotated by the ego heading), type].
        // Positions / velocities are rounded to float32 FIRST, differenced in float32, rotated in float64, rounded again.
        // A lane owns one (env, entity) and walks the egos; rows are 28 B, so the stores are scalar.
        float* base = p.o.node_obs + (size_t)n0 * A * E * 7;
        for (int sidx = t0; sidx < Gv * E; sidx += nthr) {
            const int gg = fdiv(sidx, E, p.m_E), k = sidx - gg * E;
            if (!l.flags[gg * 4 + 3]) continue;
            const int ab = gg * A, eb = gg * E;
            const bool kag = k < A;
            const int kk = kag ? k : 0;
            const float kx = (float)l.ex[eb + k], ky = (float)l.ey[eb + k];
            const float kvox = kag ? (float)l.vox[ab + kk] : 0.0f, kvoy = kag ? (float)l.voy[ab + kk] : 0.0f;
            const float kvnx = kag ? (float)l.vnx[ab + kk] : 0.0f, kvny = kag ? (float)l.vny[ab + kk] : 0.0f;
            const bool knew = kag && l.newf[ab + kk] != 0;
            const float gxk = kag ? (float)l.ex[eb + A + kk] : 0.0f, gyk = kag ? (float)l.ey[eb + A + kk] : 0.0f;
            const float typ = kag ? 0.0f : (k < A + L ? 1.0f : 2.0f);
            for (int ei = 0; ei < A; ++ei) {
                const float apx = (float)l.ex[eb + ei], apy = (float)l.ey[eb + ei];
                const bool en = l.newf[ab + ei] != 0;
                const float avx = (float)(en ? l.vnx[ab + ei] : l.vox[ab + ei]), avy = (float)(en ? l.vny[ab + ei] : l.voy[ab + ei]);
                const double cs = l.cn[ab + ei], sn = l.sn[ab + ei];          // ego heading AFTER its own reward
                const bool post = knew && k <= ei;
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
    if (p.o.node_obs && !(abl & 2) && p.F != 7) {
        // node row (ego, entity k) = 2 float4: [rel_vel, rel_pos] and [rel_goal, occupied, type].
        // A lane owns one (env, entity, half) slot, keeps that entity's data in registers and walks the egos.
        float4* base = reinterpret_cast<float4*>(p.o.node_obs + (size_t)n0 * A * E * GMPE_NODE_FEATS);
        const int E2 = 2 * E;
        for (int sidx = t0; sidx < Gv * E2; sidx += nthr) {
            const int gg = fdiv(sidx, E2, p.m_2E), rem = sidx - gg * E2;
            if (!l.flags[gg * 4 + 3]) continue;
            const int k = rem >> 1, half = rem & 1;
            const int ab = gg * A, eb = gg * E;
            const double kx = l.ex[eb + k], ky = l.ey[eb + k];
            const bool kag = k < A;
            const int kk = kag ? k : 0;
            const double kvox = kag ? l.vox[ab + kk] : 0.0, kvoy = kag ? l.voy[ab + kk] : 0.0;
            const double kvnx = kag ? l.vnx[ab + kk] : 0.0, kvny = kag ? l.vny[ab + kk] : 0.0;
            const bool knew = kag && l.newf[ab + kk] != 0;
            const double gx = kag ? l.ex[eb + A + kk] : kx, gy = kag ? l.ey[eb + A + kk] : ky;
            const float occ = kag ? 0.0f : 1.0f, typ = kag ? 0.0f : (k < A + L ? 1.0f : 2.0f);
            float4* dst = base + (size_t)gg * A * E2 + rem;
            SWEEP(ei, A) {
                const bool ok = !AP || ei < A;
                const int ec = ok ? ei : 0;
                const double px = l.ex[eb + ec], py = l.ey[eb + ec];
                const bool en = l.newf[ab + ec] != 0;
                const double evox = l.vox[ab + ec], evoy = l.voy[ab + ec], evnx = l.vnx[ab + ec], evny = l.vny[ab + ec];
                float4 val;
                if (half == 0) {
                    const double evx = en ? evnx : evox, evy = en ? evny : evoy;
                    const bool post = knew && k <= ec;
                    val = make_float4((float)((post ? kvnx : kvox) - evx), (float)((post ? kvny : kvoy) - evy), (float)(kx - px), (float)(ky - py));
                } else {
                    val = make_float4((float)(gx - px), (float)(gy - py), occ, typ);
                }
                if (ok) { if (p.nt) nt_store4(&dst[(size_t)ec * E2], val); else dst[(size_t)ec * E2] = val; }
            }
        }
    }
}

#ifndef GMPE_MIN_WAVES
#define GMPE_MIN_WAVES 1
#endif
#ifndef GMPE_MIN_WAVES_NOWALLS
#define GMPE_MIN_WAVES_NOWALLS 1
#endif
#ifdef GMPE_STAMPS
#define STAMP(k) do { if (tid == 0 && p.stamps) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------- the fused kernel
// AP > 0: compile-time bound (A, L <= AP) for the per-agent sweeps, so they unroll fully and every LDS read of a
// sweep is issued before the first use (the sweeps are latency-bound: one wave per SIMD, ~100-cycle LDS reads).
// Out-of-range iterations read a clamped index and are masked in the arithmetic. AP == 0: run-time bounds.
// WALLS = false compiles the wall-contact code (asin / cos / softplus inside the agent lane's dynamics) out: it is
// the single largest consumer of registers (187 -> 135 VGPRs), i.e. 2 -> 3 waves per SIMD for wall-less worlds.
template <int BLOCK, int AP, bool WALLS>
__global__ __launch_bounds__(BLOCK, (WALLS ? GMPE_MIN_WAVES : GMPE_MIN_WAVES_NOWALLS)) void k_env(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int A = p.A, L = p.L, O = p.O, E = p.E, D = p.D, G = p.G, N = p.c.num_envs;
    const gmpe_config& c = p.c;
    const Lds l = carve(smem, G, A, E, D);
    const int n0 = blockIdx.x * G;
    const int Gv = min(G, N - n0);                                      // envs actually present in this tile
    const bool july = c.scenario == GMPE_SCENARIO_TUBE_JULY;
    const bool rotinv = c.scenario == GMPE_SCENARIO_ROT_INV;
    const bool step = p.mode == MODE_STEP;
    const bool kin = kinematic(c);
    const int EE = E * E, EE4 = (EE + 3) / 4 * 4, AD4 = (A * D + 3) / 4 * 4;
    const double INF = __builtin_huge_val();

    // agent lane mapping: lane tid of wave 0 = (env g, agent i)
    const int g = fdiv(tid, A, p.m_A), i = tid - g * A;
    const int n = n0 + g;
    bool ag = tid < Gv * A;
    if (ag && !step && p.mask && !p.mask[n]) ag = false;               // explicit reset: masked-out env
    const size_t na = (size_t)n * A + i;
    const Lds v = env_view(l, ag ? g : 0, A, E, D);
    const unsigned long long emask = ag ? ((A >= 64 ? ~0ull : ((1ull << A) - 1ull)) << (g * A)) : 0ull;

    // ---- per-agent registers
    int prev_phase = 0, phase_reached = 0, cooldown = 0;
    double p_dist = 0, tim = 0;
    int trq = -1, dtg = -1, dleft = -1, greached = -1, nac = 0, noc = 0, sv = 0, sic = 0, conf = 0;
    double gmt = 0, dsp0 = 0, pproj = 0;
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
            trq = p.s.times_required[na]; dtg = p.s.dists_to_goal[na]; dleft = p.s.dist_left[na];
            greached = p.s.goal_reached[na]; nac = p.s.n_agent_coll[na]; noc = p.s.n_obst_coll[na];
            sv = p.s.spacing_viol[na]; sic = p.s.steps_in_corr[na]; conf = p.s.conformance[na];
            gmt = p.s.goal_min_time[na];
            if (rotinv) pproj = p.s.prev_proj[na];
            dsp0 = p.s.delta_spacing[n];
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
        const bool active = nn < N && (step || !p.mask || p.mask[nn]);
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
    if (ag && step) {
        v.ex[i] = x0; v.ey[i] = y0; v.s2[i] = a20; v.s3[i] = a30; v.s_old[i] = st0; v.gt[i] = gt0;
        act_idx = act_idx < 0 ? 0 : (act_idx >= c.n_actions ? c.n_actions - 1 : act_idx);
    }
    __syncthreads();
    STAMP(1);

    int ph1 = 0, cp = 0, prevA = 0;
    bool goal_branch = true, done = false, all_done = false;
    double dgoal = 0, rew = 0;
    if (step) {
        cur_step += 1;
        const int C = A + O;                                            // colliders: agents + obstacles (landmarks collide=False)
        double* Fx = reinterpret_cast<double*>(l.M);                    // pair forces alias the (not yet built) fp32 matrix
        double* Fy = Fx + (size_t)G * A * C;
        if (!kin) {
            // ---- 1a. contact forces, one PAIR per lane (get_entity_collision_force core.py:872-906): pair (a, k>a)
            // computed once with delta = pos[a]-pos[k]; side a gets +F, side k gets -F (summed in 1b).
            const int NP = A * (A - 1) / 2, W = NP + A * O;            // valid pairs only: agent pairs (a<k) + agent x obstacle
            for (int q = tid; q < Gv * W; q += BLOCK) {
                const int gg = fdiv(q, W, p.m_FW), w = q - gg * W;
                int a, kk;
                if (w < NP) { const int pk = l.ptab[w]; a = pk >> 8; kk = pk & 255; }
                else { const int t = w - NP; a = fdiv(t, O, p.m_O); kk = A + (t - a * O); }
                const int k = kk < A ? kk : L + kk;                     // entity index of collider kk
                const double dx = l.ex[gg * E + a] - l.ex[gg * E + k], dy = l.ey[gg * E + a] - l.ey[gg * E + k];
                const double dist = sqrt(dx * dx + dy * dy);
                double fx = 0.0, fy = 0.0;
                if (dist < c.sep_dist + 50.0 * c.contact_margin) {      // else softplus < 1e-21: below one ulp of the sum
                    const double pen = logaddexp0(-(dist - c.sep_dist) / c.contact_margin) * c.contact_margin;
                    fx = c.contact_force * dx / dist * pen; fy = c.contact_force * dy / dist * pen;
                }
                Fx[(size_t)gg * A * C + a * C + kk] = fx; Fy[(size_t)gg * A * C + a * C + kk] = fy;
            }
            __syncthreads();
        }
        STAMP(2);
        // ---- 1b. action decode + dynamics
        double nx = 0, ny = 0, nv2 = 0, nv3 = 0;
        if (ag) {
            double u0, u1; decode_action(c, act_idx, u0, u1);
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
                double sx = 1.0 * u0, sy = 1.0 * u1;
                const double* fxg = Fx + (size_t)g * A * C; const double* fyg = Fy + (size_t)g * A * C;
                const bool ego_live = v.s_old[i] == 0;                  // done side gets no agent-agent force (899-900)
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
                if (WALLS) for (int w = 0; w < c.num_walls; ++w) {
                    double wx, wy;
                    if (wall_force(c.walls[w], nx, ny, c.entity_size, c.wall_contact_force, c.wall_contact_margin, wx, wy)) { sx = sx + wx; sy = sy + wy; }
                }
                double vx = nv2 * (1 - c.damping), vy = nv3 * (1 - c.damping);
                vx += (sx / 1.0) * c.dt; vy += (sy / 1.0) * c.dt;
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
        distance_pass<BLOCK>(p, l, Gv, tid, false);                    // post-move rows: obs, reward, info, adj all read these
        static_block<BLOCK>(p, l, Gv, tid, false);
        __syncthreads();
        STAMP(4);

        // ---- 2. phase FSM + who newly reaches the goal (depends only on own data: SURVEY §8a)
        prevA = prev_phase;
        if (ag) {
            const double px = v.ex[i], py = v.ey[i];
            double vx, vy; vel_of(c, v.s2[i], v.s3[i], vx, vy);
            v.vox[i] = vx; v.voy[i] = vy;
            if (july) {
                ph1 = phase_eval(v.tube, px, py, prev_phase, prevA);     // observation's call (:1447)
                if (cooldown > 0) cooldown -= 1;
                int prevB;
                cp = phase_eval(v.tube, px, py, prevA, prevB);           // reward's call (:1113)
                if (cooldown > 0) cooldown -= 1;
                prevA = prevB;
                goal_branch = (cp == 2 && phase_reached != 0);
            } else if (rotinv) {
                // rot_inv.py:675-739: the query mutates only the cooldown, so observation's and reward's calls agree;
                // phase 2 is only returned with phase_reached >= 1, hence the goal block runs iff cp == 2 (:1281-1297)
                ph1 = phase_eval_rot(v.tube, px, py, prev_phase, phase_reached);
                if (cooldown > 0) cooldown -= 1;
                if (cooldown > 0) cooldown -= 1;
                cp = ph1;
                goal_branch = (cp == 2);
            }
            dgoal = v.Dm[(size_t)i * E + A + i];
            v.newf[i] = goal_branch && dgoal < c.goal_thresh && !v.s_old[i];
        }
        // rank of each newly-reached agent among its env's: heading re-draws follow agent order (core.py:328)
        if (tid < 64) {
            const unsigned long long bal = __ballot(ag && v.newf[i]);
            if (ag) {
                if (v.newf[i]) {
                    const int rank = __popcll(bal & emask & ((1ull << tid) - 1ull));
                    if (kin) { v.n2[i] = 0.0 + (2 * M_PI - 0.0) * draw_at(c, p.s, n, ctr0 + rank, err); v.n3[i] = c.v_min; }
                    else { v.n2[i] = 0.0; v.n3[i] = 0.0; }
                    v.gt[i] = i;
                    double vx, vy; vel_of(c, v.n2[i], v.n3[i], vx, vy);
                    v.vnx[i] = vx; v.vny[i] = vy;
                } else { v.n2[i] = v.s2[i]; v.n3[i] = v.s3[i]; v.vnx[i] = v.vox[i]; v.vny[i] = v.voy[i]; }
                if (rotinv) { double sn_, cn_; sincos(v.n2[i], &sn_, &cn_); v.cn[i] = cn_; v.sn[i] = sn_; }
                if (i == 0) v.flags[1] = kin ? __popcll(bal & emask) : 0;   // draws consumed (DI reset_velocity draws none)
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
        }
        __syncthreads();
        STAMP(5);
    }


    int any_reset = 0, any_mask = 0;
    for (int gg = 0; gg < Gv; ++gg) { any_reset |= l.flags[gg * 4 + 0]; any_mask |= l.flags[gg * 4 + 2]; }   // block-uniform
    const int abl = p.ablate;


    // Common case (no env of the tile resets): issue the 22 KB/env of graph stores FIRST, then do the reward /
    // info arithmetic under the HBM write drain. Tiles with a reset need the terminal reward/info before the
    // reset overwrites the LDS state, so they keep the reference's order.
    const bool early = step && !any_reset;
    // Multi-wave tiles specialise: wave 0 (all agent lanes) does reward / info / write-back while waves 1.. stream the
    // graph observations, so the ~7 us of per-agent arithmetic runs beside the store issue instead of after it.
    const bool spec = early && BLOCK > 64 && p.spec;
    if (early && !spec) stream_graph_fn<BLOCK, AP>(p, l, Gv, n0, tid, tid, BLOCK, true, any_mask);
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
                const double px = v.ex[i], py = v.ey[i];
                const double* row = v.Dm + (size_t)i * E;
                if (rotinv) write_obs_rot<AP>(p, v, i, ph1); else write_obs<AP>(p, v, i, v.vox[i], v.voy[i], ph1);
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
                for (int q = 0; q < ncol_r; ++q) rew -= c.collision_rew * 4;
                nac += ncol_i;
                STAMP(14);
                const bool obst_hit = obstacle_collision_ego(p, v, i);
                if (obst_hit) { rew -= c.collision_rew * 3; noc += 1; }
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
                } else if (rotinv) {
                    // Scenario.reward, rot_inv.py:1122-1338
                    const double Lt = v.tube[T_L], hw = v.tube[T_HALFW];
                    double ts, ty; tube_sy(v.tube, px, py, ts, ty);
                    if (cp == 2 && cp > prev_phase + 1) rew -= c.goal_rew;
                    if (cp == prev_phase + 1 && phase_reached == cp - 1) {
                        if (cp == 1 && in_entrance_gate(ts, ty, Lt, hw) && cooldown == 0) {
                            rew += c.goal_rew;
                            cooldown = (int)((double)c.episode_length / 10);       // float into an int32 array (:1200, :228)
                            phase_reached = 1;
                        } else if (cp == 2) { rew += c.goal_rew; phase_reached = 2; }
                    }
                    if (cp == 0) rew -= entrance_gate_distance(ts, ty, hw);
                    else if (cp == 1) {
                        const double tdx = v.tube[T_EXX] - v.tube[T_ENTX], tdy = v.tube[T_EXY] - v.tube[T_ENTY];
                        const double tlen = sqrt(tdx * tdx + tdy * tdy);
                        const double proj = (px - v.tube[T_ENTX]) * (tdx / tlen) + (py - v.tube[T_ENTY]) * (tdy / tlen);
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
                        const double gain = c.goal_rew / (c.world_size * 0.8 * 10);             // :522
                        const double dproj = proj - pproj;
                        rew += gain * (dproj > -0.05 ? dproj : -0.05);
                        sic += 1;
                        pproj = (double)(float)proj;                                           // float32 array (:374)
                    } else if (cp == 2 && phase_reached == 0) cp = 0;
                    else if (cp == 2) {
                        if (dgoal < c.goal_thresh) { if (me_new) rew += c.goal_rew * 5; }
                        else rew -= dgoal;
                    }
                    if (phase_reached == 1 && cp == 0) conf += 1;
                    if (cp > phase_reached) phase_reached = cp;
                    if (cp < prev_phase) rew -= c.collision_rew;
                    if (cp < phase_reached) rew -= c.collision_rew;
                    prev_phase = cp;
                    if (in_tube_rect(ts, ty, Lt, hw) && cp != 1) rew -= c.collision_rew;
                    if (ts > Lt && phase_reached < 1) rew -= c.goal_rew;
                } else {
                    if (dgoal < c.goal_thresh) { if (me_new) rew += c.goal_rew * 5; }
                    else rew -= dgoal;
                }
                rew = clipd(rew, -4 * c.collision_rew, c.goal_rew * 5);
                if (!rotinv) rew = clipd(rew, c.min_reward, c.max_reward);                      // rot_inv.py:1338 clips once
                v.serr[i] = serr; v.rew[i] = rew;

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
                if (p.o.reward) p.o.reward[na] = (float)(c.collaborative ? rsum : rew);
                if (p.o.done) p.o.done[na] = done ? 1 : 0;
                if (p.o.info) {
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
                    if (july || rotinv) for (int a = 0; a <= i; ++a) dsp += v.serr[a];   // same order as the list append (:1180)
                    const double dm = sd / A, tm = st / A;
                    // population variance = (A*sum(x^2) - sum(x)^2) / A^2, numerator exact
                    const double dvn = (double)A * sdd - sd * sd, tvn = (double)A * stt - st * st;
                    const double ds = sqrt(dvn) / A, ts = sqrt(tvn) / A;
                    float* o = p.o.info + na * GMPE_INFO_KEYS;
                    o[0] = (float)rew; o[1] = (float)dleft; o[2] = (float)trq; o[3] = (float)nac; o[4] = (float)noc;
                    o[5] = (float)dm; o[6] = (float)ds; o[7] = (float)(dm / (ds + 0.0001)); o[8] = (float)dtg;
                    o[9] = (float)trq; o[10] = (float)tm; o[11] = (float)ts; o[12] = (float)(tm / (ts + 0.0001));
                    o[13] = (float)((double)conf / c.episode_length);
                    o[14] = (float)(dsp / (ssv != 0 ? (double)ssv : 1.0));
                    o[15] = (float)((double)sv / (sic != 0 ? sic : 1));
                    o[16] = (float)gmt;
                    o[17] = (float)phase_reached;
                }
                if (!all_done) {                                            // persist the stepped state
                    if (i == 0) {
                        double dsp = dsp0;
                        if (july || rotinv) for (int a = 0; a < A; ++a) dsp += v.serr[a];
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
            if (spec) {                                                 // wave-local: the rows were written by this wave's own lanes
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (p.o.obs) {
                    float* dst = p.o.obs + (size_t)n0 * A * D;
                    const int AD = A * D;
                    for (int q = tid; q < Gv * AD; q += 64) { const int gg = fdiv(q, AD, p.m_AD); if (l.flags[gg * 4 + 3]) dst[q] = l.obs[(size_t)gg * AD4 + (q - gg * AD)]; }
                }
                if (p.o.agent_id) for (int q = tid; q < Gv * A; q += 64) { const int gg = fdiv(q, A, p.m_A); if (l.flags[gg * 4 + 3]) p.o.agent_id[(size_t)n0 * A + q] = q - gg * A; }
            }
        }
        else stream_graph_fn<BLOCK, AP>(p, l, Gv, n0, tid, tid - 64, BLOCK - 64, false, any_mask);
        __syncthreads();
    }
    {
        // ---- 5. reset (explicit, or the worker's auto-reset when every agent of the env is done)
        if (any_reset) {
            const bool mine = ag && v.flags[0];
            if (mine && i == 0) {                                             // one lane per resetting env
                int64_t ctr = ctr0 + (step ? v.flags[1] : 0);                // this step's heading re-draws come first
                reset_world_serial(p, v, n, ctr, err);
                p.s.rng_ctr[n] = ctr;
                p.s.current_step[n] = 0;
                p.s.delta_spacing[n] = 0.0;
                v.flags[2] = 0;
            }
            __syncthreads();
            if (mine) {
                v.s2[i] = v.n2[i]; v.s3[i] = v.n3[i];
                double vx, vy; vel_of(c, v.n2[i], v.n3[i], vx, vy);
                v.vox[i] = v.vnx[i] = vx; v.voy[i] = v.vny[i] = vy;
                v.s_old[i] = 0; v.newf[i] = 0; v.gt[i] = -1; v.moff[i] = 0; v.moff[A + i] = 0;
                int prevA = prev_phase, ph = 0;
                if (july) ph = phase_eval(v.tube, v.ex[i], v.ey[i], prev_phase, prevA);   // reset-time observation (:1447)
                prev_phase = prevA;
                if (rotinv) { ph = phase_eval_rot(v.tube, v.ex[i], v.ey[i], prev_phase, 0); double sn_, cn_; sincos(v.n2[i], &sn_, &cn_); v.cn[i] = cn_; v.sn[i] = sn_; p.s.prev_proj[na] = 0.0; }
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
            __syncthreads();                                                // positions of all agents final
            distance_pass<BLOCK>(p, l, Gv, tid, true);
            static_block<BLOCK>(p, l, Gv, tid, true);
            __syncthreads();
            if (mine) { if (rotinv) write_obs_rot<AP>(p, v, i, ph1); else write_obs<AP>(p, v, i, v.vox[i], v.voy[i], ph1); }
            any_mask = 0;
            for (int gg = 0; gg < Gv; ++gg) any_mask |= l.flags[gg * 4 + 2];
        }
        STAMP(8);
    }
    if (!early) stream_graph_fn<BLOCK, AP>(p, l, Gv, n0, tid, tid, BLOCK, true, any_mask);
    // ---- small outputs: obs staging rows and agent ids. In a specialised tile wave 0 has already stored them
    // (it wrote the staging rows itself), so nobody waits behind the barrier for the streaming waves.
    STAMP(11);
    if (!spec) {
        if (p.o.obs) {
            float* dst = p.o.obs + (size_t)n0 * A * D;
            const int AD = A * D;
            for (int q = tid; q < Gv * AD; q += BLOCK) { const int gg = fdiv(q, AD, p.m_AD); if (l.flags[gg * 4 + 3]) dst[q] = l.obs[(size_t)gg * AD4 + (q - gg * AD)]; }
        }
        if (p.o.agent_id) for (int q = tid; q < Gv * A; q += BLOCK) { const int gg = fdiv(q, A, p.m_A); if (l.flags[gg * 4 + 3]) p.o.agent_id[(size_t)n0 * A + q] = q - gg * A; }   // get_id :1554
    }
    if (ag && err) atomicOr(&p.s.error_flags[n], err);
    STAMP(12);
}

}  // namespace gmpe

namespace gmpe {
// ---------------------------------------------------------------- learner-side edge set
// process_adj (onpolicy/algorithms/utils/gnn_new.py:329-358): mask = (adj < d) & (adj > 0) on fp32
// (inclusive=1 gives update_graph's `<=`, …_july.py:1660), edges in (batch,row,col) order, node ids
// offset by batch*E. Three launches: per-graph count (a wave per graph), chunked scan, ordered per-graph compaction.
__device__ __forceinline__ bool edge_pred(float v, float d, int inclusive) {
    return (inclusive ? v <= d : v < d) && v > 0.0f;
}
// one WAVE per graph (4 graphs per 256-thread block): no LDS, no barrier
__global__ __launch_bounds__(256) void k_edge_count(const float* __restrict__ adj, int B, int EE, float d, int inclusive, int32_t* __restrict__ counts) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* g = adj + (size_t)b * EE;
    int c = 0;
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        c += __popcll(__ballot(q < EE && edge_pred(g[q < EE ? q : 0], d, inclusive)));
    }
    if (lane == 0) counts[b] = c;
}
// exclusive scan of the per-graph counts: one 1024-thread block per chunk of 1024 graphs. The chunk's base is the sum
// of ALL preceding counts, re-reduced by the block itself (L2-resident, <= B reads: no atomics, no extra pass); inside
// the chunk a wave-shuffle scan + 16 wave totals.
__global__ __launch_bounds__(1024) void k_edge_scan(const int32_t* __restrict__ counts, int B, int32_t* __restrict__ offsets, int32_t* __restrict__ total) {
    __shared__ int wtot[16];
    __shared__ int wbase[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int first = blockIdx.x * 1024, b = first + t;
    int part = 0;
    for (int q = t; q < first; q += 1024) part += counts[q];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    const int c = b < B ? counts[b] : 0;
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 0) wbase[w] = part;
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int pre = 0;
    for (int k = 0; k < 16; ++k) pre += wbase[k];
    for (int k = 0; k < w; ++k) pre += wtot[k];
    if (b < B) offsets[b] = pre + incl - c;
    if (blockIdx.x == gridDim.x - 1 && t == 1023) *total = pre + incl;
}
__global__ __launch_bounds__(256) void k_edge_write(const float* __restrict__ adj, int B, int E, float d, int inclusive,
                                                    const int32_t* __restrict__ offsets, int32_t* __restrict__ edge_index,
                                                    float* __restrict__ edge_attr, int cap) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int EE = E * E;
    const float* g = adj + (size_t)b * EE;
    int base = offsets[b];                                   // wave-uniform running position: (batch,row,col) order
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        const float v = q < EE ? g[q] : 0.0f;
        const bool f = q < EE && edge_pred(v, d, inclusive);
        const unsigned long long bal = __ballot(f);
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (f && pos < cap) {
            const int r = q / E, cc = q - r * E;
            edge_index[pos] = b * E + r;
            edge_index[cap + pos] = b * E + cc;
            edge_attr[pos] = v;
        }
        base += __popcll(bal);
    }
}

}  // namespace gmpe

// =================================================================== host side / C ABI
using namespace gmpe;

struct gmpe_handle {
    gmpe_config c;
    int device;
    DevState s;
    int A, L, O, E, D, F;
    std::vector<void*> allocs;
    bool timing = false;
    std::vector<hipEvent_t> ev;      // pairs
    size_t ev_used = 0;
    double t_total_ms = 0;
    int64_t t_launches = 0;
    int block = 0;
    int G = 1;                       // envs per workgroup
    int ablate = 0;
    int nt = 0;
    int spec = 0;
    unsigned long long* stamps = nullptr;
    hipEvent_t region_ev[2] = {nullptr, nullptr};
    int32_t* edge_ws = nullptr;      // [cap_graphs] counts | [cap_graphs] offsets | [cap_graphs/1024+2] chunk sums
    size_t edge_ws_graphs = 0;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(GMPE_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// All persistent state lives in ONE slab (256-B aligned sub-arrays): a tile's ~30 state loads then touch a
// handful of pages instead of one page per field.
template <typename T>
static void slab_take(char* base, size_t& off, T** p, size_t n, int fill_byte, std::vector<std::pair<size_t, std::pair<size_t, int>>>& fills) {
    const size_t bytes = (n ? n : 1) * sizeof(T);
    *p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    fills.push_back({off, {bytes, fill_byte}});
    off = (off + bytes + 255) / 256 * 256;
}
extern "C" {

int gmpe_abi_version(void) { return GMPE_ABI_VERSION; }
const char* gmpe_last_error(void) { return g_err.c_str(); }
int gmpe_obs_dim(const gmpe_config* c) { return c->scenario == GMPE_SCENARIO_TUBE_JULY ? 19 : 13; }
int gmpe_node_feats(const gmpe_config* c) { return c->scenario == GMPE_SCENARIO_ROT_INV ? 7 : GMPE_NODE_FEATS; }
int gmpe_num_entities(const gmpe_config* c) { return c->num_agents + c->num_landmarks + c->num_obstacles; }

static int field_info(const gmpe_handle* h, int f, void** ptr, size_t* bytes) {
    const size_t N = h->c.num_envs, NA = N * h->A;
    const DevState& s = h->s;
    switch (f) {
        case GMPE_F_X: *ptr = s.x; *bytes = NA * 8; break;
        case GMPE_F_Y: *ptr = s.y; *bytes = NA * 8; break;
        case GMPE_F_S2: *ptr = s.s2; *bytes = NA * 8; break;
        case GMPE_F_S3: *ptr = s.s3; *bytes = NA * 8; break;
        case GMPE_F_P_DIST: *ptr = s.p_dist; *bytes = NA * 8; break;
        case GMPE_F_TIME: *ptr = s.time; *bytes = NA * 8; break;
        case GMPE_F_STATUS: *ptr = s.status; *bytes = NA; break;
        case GMPE_F_PREV_PHASE: *ptr = s.prev_phase; *bytes = NA * 4; break;
        case GMPE_F_PHASE_REACHED: *ptr = s.phase_reached; *bytes = NA * 4; break;
        case GMPE_F_COOLDOWN: *ptr = s.cooldown; *bytes = NA * 4; break;
        case GMPE_F_GOAL_TRACKER: *ptr = s.goal_tracker; *bytes = NA * 4; break;
        case GMPE_F_CURRENT_STEP: *ptr = s.current_step; *bytes = N * 4; break;
        case GMPE_F_RNG_CTR: *ptr = s.rng_ctr; *bytes = N * 8; break;
        case GMPE_F_TUBE: *ptr = s.tube; *bytes = N * GMPE_TUBE_STRIDE * 8; break;
        case GMPE_F_LANDMARKS: *ptr = s.landmarks; *bytes = N * h->L * 2 * 8; break;
        case GMPE_F_OBSTACLES: *ptr = s.obstacles; *bytes = N * h->O * 2 * 8; break;
        case GMPE_F_TIMES_REQUIRED: *ptr = s.times_required; *bytes = NA * 4; break;
        case GMPE_F_DISTS_TO_GOAL: *ptr = s.dists_to_goal; *bytes = NA * 4; break;
        case GMPE_F_DIST_LEFT: *ptr = s.dist_left; *bytes = NA * 4; break;
        case GMPE_F_GOAL_REACHED: *ptr = s.goal_reached; *bytes = NA * 4; break;
        case GMPE_F_N_AGENT_COLL: *ptr = s.n_agent_coll; *bytes = NA * 4; break;
        case GMPE_F_N_OBST_COLL: *ptr = s.n_obst_coll; *bytes = NA * 4; break;
        case GMPE_F_SPACING_VIOL: *ptr = s.spacing_viol; *bytes = NA * 4; break;
        case GMPE_F_STEPS_IN_CORR: *ptr = s.steps_in_corr; *bytes = NA * 4; break;
        case GMPE_F_CONFORMANCE: *ptr = s.conformance; *bytes = NA * 4; break;
        case GMPE_F_GOAL_MIN_TIME: *ptr = s.goal_min_time; *bytes = NA * 8; break;
        case GMPE_F_DELTA_SPACING: *ptr = s.delta_spacing; *bytes = N * 8; break;
        case GMPE_F_ERROR_FLAGS: *ptr = s.error_flags; *bytes = N * 4; break;
        case GMPE_F_PREV_PROJ: *ptr = s.prev_proj; *bytes = NA * 8; break;
        default: return fail(GMPE_ERR_INVALID_ARG, "unknown field id");
    }
    return GMPE_OK;
}

int gmpe_create(const gmpe_config* cfg, int device, gmpe_handle** out) {
    if (!cfg || !out) return fail(GMPE_ERR_INVALID_ARG, "null argument");
    if (cfg->abi_version != GMPE_ABI_VERSION) return fail(GMPE_ERR_INVALID_ARG, "gmpe_config.abi_version mismatch");
    const int E = gmpe_num_entities(cfg);
    if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_agents > GMPE_MAX_AGENTS || cfg->num_landmarks < cfg->num_agents ||
        cfg->num_obstacles < 0 || cfg->num_walls < 0 || cfg->num_walls > GMPE_MAX_WALLS || E > GMPE_MAX_ENTITIES)
        return fail(GMPE_ERR_INVALID_ARG, "config out of range (agents<=64, entities<=160, walls<=8, landmarks>=agents)");
    if (cfg->scenario != GMPE_SCENARIO_NAVIGATION_GRAPH && cfg->scenario != GMPE_SCENARIO_TUBE_JULY && cfg->scenario != GMPE_SCENARIO_ROT_INV)
        return fail(GMPE_ERR_UNSUPPORTED, "unknown scenario");
    if ((cfg->scenario != GMPE_SCENARIO_NAVIGATION_GRAPH) == (cfg->dynamics == GMPE_DYN_DOUBLE_INTEGRATOR))
        return fail(GMPE_ERR_UNSUPPORTED, "the tube scenarios are kinematic; navigation_graph is double_integrator");
    if (cfg->dynamics == GMPE_DYN_DOUBLE_INTEGRATOR ? (cfg->n_actions != 5 && cfg->n_actions != 9) : cfg->n_actions != 25)
        return fail(GMPE_ERR_INVALID_ARG, "n_actions does not match the dynamics");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GMPE_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(GMPE_ERR_INVALID_ARG, "bad device index");
    HIPCHK(hipSetDevice(device));
    gmpe_handle* h = new gmpe_handle();
    h->c = *cfg; h->device = device;
    h->A = cfg->num_agents; h->L = cfg->num_landmarks; h->O = cfg->num_obstacles; h->E = E; h->D = gmpe_obs_dim(cfg); h->F = gmpe_node_feats(cfg);
    const size_t N = cfg->num_envs, NA = N * h->A;
    DevState& s = h->s;
    std::vector<std::pair<size_t, std::pair<size_t, int>>> fills;
    char* slab = nullptr;
    for (int pass = 0; pass < 2; ++pass) {                  // pass 0 sizes the slab, pass 1 hands out the pointers
        size_t off = 0;
        fills.clear();
#define AL(p, n, fill) slab_take(slab, off, &(p), (n), (fill), fills);
        AL(s.x, NA, 0) AL(s.y, NA, 0) AL(s.s2, NA, 0) AL(s.s3, NA, 0) AL(s.p_dist, NA, 0) AL(s.time, NA, 0)
        AL(s.status, NA, 0) AL(s.prev_phase, NA, 0) AL(s.phase_reached, NA, 0) AL(s.cooldown, NA, 0)
        AL(s.goal_tracker, NA, 0xFF) AL(s.current_step, N, 0) AL(s.rng_ctr, N, 0)
        AL(s.tube, N * GMPE_TUBE_STRIDE, 0) AL(s.landmarks, N * h->L * 2, 0) AL(s.obstacles, N * h->O * 2, 0)
        AL(s.times_required, NA, 0xFF) AL(s.dists_to_goal, NA, 0xFF) AL(s.dist_left, NA, 0xFF) AL(s.goal_reached, NA, 0xFF)
        AL(s.n_agent_coll, NA, 0) AL(s.n_obst_coll, NA, 0) AL(s.spacing_viol, NA, 0) AL(s.steps_in_corr, NA, 0)
        AL(s.conformance, NA, 0) AL(s.goal_min_time, NA, 0) AL(s.delta_spacing, N, 0) AL(s.error_flags, N, 0) AL(s.prev_proj, NA, 0)
#undef AL
        if (pass == 0) {
            void* q = nullptr;
            hipError_t e = hipMalloc(&q, off);
            if (e != hipSuccess) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, std::string("hipMalloc(state slab): ") + hipGetErrorString(e)); }
            h->allocs.push_back(q);
            slab = static_cast<char*>(q);
        }
    }
    for (auto& f : fills) {
        hipError_t e = hipMemset(slab + f.first, f.second.second, f.second.first);
        if (e != hipSuccess) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e)); }
    }
    s.tape = nullptr; s.tape_len = 0;
    // Tile shape. G envs per workgroup so that the sequential-semantics passes fill wave 0 (G*A <= 64)
    // while the per-tile LDS stays small enough for several workgroups per CU; BLOCK threads share the
    // distance pass and the stores. GMPE_G / GMPE_BLOCK override the heuristic (tuning, tests).
    const char* env_g = getenv("GMPE_G");
    const char* env_block = getenv("GMPE_BLOCK");
    // Heuristic (measured on MI355X, profiles/README.md): the per-agent passes are a dependent fp64 chain, so a tile packs
    // as many envs as fit one wave (G*A <= 64) once there are enough tiles (~700) to cover the chip; multi-wave tiles
    // specialise (wave 0: reward/info, waves 1..: graph stores). C2/C3: G = 6, BLOCK = 256 -> 683 tiles x 4 waves.
    int G = env_g ? atoi(env_g) : (int)(N / 680);
    if (G < 1) G = 1;
    if (G > 64 / h->A) G = 64 / h->A;
    if (G < 1) G = 1;
    if (G > (int)N) G = (int)N;
    while (G > 1 && lds_bytes(G, h->A, E, h->D) > 48 * 1024) --G;
    {   // exact magic division (fdiv) needs q*d < 2^32 for every (range, divisor) pair the kernel uses
        const uint64_t S = (uint64_t)h->L + h->O;
        uint64_t d = (uint64_t)h->A * E;
        const uint64_t cand[] = {(uint64_t)E * E, S * S, (uint64_t)h->A * (h->A + h->O), (uint64_t)h->A * h->D, 2ull * E};
        for (uint64_t x : cand) if (x > d) d = x;
        if ((uint64_t)G * d * d >= (1ull << 32)) { gmpe_destroy(h); return fail(GMPE_ERR_UNSUPPORTED, "tile too large for the index arithmetic"); }
    }
    h->G = G;
    h->ablate = getenv("GMPE_ABLATE") ? atoi(getenv("GMPE_ABLATE")) : 0;
    // Graph outputs per launch vs the 256 MiB Infinity Cache: small launches (C2/C3: 92 MB) are absorbed by it and run
    // faster with ordinary stores (37.0 vs 40.4 us measured); big ones (C4 6 GB, C5 9 GB) stream past it and gain ~10 %
    // from nontemporal stores (C4 1417 -> 1285 us, C5 2005 -> 1852 us).
    {
        const double out_bytes = (double)N * h->A * ((double)E * E + 8.0 * E) * 4.0;
        h->nt = getenv("GMPE_NT") ? atoi(getenv("GMPE_NT")) : (out_bytes > 192.0 * 1024 * 1024 ? 1 : 0);
    }
    const size_t stream_f4 = (size_t)G * h->A * ((size_t)E * E / 4 + 2 * (size_t)E);
    h->block = env_block ? atoi(env_block) : (stream_f4 <= 2048 ? 64 : (stream_f4 <= 6144 ? 128 : 256));
    if (h->block != 64 && h->block != 128 && h->block != 256) h->block = 256;
    // Specialisation pays while the per-agent arithmetic is comparable to the tile's store work (C2/C3: 36.7 -> 33.4 us);
    // store-dominated tiles (C4/C5) want every wave on the stores (C4: 1337 us vs 1421 us specialised).
    h->spec = getenv("GMPE_SPEC") ? atoi(getenv("GMPE_SPEC")) : (stream_f4 <= 16384 ? 1 : 0);
    const size_t lds = lds_bytes(h->G, h->A, E, h->D);
    if (lds > 160 * 1024) { gmpe_destroy(h); return fail(GMPE_ERR_UNSUPPORTED, "per-tile LDS exceeds 160 KiB"); }
    if (lds > 48 * 1024) {                                   // opt in to >64 KiB dynamic LDS (gfx950: 160 KiB per CU)
        hipError_t e = hipSuccess;
#define FN(B, P, W) reinterpret_cast<const void*>(&k_env<B, P, W>)
        const void* fns[18] = {FN(64, 0, true), FN(64, 3, true), FN(64, 10, true), FN(128, 0, true), FN(128, 3, true), FN(128, 10, true),
                               FN(256, 0, true), FN(256, 3, true), FN(256, 10, true), FN(64, 0, false), FN(64, 3, false), FN(64, 10, false),
                               FN(128, 0, false), FN(128, 3, false), FN(128, 10, false), FN(256, 0, false), FN(256, 3, false), FN(256, 10, false)};
#undef FN
        for (int q = 0; q < 18 && e == hipSuccess; ++q) e = hipFuncSetAttribute(fns[q], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { gmpe_destroy(h); return fail(GMPE_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); }
    }
#ifdef GMPE_STAMPS
    {
        const size_t grid = (N + G - 1) / G;
        void* q = nullptr;
        if (hipMalloc(&q, grid * 16 * 8) == hipSuccess) { (void)hipMemset(q, 0, grid * 16 * 8); h->allocs.push_back(q); h->stamps = static_cast<unsigned long long*>(q); }
    }
#endif
    *out = h;
    return GMPE_OK;
}

#ifdef GMPE_STAMPS
// diagnostic build only: copy the per-workgroup phase stamps of the LAST launch to the host
int gmpe_debug_stamps(gmpe_handle* h, unsigned long long* host_dst, int64_t max_blocks) {
    const int64_t grid = (h->c.num_envs + h->G - 1) / h->G;
    const int64_t nb = grid < max_blocks ? grid : max_blocks;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_dst, h->stamps, (size_t)nb * 16 * 8, hipMemcpyDeviceToHost));
    return (int)nb;
}
#endif

int gmpe_destroy(gmpe_handle* h) {
    if (!h) return GMPE_OK;
    (void)hipSetDevice(h->device);
    for (void* q : h->allocs) (void)hipFree(q);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->edge_ws) (void)hipFree(h->edge_ws);
    for (hipEvent_t e : h->region_ev) if (e) (void)hipEventDestroy(e);
    delete h;
    return GMPE_OK;
}

int gmpe_set_rng_tape(gmpe_handle* h, const double* tape_dev, int64_t len_per_env) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    h->s.tape = tape_dev; h->s.tape_len = tape_dev ? len_per_env : 0;
    return GMPE_OK;
}

int gmpe_field_bytes(const gmpe_handle* h, int field, size_t* bytes) {
    void* p; return field_info(h, field, &p, bytes);
}
int gmpe_get_field(gmpe_handle* h, int field, void* host_dst, size_t bytes) {
    void* p; size_t b;
    int rc = field_info(h, field, &p, &b); if (rc) return rc;
    if (b != bytes) return fail(GMPE_ERR_INVALID_ARG, "gmpe_get_field: size mismatch");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    if (b) HIPCHK(hipMemcpy(host_dst, p, b, hipMemcpyDeviceToHost));
    return GMPE_OK;
}
int gmpe_set_field(gmpe_handle* h, int field, const void* host_src, size_t bytes) {
    void* p; size_t b;
    int rc = field_info(h, field, &p, &b); if (rc) return rc;
    if (b != bytes) return fail(GMPE_ERR_INVALID_ARG, "gmpe_set_field: size mismatch");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    if (b) HIPCHK(hipMemcpy(p, host_src, b, hipMemcpyHostToDevice));
    return GMPE_OK;
}

static int launch(gmpe_handle* h, int mode, const int32_t* act, const float* onehot, const uint8_t* mask,
                  const gmpe_outputs* out, void* stream) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    KParams p;
    memset(&p, 0, sizeof p);
    p.c = h->c; p.s = h->s;
    if (out) p.o = *out;
    p.act = act; p.onehot = onehot; p.mask = mask; p.mode = mode;
    p.A = h->A; p.L = h->L; p.O = h->O; p.E = h->E; p.D = h->D; p.F = h->F; p.G = h->G;
    p.ablate = h->ablate;
    p.nt = h->nt;
    p.spec = h->spec;
    p.stamps = h->stamps;
    p.m_E = magic_of(p.E); p.m_AE = magic_of(p.A * p.E); p.m_EE = magic_of(p.E * p.E); p.m_nq = magic_of(p.E * p.E / 4);
    p.m_2E = magic_of(2 * p.E); p.m_pe = magic_of(p.A * p.E * 2); p.m_AD = magic_of(p.A * p.D); p.m_A = magic_of(p.A);
    p.m_L = magic_of(p.L); p.m_O = magic_of(p.O);
    p.m_S = magic_of(p.L + p.O); p.m_SS = magic_of((p.L + p.O) * (p.L + p.O)); p.m_C = magic_of(p.A + p.O);
    p.m_AC = magic_of(p.A * (p.A + p.O)); p.m_AEE = magic_of(p.A * p.E * p.E);
    p.m_W = magic_of(p.A * (p.A - 1) / 2 + p.A * (p.E - p.A)); p.m_Sx = magic_of(p.E - p.A); p.m_FW = magic_of(p.A * (p.A - 1) / 2 + p.A * p.O);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = lds_bytes(h->G, h->A, h->E, h->D);
    const dim3 grid((h->c.num_envs + h->G - 1) / h->G);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        if (h->ev_used + 2 > h->ev.size()) {
            for (int q = 0; q < 2; ++q) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
        }
        e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1];
        HIPCHK(hipEventRecord(e0, st));
    }
    const int ap = (h->A == h->L && (h->A == 10 || h->A == 3)) ? h->A : 0;   // exact-size instantiations of the common cases
    const bool walls = h->c.num_walls > 0;
#define LAUNCH_ENV(B) do { \
        if (walls) { if (ap == 10) hipLaunchKernelGGL((k_env<B, 10, true>), grid, dim3(B), lds, st, p); else if (ap == 3) hipLaunchKernelGGL((k_env<B, 3, true>), grid, dim3(B), lds, st, p); else hipLaunchKernelGGL((k_env<B, 0, true>), grid, dim3(B), lds, st, p); } \
        else { if (ap == 10) hipLaunchKernelGGL((k_env<B, 10, false>), grid, dim3(B), lds, st, p); else if (ap == 3) hipLaunchKernelGGL((k_env<B, 3, false>), grid, dim3(B), lds, st, p); else hipLaunchKernelGGL((k_env<B, 0, false>), grid, dim3(B), lds, st, p); } \
    } while (0)
    switch (h->block) {
        case 64: LAUNCH_ENV(64); break;
        case 128: LAUNCH_ENV(128); break;
        default: LAUNCH_ENV(256); break;
    }
    HIPCHK(hipGetLastError());
    if (h->timing) { HIPCHK(hipEventRecord(e1, st)); h->ev_used += 2; }
    return GMPE_OK;
}

int gmpe_reset(gmpe_handle* h, const uint8_t* env_mask_dev, const gmpe_outputs* out, void* stream) {
    return launch(h, MODE_RESET, nullptr, nullptr, env_mask_dev, out, stream);
}
int gmpe_step(gmpe_handle* h, const int32_t* action_idx_dev, const gmpe_outputs* out, void* stream) {
    if (!action_idx_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step: null actions");
    return launch(h, MODE_STEP, action_idx_dev, nullptr, nullptr, out, stream);
}
int gmpe_step_many(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets,
                   const gmpe_outputs* out, void* stream) {
    if (!actions_dev || num_steps < 0 || num_action_sets < 1) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_many: bad arguments");
    const size_t stride = (size_t)h->c.num_envs * h->A;
    for (int32_t k = 0; k < num_steps; ++k) {
        const int rc = launch(h, MODE_STEP, actions_dev + (size_t)(k % num_action_sets) * stride, nullptr, nullptr, out, stream);
        if (rc) return rc;
    }
    return GMPE_OK;
}
int gmpe_step_onehot(gmpe_handle* h, const float* onehot_dev, const gmpe_outputs* out, void* stream) {
    if (!onehot_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_step_onehot: null actions");
    return launch(h, MODE_STEP, nullptr, onehot_dev, nullptr, out, stream);
}


int gmpe_edges_from_adj(gmpe_handle* h, const float* adj_dev, int32_t batch, int32_t num_nodes, float max_edge_dist,
                        int32_t inclusive, int32_t* edge_index_dev, float* edge_attr_dev, int32_t cap,
                        int32_t* n_edges_dev, void* stream) {
    if (!h || !adj_dev || !edge_index_dev || !edge_attr_dev || !n_edges_dev) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj: null argument");
    if (batch < 1 || num_nodes < 1 || cap < 0) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj: bad sizes");
    if ((int64_t)batch * num_nodes > INT32_MAX) return fail(GMPE_ERR_INVALID_ARG, "gmpe_edges_from_adj: node ids overflow int32");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h->edge_ws_graphs < (size_t)batch) {                 // workspace grows on first use only
        if (h->edge_ws) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(h->edge_ws)); h->edge_ws = nullptr; }
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->edge_ws), sizeof(int32_t) * (2 * (size_t)batch + (size_t)batch / 1024 + 2)));
        h->edge_ws_graphs = batch;
    }
    int32_t* counts = h->edge_ws; int32_t* offsets = h->edge_ws + h->edge_ws_graphs;
    const int EE = num_nodes * num_nodes, nchunks = (batch + 1023) / 1024;
    hipLaunchKernelGGL(k_edge_count, dim3((batch + 3) / 4), dim3(256), 0, st, adj_dev, batch, EE, max_edge_dist, inclusive, counts);
    hipLaunchKernelGGL(k_edge_scan, dim3(nchunks), dim3(1024), 0, st, counts, batch, offsets, n_edges_dev);
    hipLaunchKernelGGL(k_edge_write, dim3((batch + 3) / 4), dim3(256), 0, st, adj_dev, batch, num_nodes, max_edge_dist, inclusive, offsets,
                       edge_index_dev, edge_attr_dev, cap);
    HIPCHK(hipGetLastError());
    return GMPE_OK;
}

int gmpe_timing_mark(gmpe_handle* h, int32_t which, void* stream) {
    if (!h || which < 0 || which > 1) return fail(GMPE_ERR_INVALID_ARG, "gmpe_timing_mark: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (!h->region_ev[which]) HIPCHK(hipEventCreate(&h->region_ev[which]));
    HIPCHK(hipEventRecord(h->region_ev[which], static_cast<hipStream_t>(stream)));
    return GMPE_OK;
}
int gmpe_timing_region_ms(gmpe_handle* h, double* ms) {
    if (!h || !ms || !h->region_ev[0] || !h->region_ev[1]) return fail(GMPE_ERR_INVALID_ARG, "gmpe_timing_region_ms: region not marked");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->region_ev[1]));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, h->region_ev[0], h->region_ev[1]));
    *ms = f;
    return GMPE_OK;
}
int gmpe_timing_enable(gmpe_handle* h, int32_t enable) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    h->timing = enable != 0;
    return GMPE_OK;
}
int gmpe_timing_read(gmpe_handle* h, double* total_ms, int64_t* launches, int32_t reset_counters) {
    if (!h) return fail(GMPE_ERR_INVALID_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    for (size_t q = 0; q + 1 < h->ev_used; q += 2) {
        HIPCHK(hipEventSynchronize(h->ev[q + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->ev[q], h->ev[q + 1]));
        h->t_total_ms += ms; h->t_launches += 1;
    }
    h->ev_used = 0;
    if (total_ms) *total_ms = h->t_total_ms;
    if (launches) *launches = h->t_launches;
    if (reset_counters) { h->t_total_ms = 0; h->t_launches = 0; }
    return GMPE_OK;
}

}  // extern "C"
