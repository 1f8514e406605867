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
    int A, L, O, E, D;
};

// ---------------------------------------------------------------- LDS carve (dynamic, 16-B aligned)
struct Lds {
    double *ex, *ey;                  // [E]  entity positions (agents: post-integration)
    double *s2, *s3;                  // [A]  theta/speed or vx/vy BEFORE this step's reward loop
    double *n2, *n3;                  // [A]  ... AFTER it (reset_velocity on goal reach)
    double *vox, *voy, *vnx, *vny;    // [A]  p_vel before / after
    double *serr;                     // [A]  spacing error of this step (…_july.py:1168-1180)
    double *rew;                      // [A]
    double *tube;                     // [12]
    int *s_old, *newf, *gt;           // [A]  status before, newly-reached flag, goal_tracker (final)
    int *dtg_o, *dtg_n, *trq_o, *trq_n, *sv_o, *sv_n;   // [A] info counters old/new
    int *flags;                       // [4]
    float *obs;                       // [A*D] staging
    float *M;                         // [E*E] masked distance matrix, fp32
};
__host__ __device__ inline size_t lds_bytes(int A, int E, int D) {
    size_t d = (size_t)2 * E + 11 * A + 12;              // doubles
    size_t i = (size_t)9 * A + 4;                        // ints
    size_t f = (size_t)A * D + (size_t)E * E;            // floats
    return d * 8 + ((i * 4 + 15) / 16) * 16 + ((f * 4 + 15) / 16) * 16 + 64;
}
__device__ inline Lds carve(char* base, int A, int E, int D) {
    Lds l;
    double* d = reinterpret_cast<double*>(base);
    l.ex = d; d += E; l.ey = d; d += E;
    l.s2 = d; d += A; l.s3 = d; d += A; l.n2 = d; d += A; l.n3 = d; d += A;
    l.vox = d; d += A; l.voy = d; d += A; l.vnx = d; d += A; l.vny = d; d += A;
    l.serr = d; d += A; l.rew = d; d += A; l.tube = d; d += 12;
    if ((uintptr_t)d & 15) d += 1;
    float* f = reinterpret_cast<float*>(d);
    l.M = f; f += ((size_t)E * E + 3) / 4 * 4;
    l.obs = f; f += ((size_t)A * D + 3) / 4 * 4;
    int* i = reinterpret_cast<int*>(f);
    l.s_old = i; i += A; l.newf = i; i += A; l.gt = i; i += A;
    l.dtg_o = i; i += A; l.dtg_n = i; i += A; l.trq_o = i; i += A; l.trq_n = i; i += A;
    l.sv_o = i; i += A; l.sv_n = i; i += A; l.flags = i;
    return l;
}

__device__ __forceinline__ bool kinematic(const gmpe_config& c) { return c.dynamics != GMPE_DYN_DOUBLE_INTEGRATOR; }

__device__ __forceinline__ void vel_of(const gmpe_config& c, double a2, double a3, double& vx, double& vy) {
    if (kinematic(c)) { vx = a3 * cos(a2); vy = a3 * sin(a2); }      // core.py:281-286
    else { vx = a2; vy = a3; }                                       // core.py:191-193
}

// Scenario.is_obstacle_collision (…_july.py:864-890)
__device__ inline bool obstacle_collision(const KParams& p, const Lds& l, double px, double py, double size) {
    const int o0 = p.A + p.L;
    for (int o = 0; o < p.O; ++o)
        if (norm2(l.ex[o0 + o] - px, l.ey[o0 + o] - py) < 2.0 * (p.c.entity_size + size)) return true;
    for (int w = 0; w < p.c.num_walls; ++w) {
        const gmpe_wall& wl = p.c.walls[w];
        const double band = 1.5 * size;
        const double perp = wl.orient == 0 ? py : px, prll = wl.orient == 0 ? px : py;
        if (wl.axis_pos - band <= perp && perp <= wl.axis_pos + band && wl.end0 - band <= prll && prll <= wl.end1 + band)
            return true;
    }
    return false;
}

// _set_action (environment.py:336-475)
__device__ inline void decode_action(const gmpe_config& c, int idx, double& u0, double& u1) {
    if (c.dynamics == GMPE_DYN_DOUBLE_INTEGRATOR) {
        if (c.n_actions == 5) {
            u0 = (idx == 1 ? 1.0 : 0.0) - (idx == 2 ? 1.0 : 0.0);
            u1 = (idx == 3 ? 1.0 : 0.0) - (idx == 4 ? 1.0 : 0.0);
        } else {
            const double m0[9] = {0, -1, -0.71, 0, 0.71, 1, 0.71, 0, -0.71};
            const double m1[9] = {0, 0, -0.71, -1, -0.71, 0, 0.71, 1, 0.71};
            u0 = m0[idx]; u1 = m1[idx];
        }
    } else {
        const int wi = idx / 5, ai = idx - wi * 5;
        u0 = c.ang_rate_opt[wi]; u1 = c.accel_opt[ai];
    }
    u0 *= c.sensitivity; u1 *= c.sensitivity;
}

// Scenario.observation (…_july.py:1337-1463) for ego i into the LDS staging row; `phase` = value of
// the first get_agent_phase call of the step. Uses the PRE-reward own velocity (vox/voy).
__device__ inline void write_obs(const KParams& p, const Lds& l, int i, double vx, double vy, int phase) {
    float* o = l.obs + (size_t)i * p.D;
    const double px = l.ex[i], py = l.ey[i];
    const double gx = l.ex[p.A + i] - px, gy = l.ey[p.A + i] - py;
    o[0] = (float)px; o[1] = (float)py; o[2] = (float)vx; o[3] = (float)vy;
    o[4] = (float)gx; o[5] = (float)gy; o[6] = 0.0f; o[7] = (float)gx; o[8] = (float)gy;
    int b1 = -1, b2 = -1; double d1 = 0, d2 = 0;                   // stable two-smallest (1398-1417)
    for (int k = 0; k < p.A; ++k) {
        if (k == i) continue;
        const double d = norm2(l.ex[k] - px, l.ey[k] - py);
        if (b1 < 0 || d < d1) { b2 = b1; d2 = d1; b1 = k; d1 = d; }
        else if (b2 < 0 || d < d2) { b2 = k; d2 = d; }
    }
    o[9] = b1 >= 0 ? (float)(l.ex[b1] - px) : 0.f; o[10] = b1 >= 0 ? (float)(l.ey[b1] - py) : 0.f;
    o[11] = b2 >= 0 ? (float)(l.ex[b2] - px) : 0.f; o[12] = b2 >= 0 ? (float)(l.ey[b2] - py) : 0.f;
    if (p.c.scenario == GMPE_SCENARIO_TUBE_JULY) {
        o[13] = (float)(l.tube[T_ENTX] - px); o[14] = (float)(l.tube[T_ENTY] - py);
        o[15] = (float)(l.tube[T_EXX] - px); o[16] = (float)(l.tube[T_EXY] - py);
        o[17] = (float)l.tube[T_WIDTH]; o[18] = (float)phase;
    }
}

// Serial reset of one env by ONE lane (reset_world: …_july.py:339-420, 440-515, 518-613,
// custom_scenarios/utils.py:165-193; navigation_graph: DESIGN.md). Writes positions / headings to
// LDS (ex, ey, n2, n3, tube) and landmark / obstacle / tube records to HBM. Bounded rejection loop.
__device__ void reset_world_serial(const KParams& p, const Lds& l, int n, int64_t& ctr, int& err) {
    const gmpe_config& c = p.c;
    const double ws = c.world_size, size = c.entity_size;
    const int A = p.A, L = p.L, O = p.O;
    if (c.scenario == GMPE_SCENARIO_TUBE_JULY) {
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
            const double jx = 0.2 * (-ws + (ws - (-ws)) * u0), jy = 0.2 * (-ws + (ws - (-ws)) * u1);
            const double dfe = (ws + k) / 5;
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

// ---------------------------------------------------------------- the fused kernel
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_env(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int A = p.A, L = p.L, O = p.O, E = p.E, D = p.D;
    const gmpe_config& c = p.c;
    if (p.mode == MODE_RESET && p.mask && !p.mask[n]) return;          // block-uniform
    const Lds l = carve(smem, A, E, D);
    const size_t na = (size_t)n * A + tid;
    const bool ag = tid < A;                                            // agent lanes live in wave 0
    const bool july = c.scenario == GMPE_SCENARIO_TUBE_JULY;

    // ---- per-agent registers
    int prev_phase = 0, phase_reached = 0, cooldown = 0;
    double p_dist = 0, tim = 0;
    int trq = -1, dtg = -1, dleft = -1, greached = -1, nac = 0, noc = 0, sv = 0, sic = 0, conf = 0;
    double gmt = 0;
    int cur_step = p.s.current_step[n];
    int err = 0;

    // ---- 0. load state
    if (tid < GMPE_TUBE_STRIDE) l.tube[tid] = p.s.tube[(size_t)n * GMPE_TUBE_STRIDE + tid];
    for (int k = tid; k < L; k += BLOCK) { l.ex[A + k] = p.s.landmarks[((size_t)n * L + k) * 2]; l.ey[A + k] = p.s.landmarks[((size_t)n * L + k) * 2 + 1]; }
    for (int k = tid; k < O; k += BLOCK) { l.ex[A + L + k] = p.s.obstacles[((size_t)n * O + k) * 2]; l.ey[A + L + k] = p.s.obstacles[((size_t)n * O + k) * 2 + 1]; }
    if (ag) {
        prev_phase = p.s.prev_phase[na];
        if (p.mode == MODE_STEP) {
            l.ex[tid] = p.s.x[na]; l.ey[tid] = p.s.y[na]; l.s2[tid] = p.s.s2[na]; l.s3[tid] = p.s.s3[na];
            l.s_old[tid] = p.s.status[na]; l.gt[tid] = p.s.goal_tracker[na];
            phase_reached = p.s.phase_reached[na]; cooldown = p.s.cooldown[na];
            p_dist = p.s.p_dist[na]; tim = p.s.time[na];
            trq = p.s.times_required[na]; dtg = p.s.dists_to_goal[na]; dleft = p.s.dist_left[na];
            greached = p.s.goal_reached[na]; nac = p.s.n_agent_coll[na]; noc = p.s.n_obst_coll[na];
            sv = p.s.spacing_viol[na]; sic = p.s.steps_in_corr[na]; conf = p.s.conformance[na];
            gmt = p.s.goal_min_time[na];
        }
    }
    if (tid == 0) l.flags[0] = (p.mode == MODE_RESET);
    __syncthreads();

    int ph1 = 0;
    if (p.mode == MODE_STEP) {
        cur_step += 1;
        // ---- 1. action decode + dynamics
        double nx = 0, ny = 0, nv2 = 0, nv3 = 0;
        if (ag) {
            int idx;
            if (p.act) idx = p.act[na];
            else {                                                      // np.argmax: first maximum
                const float* oh = p.onehot + na * c.n_actions;
                idx = 0; float best = oh[0];
                for (int q = 1; q < c.n_actions; ++q) if (oh[q] > best) { best = oh[q]; idx = q; }
            }
            idx = idx < 0 ? 0 : (idx >= c.n_actions ? c.n_actions - 1 : idx);
            double u0, u1; decode_action(c, idx, u0, u1);
            nx = l.ex[tid]; ny = l.ey[tid]; nv2 = l.s2[tid]; nv3 = l.s3[tid];
            if (kinematic(c)) {
                if (!l.s_old[tid]) {                                    // update_agent_state core.py:819-826
                    const double dt = c.dt, th0 = nv2, v0 = nv3;
                    const double th1 = th0 + u0 * dt, v1 = v0 + u1 * dt;
                    if (u0 != 0.0) {
                        const double s0 = sin(th0), c0 = cos(th0), s1 = sin(th1), c1 = cos(th1);
                        nx += (v1 * s1 - v0 * s0) / u0 + u1 * (c1 - c0) / (u0 * u0);
                        ny += (-v1 * c1 + v0 * c0) / u0 + u1 * (s1 - s0) / (u0 * u0);
                    } else {
                        const double d = (v0 + 0.5 * u1 * dt) * dt;
                        nx += d * cos(th0); ny += d * sin(th0);
                    }
                    double v = v1;
                    if (v > c.v_max) v = c.v_max;
                    if (v < c.v_min) v = c.v_min;
                    nv2 = th1; nv3 = v;
                    p_dist += v * dt; tim += dt;
                }
            } else {
                // force path core.py:766-845, 872-964: accumulate in the reference's order for this agent:
                // other entities by ascending index (as side b below its own index, side a above), then walls.
                double Fx = 1.0 * u0, Fy = 1.0 * u1;
                const double pax = nx, pay = ny;
                for (int k = 0; k < E; ++k) {
                    if (k == tid) continue;
                    if (k >= A && k < A + L) continue;                  // landmarks: collide=False
                    const bool ego_is_b = k < tid;
                    const double dx = ego_is_b ? l.ex[k] - pax : pax - l.ex[k];
                    const double dy = ego_is_b ? l.ey[k] - pay : pay - l.ey[k];
                    const double dist = sqrt(dx * dx + dy * dy);
                    const double z = -(dist - c.sep_dist) / c.contact_margin;
                    if (z < -50.0) continue;                            // softplus < 1e-21: below one ulp of the sum
                    if (k < A && l.s_old[tid]) continue;                // done side gets no force (899-900)
                    const double pen = logaddexp0(z) * c.contact_margin;
                    const double fx = c.contact_force * dx / dist * pen, fy = c.contact_force * dy / dist * pen;
                    if (ego_is_b) { Fx = -fx + Fx; Fy = -fy + Fy; } else { Fx = fx + Fx; Fy = fy + Fy; }
                }
                for (int w = 0; w < c.num_walls; ++w) {
                    double wx, wy;
                    if (wall_force(c.walls[w], pax, pay, c.entity_size, c.wall_contact_force, c.wall_contact_margin, wx, wy)) { Fx = Fx + wx; Fy = Fy + wy; }
                }
                double vx = nv2 * (1 - c.damping), vy = nv3 * (1 - c.damping);
                vx += (Fx / 1.0) * c.dt; vy += (Fy / 1.0) * c.dt;
                if (c.max_speed > 0) {
                    const double sp = sqrt(vx * vx + vy * vy);
                    if (sp > c.max_speed) { const double q = sqrt(vx * vx + vy * vy); vx = vx / q * c.max_speed; vy = vy / q * c.max_speed; }
                }
                nv2 = vx; nv3 = vy;
                nx += vx * c.dt; ny += vy * c.dt;
                const double ax = vx * c.dt, ay = vy * c.dt;
                p_dist += sqrt(ax * ax + ay * ay); tim += c.dt;
            }
        }
        __syncthreads();                                                // all lanes have read the old positions
        if (ag) { l.ex[tid] = nx; l.ey[tid] = ny; l.s2[tid] = nv2; l.s3[tid] = nv3; }
        __syncthreads();

        // ---- 2. phase FSM + who newly reaches the goal (depends only on own data: SURVEY §8a)
        int cp = 0, prevA = prev_phase;
        bool goal_branch = true;
        double dgoal = 0;
        if (ag) {
            const double px = l.ex[tid], py = l.ey[tid];
            double vx, vy; vel_of(c, l.s2[tid], l.s3[tid], vx, vy);
            l.vox[tid] = vx; l.voy[tid] = vy;
            if (july) {
                ph1 = phase_eval(l.tube, px, py, prev_phase, prevA);     // observation's call (:1447)
                if (cooldown > 0) cooldown -= 1;
                int prevB;
                cp = phase_eval(l.tube, px, py, prevA, prevB);           // reward's call (:1113)
                if (cooldown > 0) cooldown -= 1;
                prevA = prevB;
                goal_branch = (cp == 2 && phase_reached != 0);
            }
            dgoal = norm2(px - l.ex[A + tid], py - l.ey[A + tid]);
            const bool nf = goal_branch && dgoal < c.goal_thresh && !l.s_old[tid];
            l.newf[tid] = nf;
        }
        // rank of each newly-reached agent among them: the heading re-draws follow agent order (core.py:328)
        int64_t ctr0 = p.s.rng_ctr[n];
        if (tid < 64) {
            const unsigned long long bal = __ballot(ag && l.newf[tid]);
            if (ag) {
                if (l.newf[tid]) {
                    const int rank = __popcll(bal & ((1ull << tid) - 1ull));
                    if (kinematic(c)) { l.n2[tid] = 0.0 + (2 * M_PI - 0.0) * draw_at(c, p.s, n, ctr0 + rank, err); l.n3[tid] = c.v_min; }
                    else { l.n2[tid] = 0.0; l.n3[tid] = 0.0; }
                    l.gt[tid] = tid;
                } else { l.n2[tid] = l.s2[tid]; l.n3[tid] = l.s3[tid]; }
                double vx, vy; vel_of(c, l.n2[tid], l.n3[tid], vx, vy);
                l.vnx[tid] = vx; l.vny[tid] = vy;
            }
            if (tid == 0) l.flags[1] = kinematic(c) ? __popcll(bal) : 0;   // draws consumed (DI reset_velocity draws none)
        }
        __syncthreads();
        const int n_new = l.flags[1];

        // ---- 3. obs, reward, done (ego i sees agent k done iff s_old[k] || (new[k] && k < i))
        double rew = 0; bool done = false;
        if (ag) {
            const int i = tid;
            const double px = l.ex[i], py = l.ey[i];
            write_obs(p, l, i, l.vox[i], l.voy[i], ph1);
            // collision block (…_july.py:1117-1124)
            if (!l.s_old[i])
                for (int a = 0; a < A; ++a) {
                    if (a == i) continue;
                    const bool a_done = l.s_old[a] || (l.newf[a] && a < i);
                    if (a_done) continue;
                    if (norm2(l.ex[a] - px, l.ey[a] - py) < c.sep_dist) rew -= c.collision_rew * 4;
                }
            if (obstacle_collision(p, l, px, py, c.entity_size)) rew -= c.collision_rew * 3;
            double serr = 0;
            if (july) {
                const double tdx = l.tube[T_EXX] - l.tube[T_ENTX], tdy = l.tube[T_EXY] - l.tube[T_ENTY];
                const double tlen = sqrt(tdx * tdx + tdy * tdy);
                const double hx = cos(l.s2[i]), hy = sin(l.s2[i]);
                int front = -1, back = -1; double fproj = 0, bproj = 0;
                for (int k = 0; k < A; ++k) {
                    if (k == i) continue;
                    const double proj = (l.ex[k] - px) * hx + (l.ey[k] - py) * hy;
                    if (proj > 0) { if (front < 0 || proj < fproj) { front = k; fproj = proj; } }
                    else { if (back < 0 || proj > bproj) { back = k; bproj = proj; } }
                }
                // prevA here is previous_phase as the reward sees it (after both phase calls)
                if (cp == 2 && cp > prevA + 1) rew -= c.goal_rew * 3;
                const double ux = tdx / tlen, uy = tdy / tlen;
                const double qx = px - l.tube[T_ENTX], qy = py - l.tube[T_ENTY];
                const double proj = qx * ux + qy * uy;
                const double edist = norm2(qx - proj * tdx, qy - proj * tdy);      // un-normalised (:1154)
                if (cp == prevA + 1 && phase_reached == cp - 1) {
                    if (cp == 1 && 0 <= proj && proj < 0.1 * tlen && edist < 0.2 * tlen) rew += c.goal_rew * 3;
                    else if (cp == 2) rew += c.goal_rew * 3;
                }
                if (cp == 0) rew -= norm2(l.tube[T_ENTX] - px, l.tube[T_ENTY] - py);
                else if (cp == 1) {
                    if (front >= 0) { const double df = norm2(l.ex[front] - px, l.ey[front] - py) - c.sep_dist; serr += df < 0 ? fabs(df) : 0; }
                    if (back >= 0) { const double df = norm2(l.ex[back] - px, l.ey[back] - py) - c.sep_dist; serr += df < 0 ? fabs(df) : 0; }
                    if (serr > 0) sv += 1;
                    rew -= serr * c.formation_rew;
                    rew -= norm2(l.tube[T_EXX] - px, l.tube[T_EXY] - py);
                    sic += 1;
                } else if (cp == 2 && phase_reached == 0) cp = 0;
                else {
                    if (dgoal < c.goal_thresh) { if (l.newf[i]) rew += c.goal_rew * 5; }
                    else rew -= dgoal;
                }
                if (phase_reached == 1 && cp == 0) conf += 1;
                if (cp > phase_reached) phase_reached = cp;
                if (cp < prevA) rew -= c.collision_rew * 3;
                if (cp < phase_reached) rew -= c.collision_rew;
                prev_phase = cp;
            } else {
                if (dgoal < c.goal_thresh) { if (l.newf[i]) rew += c.goal_rew * 5; }
                else rew -= dgoal;
            }
            rew = clipd(rew, -4 * c.collision_rew, c.goal_rew * 5);
            rew = clipd(rew, c.min_reward, c.max_reward);
            l.serr[i] = serr; l.rew[i] = rew;
            const bool st = l.s_old[i] || l.newf[i];
            done = st || cur_step >= c.episode_length;                   // _get_done environment.py:264-271

            // ---- info counters that depend on own data only (…_july.py:744-773)
            l.dtg_o[i] = dtg; l.trq_o[i] = trq; l.sv_o[i] = l.sv_n[i] = 0;
            int nearest = 0; double dmin = 0;
            for (int q = 0; q < L; ++q) {
                const double d = norm2(px - l.ex[A + q], py - l.ey[A + q]);
                if (q == 0 || d < dmin) { dmin = d; nearest = q; }
            }
            const double thr = c.goal_thresh;
            const int tnow = (int)((double)cur_step * c.dt);
            if (dmin < thr && (nearest != greached && greached != -1)) { greached = nearest; dleft = (int)dmin; }
            if (dmin < thr && trq == -1) { trq = tnow; dtg = (int)p_dist; dleft = (int)dmin; greached = nearest; }
            if (trq == -1) { dtg = (int)p_dist; dleft = (int)dmin; }
            if (dmin > thr && trq != -1) { dtg = (int)p_dist; trq = tnow; dleft = (int)dmin; }
            if (dmin < thr && nearest == greached) { dleft = (int)dmin; greached = nearest; }
            l.dtg_n[i] = dtg; l.trq_n[i] = trq;
            l.sv_n[i] = sv; l.sv_o[i] = sv - (serr > 0 ? 1 : 0);
        }
        __syncthreads();

        // ---- 4. info (sequential view: agents j<=i already updated, j>i not yet), outputs, write-back
        const bool all_done = __syncthreads_and(ag ? (int)done : 1) != 0;
        if (ag) {
            const int i = tid;
            const double px = l.ex[i], py = l.ey[i];
            if (obstacle_collision(p, l, px, py, c.entity_size)) noc += 1;
            const bool me_done = l.s_old[i] || l.newf[i];
            if (!me_done)
                for (int a = 0; a < A; ++a) {
                    if (a == i) continue;
                    const bool a_done = l.s_old[a] || (l.newf[a] && a <= i);
                    if (!a_done && norm2(px - l.ex[a], py - l.ey[a]) < c.sep_dist) nac += 1;
                }
            double rsum = 0;
            if (c.collaborative) for (int a = 0; a < A; ++a) rsum += l.rew[a];
            if (p.o.reward) p.o.reward[na] = (float)(c.collaborative ? rsum : rew);
            if (p.o.done) p.o.done[na] = done ? 1 : 0;
            if (p.o.info) {
                double dm = 0, tm = 0, svsum = 0, dsp = p.s.delta_spacing[n];
                for (int a = 0; a < A; ++a) {
                    dm += a <= i ? l.dtg_n[a] : l.dtg_o[a];
                    tm += a <= i ? l.trq_n[a] : l.trq_o[a];
                    svsum += a <= i ? l.sv_n[a] : l.sv_o[a];
                }
                for (int a = 0; a <= i; ++a) dsp += l.serr[a];          // same order as the list append
                dm /= A; tm /= A;
                double dv = 0, tv = 0;
                for (int a = 0; a < A; ++a) {
                    const double pq = (a <= i ? l.dtg_n[a] : l.dtg_o[a]) - dm, qq = (a <= i ? l.trq_n[a] : l.trq_o[a]) - tm;
                    dv += pq * pq; tv += qq * qq;
                }
                const double ds = sqrt(dv / A), ts = sqrt(tv / A);
                float* o = p.o.info + na * GMPE_INFO_KEYS;
                o[0] = (float)rew; o[1] = (float)dleft; o[2] = (float)trq; o[3] = (float)nac; o[4] = (float)noc;
                o[5] = (float)dm; o[6] = (float)ds; o[7] = (float)(dm / (ds + 0.0001)); o[8] = (float)dtg;
                o[9] = (float)trq; o[10] = (float)tm; o[11] = (float)ts; o[12] = (float)(tm / (ts + 0.0001));
                o[13] = (float)((double)conf / c.episode_length);
                o[14] = (float)(dsp / (svsum != 0 ? svsum : 1));
                o[15] = (float)((double)sv / (sic != 0 ? sic : 1));
                o[16] = (float)gmt;
            }
        }
        if (tid == 0) {
            if (!all_done) {
                double dsp = p.s.delta_spacing[n];
                for (int a = 0; a < A; ++a) dsp += l.serr[a];
                p.s.delta_spacing[n] = dsp;
                p.s.rng_ctr[n] = ctr0 + n_new;
                p.s.current_step[n] = cur_step;
            }
            l.flags[0] = all_done;
        }
        if (ag && !all_done) {                                          // persist the stepped state
            p.s.x[na] = l.ex[tid]; p.s.y[na] = l.ey[tid]; p.s.s2[na] = l.n2[tid]; p.s.s3[na] = l.n3[tid];
            p.s.status[na] = (uint8_t)(l.s_old[tid] || l.newf[tid]);
            p.s.prev_phase[na] = prev_phase; p.s.phase_reached[na] = phase_reached; p.s.cooldown[na] = cooldown;
            p.s.goal_tracker[na] = l.gt[tid]; p.s.p_dist[na] = p_dist; p.s.time[na] = tim;
            p.s.times_required[na] = trq; p.s.dists_to_goal[na] = dtg; p.s.dist_left[na] = dleft;
            p.s.goal_reached[na] = greached; p.s.n_agent_coll[na] = nac; p.s.n_obst_coll[na] = noc;
            p.s.spacing_viol[na] = sv; p.s.steps_in_corr[na] = sic; p.s.conformance[na] = conf;
        }
        __syncthreads();
    }

    // ---- 5. reset (explicit, or the worker's auto-reset when every agent is done)
    const bool do_reset = l.flags[0] != 0;                               // block-uniform
    if (do_reset) {
        if (tid == 0) {
            int64_t ctr = p.s.rng_ctr[n];
            if (p.mode == MODE_STEP) ctr += l.flags[1];                  // this step's heading re-draws come first
            reset_world_serial(p, l, n, ctr, err);
            p.s.rng_ctr[n] = ctr;
            p.s.current_step[n] = 0;
            p.s.delta_spacing[n] = 0.0;
        }
        __syncthreads();
        if (ag) {
            const int i = tid;
            l.s2[i] = l.n2[i]; l.s3[i] = l.n3[i];
            double vx, vy; vel_of(c, l.n2[i], l.n3[i], vx, vy);
            l.vox[i] = l.vnx[i] = vx; l.voy[i] = l.vny[i] = vy;
            l.s_old[i] = 0; l.newf[i] = 0; l.gt[i] = -1;
            int prevA = prev_phase, ph = 0;
            if (july) ph = phase_eval(l.tube, l.ex[i], l.ey[i], prev_phase, prevA);   // reset-time observation (:1447)
            prev_phase = prevA;
            const double dx = l.ex[i] - l.ex[A + i], dy = l.ey[i] - l.ey[A + i];
            gmt = c.max_speed > 0 ? sqrt(dx * dx + dy * dy) / c.max_speed : 0.0;
            p.s.x[na] = l.ex[i]; p.s.y[na] = l.ey[i]; p.s.s2[na] = l.n2[i]; p.s.s3[na] = l.n3[i];
            p.s.status[na] = 0; p.s.prev_phase[na] = prev_phase; p.s.phase_reached[na] = 0; p.s.cooldown[na] = 0;
            p.s.goal_tracker[na] = -1; p.s.p_dist[na] = 0.0; p.s.time[na] = 0.0;
            p.s.times_required[na] = -1; p.s.dists_to_goal[na] = -1; p.s.dist_left[na] = -1;
            p.s.goal_reached[na] = -1; p.s.n_agent_coll[na] = 0; p.s.n_obst_coll[na] = 0;
            p.s.spacing_viol[na] = 0; p.s.steps_in_corr[na] = 0; p.s.conformance[na] = 0;
            p.s.goal_min_time[na] = gmt;
            ph1 = ph;
        }
        __syncthreads();                                                // positions of all agents final
        if (ag) write_obs(p, l, tid, l.vox[tid], l.voy[tid], ph1);
    }
    if (err) atomicOr(&p.s.error_flags[n], err);
    __syncthreads();

    // ---- 6. masked distance matrix (calculate_distances core.py:600-624 + mask …_july.py:1627-1648), fp32 in LDS
    for (int q = tid; q < E * E; q += BLOCK) {
        const int r = q / E, cc = q - r * E;
        double d = 0.0;
        if (r != cc) {
            const int a = r < cc ? r : cc, b = r < cc ? cc : r;       // upper-triangle delta, mirrored
            const double dx = l.ex[a] - l.ex[b], dy = l.ey[a] - l.ey[b];
            d = sqrt(dx * dx + dy * dy);
            bool off = false;
            if (r < A) off |= (l.s_old[r] || l.newf[r]) != 0;
            if (cc < A) off |= (l.s_old[cc] || l.newf[cc]) != 0;
            if (r >= A && r < A + L) off |= (r - A < A && l.gt[r - A] == r - A);
            if (cc >= A && cc < A + L) off |= (cc - A < A && l.gt[cc - A] == cc - A);
            if (off) d = 0.0;
        }
        l.M[q] = (float)d;
    }
    __syncthreads();

    // ---- 7. stream the observations out (16-byte stores wherever the row length allows)
    const int EE = E * E;
    if (p.o.adj) {
        if (p.o.adj_compact) {
            float* dst = p.o.adj + (size_t)n * EE;
            if ((EE & 3) == 0) for (int q = tid; q < EE / 4; q += BLOCK) reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(l.M)[q];
            else for (int q = tid; q < EE; q += BLOCK) dst[q] = l.M[q];
        } else {
            float* dst = p.o.adj + (size_t)n * A * EE;
            if ((EE & 3) == 0) {
                const int nq = EE / 4;
                for (int q = tid; q < A * nq; q += BLOCK) {
                    const int m = q % nq;                                // same E×E for every ego
                    reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(l.M)[m];
                }
            } else for (int q = tid; q < A * EE; q += BLOCK) dst[q] = l.M[q % EE];
        }
    }
    if (p.o.node_obs) {
        // node row (ego i, entity k) = 2 float4: [rel_vel, rel_pos] and [rel_goal, occupied, type]
        float4* dst = reinterpret_cast<float4*>(p.o.node_obs + (size_t)n * A * E * GMPE_NODE_FEATS);
        for (int q = tid; q < A * E * 2; q += BLOCK) {
            const int half = q & 1, row = q >> 1;
            const int i = row / E, k = row - i * E;
            const double px = l.ex[i], py = l.ey[i];
            const double rx = l.ex[k] - px, ry = l.ey[k] - py;
            float4 v;
            if (half == 0) {
                const double evx = l.newf[i] ? l.vnx[i] : l.vox[i], evy = l.newf[i] ? l.vny[i] : l.voy[i];
                double kvx = 0.0, kvy = 0.0;
                if (k < A) { const bool post = l.newf[k] && k <= i; kvx = post ? l.vnx[k] : l.vox[k]; kvy = post ? l.vny[k] : l.voy[k]; }
                v = make_float4((float)(kvx - evx), (float)(kvy - evy), (float)rx, (float)ry);
            } else {
                if (k < A) v = make_float4((float)(l.ex[A + k] - px), (float)(l.ey[A + k] - py), 0.0f, 0.0f);
                else v = make_float4((float)rx, (float)ry, 1.0f, k < A + L ? 1.0f : 2.0f);
            }
            dst[q] = v;
        }
    }
    if (p.o.obs) {
        float* dst = p.o.obs + (size_t)n * A * D;
        for (int q = tid; q < A * D; q += BLOCK) dst[q] = l.obs[q];
    }
    if (p.o.agent_id) for (int q = tid; q < A; q += BLOCK) p.o.agent_id[(size_t)n * A + q] = q;   // get_id :1554
}

// ---------------------------------------------------------------- learner-side edge set
// process_adj (onpolicy/algorithms/utils/gnn_new.py:329-358): mask = (adj < d) & (adj > 0) on fp32
// (inclusive=1 gives update_graph's `<=`, …_july.py:1660), edges in (batch,row,col) order, node ids
// offset by batch*E. Three launches: per-graph count, single-block scan, ordered per-graph compaction.
__device__ __forceinline__ bool edge_pred(float v, float d, int inclusive) {
    return (inclusive ? v <= d : v < d) && v > 0.0f;
}
__global__ __launch_bounds__(256) void k_edge_count(const float* __restrict__ adj, int EE, float d, int inclusive, int32_t* __restrict__ counts) {
    __shared__ int wsum[4];
    const float* g = adj + (size_t)blockIdx.x * EE;
    int c = 0;
    for (int q = threadIdx.x; q < EE; q += 256) c += edge_pred(g[q], d, inclusive) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(1024) void k_edge_scan(const int32_t* __restrict__ counts, int B, int32_t* __restrict__ offsets, int32_t* __restrict__ total) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int lo = t * per, hi = min(B, lo + per);
    int s = 0;
    for (int q = lo; q < hi; ++q) s += counts[q];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                    // Hillis-Steele inclusive scan
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int q = lo; q < hi; ++q) { offsets[q] = run; run += counts[q]; }
    if (t == 1023) *total = part[1023];
}
__global__ __launch_bounds__(256) void k_edge_write(const float* __restrict__ adj, int E, float d, int inclusive,
                                                    const int32_t* __restrict__ offsets, int32_t* __restrict__ edge_index,
                                                    float* __restrict__ edge_attr, int cap) {
    __shared__ int wtot[4];
    __shared__ int base_s;
    const int EE = E * E, b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* g = adj + (size_t)b * EE;
    if (t == 0) base_s = offsets[b];
    __syncthreads();
    for (int q0 = 0; q0 < EE; q0 += 256) {
        const int q = q0 + t;
        const float v = q < EE ? g[q] : 0.0f;
        const bool f = q < EE && edge_pred(v, d, inclusive);
        const unsigned long long bal = __ballot(f);
        if (lane == 0) wtot[w] = __popcll(bal);
        __syncthreads();
        int pre = __popcll(bal & ((1ull << lane) - 1ull));
        for (int k = 0; k < w; ++k) pre += wtot[k];
        const int pos = base_s + pre;
        if (f && pos < cap) {
            const int r = q / E, cc = q - r * E;
            edge_index[pos] = b * E + r;
            edge_index[cap + pos] = b * E + cc;
            edge_attr[pos] = v;
        }
        __syncthreads();
        if (t == 0) base_s += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
}

}  // namespace gmpe

// =================================================================== host side / C ABI
using namespace gmpe;

struct gmpe_handle {
    gmpe_config c;
    int device;
    DevState s;
    int A, L, O, E, D;
    std::vector<void*> allocs;
    bool timing = false;
    std::vector<hipEvent_t> ev;      // pairs
    size_t ev_used = 0;
    double t_total_ms = 0;
    int64_t t_launches = 0;
    int block = 0;
    int32_t* edge_ws = nullptr;      // [2*cap_graphs] counts | offsets
    size_t edge_ws_graphs = 0;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(GMPE_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

template <typename T>
static int dev_alloc(gmpe_handle* h, T** p, size_t n, int fill_byte) {
    const size_t bytes = (n ? n : 1) * sizeof(T);
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, bytes));
    HIPCHK(hipMemset(q, fill_byte, bytes));
    h->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return GMPE_OK;
}

extern "C" {

int gmpe_abi_version(void) { return GMPE_ABI_VERSION; }
const char* gmpe_last_error(void) { return g_err.c_str(); }
int gmpe_obs_dim(const gmpe_config* c) { return c->scenario == GMPE_SCENARIO_TUBE_JULY ? 19 : 13; }
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
    if (cfg->scenario != GMPE_SCENARIO_NAVIGATION_GRAPH && cfg->scenario != GMPE_SCENARIO_TUBE_JULY)
        return fail(GMPE_ERR_UNSUPPORTED, "unknown scenario");
    if ((cfg->scenario == GMPE_SCENARIO_TUBE_JULY) == (cfg->dynamics == GMPE_DYN_DOUBLE_INTEGRATOR))
        return fail(GMPE_ERR_UNSUPPORTED, "tube_july is kinematic; navigation_graph is double_integrator");
    if (cfg->dynamics == GMPE_DYN_DOUBLE_INTEGRATOR ? (cfg->n_actions != 5 && cfg->n_actions != 9) : cfg->n_actions != 25)
        return fail(GMPE_ERR_INVALID_ARG, "n_actions does not match the dynamics");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GMPE_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(GMPE_ERR_INVALID_ARG, "bad device index");
    HIPCHK(hipSetDevice(device));
    gmpe_handle* h = new gmpe_handle();
    h->c = *cfg; h->device = device;
    h->A = cfg->num_agents; h->L = cfg->num_landmarks; h->O = cfg->num_obstacles; h->E = E; h->D = gmpe_obs_dim(cfg);
    const size_t N = cfg->num_envs, NA = N * h->A;
    DevState& s = h->s;
    int rc = 0;
#define AL(p, n, fill) if ((rc = dev_alloc(h, &(p), (n), (fill)))) { gmpe_destroy(h); return rc; }
    AL(s.x, NA, 0) AL(s.y, NA, 0) AL(s.s2, NA, 0) AL(s.s3, NA, 0) AL(s.p_dist, NA, 0) AL(s.time, NA, 0)
    AL(s.status, NA, 0) AL(s.prev_phase, NA, 0) AL(s.phase_reached, NA, 0) AL(s.cooldown, NA, 0)
    AL(s.goal_tracker, NA, 0xFF) AL(s.current_step, N, 0) AL(s.rng_ctr, N, 0)
    AL(s.tube, N * GMPE_TUBE_STRIDE, 0) AL(s.landmarks, N * h->L * 2, 0) AL(s.obstacles, N * h->O * 2, 0)
    AL(s.times_required, NA, 0xFF) AL(s.dists_to_goal, NA, 0xFF) AL(s.dist_left, NA, 0xFF) AL(s.goal_reached, NA, 0xFF)
    AL(s.n_agent_coll, NA, 0) AL(s.n_obst_coll, NA, 0) AL(s.spacing_viol, NA, 0) AL(s.steps_in_corr, NA, 0)
    AL(s.conformance, NA, 0) AL(s.goal_min_time, NA, 0) AL(s.delta_spacing, N, 0) AL(s.error_flags, N, 0)
#undef AL
    s.tape = nullptr; s.tape_len = 0;
    // workgroup size: one wave for small graphs, up to 4 waves when E*E output rows dominate
    const char* env_block = getenv("GMPE_BLOCK");
    h->block = env_block ? atoi(env_block) : (E <= 24 ? 64 : (E <= 48 ? 128 : 256));
    if (h->block != 64 && h->block != 128 && h->block != 256) h->block = 256;
    const size_t lds = lds_bytes(h->A, E, h->D);
    if (lds > 160 * 1024) { gmpe_destroy(h); return fail(GMPE_ERR_UNSUPPORTED, "per-env LDS tile exceeds 160 KiB"); }
    *out = h;
    return GMPE_OK;
}

int gmpe_destroy(gmpe_handle* h) {
    if (!h) return GMPE_OK;
    (void)hipSetDevice(h->device);
    for (void* q : h->allocs) (void)hipFree(q);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->edge_ws) (void)hipFree(h->edge_ws);
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
    p.A = h->A; p.L = h->L; p.O = h->O; p.E = h->E; p.D = h->D;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = lds_bytes(h->A, h->E, h->D);
    const dim3 grid(h->c.num_envs);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        if (h->ev_used + 2 > h->ev.size()) {
            for (int q = 0; q < 2; ++q) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->ev.push_back(e); }
        }
        e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1];
        HIPCHK(hipEventRecord(e0, st));
    }
    switch (h->block) {
        case 64: hipLaunchKernelGGL(k_env<64>, grid, dim3(64), lds, st, p); break;
        case 128: hipLaunchKernelGGL(k_env<128>, grid, dim3(128), lds, st, p); break;
        default: hipLaunchKernelGGL(k_env<256>, grid, dim3(256), lds, st, p); break;
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
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->edge_ws), sizeof(int32_t) * 2 * (size_t)batch));
        h->edge_ws_graphs = batch;
    }
    int32_t* counts = h->edge_ws; int32_t* offsets = h->edge_ws + h->edge_ws_graphs;
    const int EE = num_nodes * num_nodes;
    hipLaunchKernelGGL(k_edge_count, dim3(batch), dim3(256), 0, st, adj_dev, EE, max_edge_dist, inclusive, counts);
    hipLaunchKernelGGL(k_edge_scan, dim3(1), dim3(1024), 0, st, counts, batch, offsets, n_edges_dev);
    hipLaunchKernelGGL(k_edge_write, dim3(batch), dim3(256), 0, st, adj_dev, num_nodes, max_edge_dist, inclusive, offsets,
                       edge_index_dev, edge_attr_dev, cap);
    HIPCHK(hipGetLastError());
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
