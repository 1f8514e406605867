// gmpe_device.h — device-side helpers of the gfx950 GraphMPE step engine.
// Compiled with -ffp-contract=off: the reference (NumPy fp64) never fuses a*b+c, and the adjacency
// thresholds / phase decisions must see the same roundings.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gmpe.h"

#define GMPE_MAX_TRIES 4096

namespace gmpe {

// ---------------------------------------------------------------- Philox4x32-10
// Same stream as oracle/gmpe_oracle.c: counter = (k_lo, k_hi, env_id, 'GMPE'), key = seed.
__device__ __forceinline__ double philox_uniform(uint64_t seed, uint32_t env_id, uint64_t k) {
    uint32_t c0 = (uint32_t)k, c1 = (uint32_t)(k >> 32), c2 = env_id, c3 = 0x474D5045u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint64_t bits = ((uint64_t)c1 << 32) | c0;
    return (double)(bits >> 11) * 0x1.0p-53;
}

// Persistent per-env state in HBM, struct-of-arrays ([N,A] unless noted) — include/gmpe.h gmpe_field.
struct DevState {
    double *x, *y, *s2, *s3, *p_dist, *time;
    uint8_t* status;
    int32_t *prev_phase, *phase_reached, *cooldown, *goal_tracker;
    int32_t* current_step;   // [N]
    int64_t* rng_ctr;        // [N]
    double *tube;            // [N,12]
    double *landmarks;       // [N,L,2]
    double *obstacles;       // [N,O,2]
    int32_t *times_required, *dists_to_goal, *dist_left, *goal_reached, *n_agent_coll, *n_obst_coll,
        *spacing_viol, *steps_in_corr, *conformance;
    double *goal_min_time;
    double *prev_proj;       // [N,A] rot_inv: float32 values held in f64
    double *delta_spacing;   // [N]
    int32_t* error_flags;    // [N]
    const double* tape;      // [N,tape_len] or nullptr
    int64_t tape_len;
};

enum { T_ANGLE = 0, T_ENTX, T_ENTY, T_EXX, T_EXY, T_EX, T_EY, T_NX, T_NY, T_L, T_HALFW, T_WIDTH };

// One uniform draw of env n's stream (tape in parity mode, Philox otherwise). `ctr` is the env's
// running draw counter held by the calling lane; `err` accumulates sticky error bits.
__device__ __forceinline__ double draw_at(const gmpe_config& c, const DevState& s, int n, int64_t k, int& err) {
    if (s.tape) {
        if (k >= s.tape_len) { err |= 1; return 0.5; }
        return s.tape[(size_t)n * s.tape_len + k];
    }
    return philox_uniform(c.seed, (uint32_t)(c.env_id_base + n), (uint64_t)k);
}

__device__ __forceinline__ double norm2(double dx, double dy) { return sqrt(dx * dx + dy * dy); }
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// np.logaddexp(0, x)
__device__ __forceinline__ double logaddexp0(double x) {
    if (x == 0.0) return 0.6931471805599453094;
    const double tmp = 0.0 - x;
    if (tmp > 0) return log1p(exp(-tmp));
    return x + log1p(exp(tmp));
}

// get_agent_phase (…_july.py:683-733) as a pure function of (pos, previous_phase): returns the phase
// and writes the (possibly mutated) previous_phase. tube = the 12-double record.
__device__ __forceinline__ int phase_eval(const double* tube, double px, double py, int prev, int& prev_out) {
    const double eps = 0.05, L = tube[T_L], hw = tube[T_HALFW];
    const double rx = (double)(float)px - tube[T_ENTX], ry = (double)(float)py - tube[T_ENTY];   // :624
    const double s = rx * tube[T_EX] + ry * tube[T_EY];
    const double yy = rx * tube[T_NX] + ry * tube[T_NY];
    const bool in_tube = (-eps <= s && s <= L + eps) && (fabs(yy) <= hw + eps);
    const double tdx = tube[T_EXX] - tube[T_ENTX], tdy = tube[T_EXY] - tube[T_ENTY];
    const double tn = sqrt(tdx * tdx + tdy * tdy);
    const double ux = tdx / tn, uy = tdy / tn;
    const bool passed = ((px - tube[T_EXX]) * ux + (py - tube[T_EXY]) * uy) > 0;
    const double gate_front = 0.08 * L, gate_back = 0.02 * L;
    const bool valid = (-gate_back - eps <= s && s <= gate_front + eps) && (fabs(yy) <= hw + eps);
    prev_out = prev;
    if (!in_tube && !passed) return 0;
    if (in_tube) return prev == 0 ? (valid ? 1 : 0) : 1;
    if (prev == 1) { prev_out = 2; return 2; }       // passed is implied here
    if (prev == 2) return 2;
    return 0;
}

// ---- nav_graph_metered_single_corridor_rot_inv.py geometry (:627-669) and phase FSM (:675-739)
__device__ __forceinline__ void tube_sy(const double* tube, double px, double py, double& s, double& yy) {
    const double rx = (double)(float)px - tube[T_ENTX], ry = (double)(float)py - tube[T_ENTY];   // pos rounded to fp32 first
    s = rx * tube[T_EX] + ry * tube[T_EY];
    yy = rx * tube[T_NX] + ry * tube[T_NY];
}
__device__ __forceinline__ bool in_tube_rect(double s, double y, double L, double hw) { return (-0.05 <= s && s <= L + 0.05) && (fabs(y) <= hw + 0.05); }
__device__ __forceinline__ bool in_entrance_gate(double s, double y, double L, double hw) {
    return (-(0.02 * L) - 0.05 <= s && s <= 0.08 * L + 0.05) && (fabs(y) <= hw + 0.05);
}
// exit_back_ratio: 0.05 in rot_inv (:619), 0.02 in two_phase_graph.py:589 / three_phase_graph.py
__device__ __forceinline__ bool in_exit_gate(double s, double y, double L, double hw, double back_ratio) {
    return (L - back_ratio * L - 0.05 <= s && s <= L + 0.08 * L + 0.05) && (fabs(y) <= hw + 0.05);
}
__device__ __forceinline__ double entrance_gate_distance(double s, double y, double hw) { return hypot(fabs(s), y - clipd(y, -hw, hw)); }
__device__ __forceinline__ double exit_gate_distance(double s, double y, double L, double hw) {
    const double ds = (L - s) > 0.0 ? (L - s) : 0.0;
    return hypot(ds, y - clipd(y, -hw, hw));
}
// pure function of (pos, previous_phase, phase_reached); the caller decrements the cooldown (:700-702)
// VARIANT 0: rot_inv; 1: two_phase_graph.py:660-701 (exit gate reached from inside the tube); 2: three_phase_graph.py:656-699
template <int VARIANT>
__device__ __forceinline__ int phase_eval_rot(const double* tube, double px, double py, int prev, int phase_reached) {
    const double L = tube[T_L], hw = tube[T_HALFW];
    double s, yy; tube_sy(tube, px, py, s, yy);
    const bool in_tube = in_tube_rect(s, yy, L, hw), passed = s > L;
    const bool valid_exit = in_exit_gate(s, yy, L, hw, VARIANT ? 0.02 : 0.05);
    if (!in_tube && !passed) return 0;
    if (in_tube) {
        if (prev == 0) return in_entrance_gate(s, yy, L, hw) ? 1 : 0;
        if (VARIANT >= 1 && prev == 1 && valid_exit) return 2;
        if (VARIANT == 2 && prev == 2 && valid_exit) return 2;
        return 1;
    }
    if (phase_reached >= 1) {
        if (prev == 1 && valid_exit) return 2;
        if (prev == 2) return 2;
    }
    return 0;
}
// (theta - arctan2(e_y, e_x) + pi) % 2pi - pi with NumPy's float modulo (sign of the divisor) — two_phase_graph.py:1060-1063
__device__ __forceinline__ double heading_error_signed(const double* tube, double th) {
    const double ch = atan2(tube[T_EY], tube[T_EX]);
    const double a = th - ch + M_PI, b = 2 * M_PI;
    double m = fmod(a, b);
    if (m != 0.0) { if (m < 0) m += b; } else m = 0.0;
    return m - M_PI;
}
// get_rotated_position_from_relative (:91-97): [[c, s], [-s, c]] @ v
__device__ __forceinline__ void rot2(double c, double s, double vx, double vy, double& ox, double& oy) { ox = c * vx + s * vy; oy = -s * vx + c * vy; }

// get_wall_collision_force (core.py:909-964); returns false for None.
__device__ __forceinline__ bool wall_force(const gmpe_wall& wl, double px, double py, double size, double kf,
                                           double km, double& fx, double& fy) {
    const bool horiz = wl.orient == 0;
    const double prll = horiz ? px : py, perp = horiz ? py : px;
    double theta, dist_min;
    if (prll < wl.end0 - size || prll > wl.end1 + size) return false;
    if (prll < wl.end0 || prll > wl.end1) {
        const double past = prll < wl.end0 ? prll - wl.end0 : prll - wl.end1;
        theta = asin(past / size);
        dist_min = cos(theta) * size + 0.5 * wl.width;
    } else { theta = 0; dist_min = size + 0.5 * wl.width; }
    const double delta = perp - wl.axis_pos;
    const double dist = fabs(delta);
    const double pen = logaddexp0(-(dist - dist_min) / km) * km;
    const double fm = kf * delta / dist * pen;
    const double f_perp = cos(theta) * fm, f_prll = sin(theta) * fabs(fm);
    fx = horiz ? f_prll : f_perp;
    fy = horiz ? f_perp : f_prll;
    return true;
}

}  // namespace gmpe
