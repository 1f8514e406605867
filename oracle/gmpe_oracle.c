/* gmpe_oracle.c — CPU restatement of the reference's GraphMPE step path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the CHECKER (or as the timed CPU "port" baseline). The product path (libgmpe.so) never
 * links or calls it.
 *
 * Parity status: PINNED. tests/test_oracle_golden.py checks this file against vectors captured by
 * running the reference itself in the build container (tests/golden/make_fixtures.py):
 * end-to-end rollouts of MultiAgentGraphEnv on the July tube scenario and on the three shipped-weight
 * scenarios (nav_graph_metered_single_corridor_rot_inv, two_phase_graph, three_phase_graph: 6 rollouts each)
 * incl. auto-resets and the np.random draw order, scipy-RK45 single steps, and both force-path variants.
 * `navigation_graph` has no scenario file in the reference: its blocks are
 * pinned individually (force path, graph, obs slice, reward blocks) but their composition is this
 * project's own — "end-to-end parity unpinned" for that scenario.
 *
 * Style: deliberately literal and SEQUENTIAL — one env at a time, agents in index order, state
 * mutated exactly where the reference mutates it (multiagent/environment.py:1036-1053). The HIP
 * engine uses a parallel two-pass restatement instead; agreement between the two is the test.
 *
 * Compile with -O2 -ffp-contract=off (no FMA contraction: NumPy evaluates a*b+c*d unfused).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gmpe.h"

#define MAX_TRIES 4096

typedef struct gmpo {
    gmpe_config c;
    int N, A, L, O, E, D, F;
    double *x, *y, *s2, *s3, *p_dist, *time, *prev_proj;
    uint8_t* status;
    int32_t *prev_phase, *phase_reached, *cooldown, *goal_tracker;
    int32_t* current_step;
    int64_t* rng_ctr;
    double *tube, *landmarks, *obstacles;
    int32_t *times_required, *dists_to_goal, *dist_left, *goal_reached, *n_agent_coll, *n_obst_coll,
        *spacing_viol, *steps_in_corr, *conformance;
    double *goal_min_time, *delta_spacing;
    int32_t* error_flags;
    double* dist;                 /* [N,E,E] world.cached_dist_mag incl. in-place masking */
    double* etab;                 /* [N,W] entity table of the last step / reset (include/gmpe.h gmpe_outputs.entity_table) */
    int W;
    const double* tape;
    const double* ovr_ctrl;   /* safety-filter hook slot (multiagent/core.py:692-736): [N,A,2] filtered controls, or NULL */
    const uint8_t* ovr_use;   /* [N,A] `filtered` flags, NULL = everywhere */
    int64_t tape_len;
} gmpo;

static char g_err[256] = "";
const char* gmpo_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ RNG */
/* Philox4x32-10 (Salmon et al., SC'11), counter = (k_lo, k_hi, env_id, 'GMPE'), key = seed. */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
double gmpo_philox_uniform(uint64_t seed, uint32_t env_id, uint64_t k) {
    uint32_t c[4] = {(uint32_t)k, (uint32_t)(k >> 32), env_id, 0x474D5045u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, key);
        key[0] += 0x9E3779B9u;
        key[1] += 0xBB67AE85u;
    }
    const uint64_t bits = ((uint64_t)c[1] << 32) | c[0];
    return (double)(bits >> 11) * 0x1.0p-53;
}
static double draw(gmpo* h, int n) {
    const int64_t k = h->rng_ctr[n]++;
    if (h->tape) {
        if (k >= h->tape_len) { h->error_flags[n] |= 1; return 0.5; }
        return h->tape[(size_t)n * h->tape_len + k];
    }
    return gmpo_philox_uniform(h->c.seed, (uint32_t)(h->c.env_id_base + n), (uint64_t)k);
}
/* legacy RandomState.uniform: low + (high-low)*u */
static double uniform(gmpo* h, int n, double lo, double hi) { return lo + (hi - lo) * draw(h, n); }

/* ------------------------------------------------------------------ small helpers */
static inline double norm2(double dx, double dy) { return sqrt(dx * dx + dy * dy); }
static inline double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* np.logaddexp(0, x) (numpy/core/src/umath/loops_utils / npy_math: npy_logaddexp) */
static inline double logaddexp0(double x) {
    if (x == 0.0) return 0.0 + 0.6931471805599453094; /* x1 == x2 */
    const double tmp = 0.0 - x;
    if (tmp > 0) return 0.0 + log1p(exp(-tmp));
    if (tmp <= 0) return x + log1p(exp(tmp));
    return tmp; /* NaN */
}
static inline int is_kinematic(const gmpo* h) { return h->c.dynamics != GMPE_DYN_DOUBLE_INTEGRATOR; }

static inline int is_rotfam(const gmpe_config* c) { return c->scenario >= GMPE_SCENARIO_ROT_INV; }      /* rot_inv, two_phase, three_phase */
static inline int is_phasefam(const gmpe_config* c) { return c->scenario == GMPE_SCENARIO_TWO_PHASE || c->scenario == GMPE_SCENARIO_THREE_PHASE; }
int gmpo_obs_dim(const gmpe_config* c) { return c->scenario == GMPE_SCENARIO_TUBE_JULY ? 19 : (is_phasefam(c) ? 15 : 13); }
int gmpo_node_feats(const gmpe_config* c) { return (is_rotfam(c) || c->graph_feat_type == 1) ? 7 : 8; }
static inline int is_tube(const gmpe_config* c) { return c->scenario != GMPE_SCENARIO_NAVIGATION_GRAPH; }
int gmpo_num_entities(const gmpe_config* c) { return c->num_agents + c->num_landmarks + c->num_obstacles; }
/* width of the entity table (include/gmpe.h gmpe_outputs.entity_table): x[E], y[E], vox / voy / vnx / vny [A] (+ cos / sin [A] rot_inv family) (+ exit x, y two_phase) */
int gmpo_entity_table_width(const gmpe_config* c) {
    return 2 * gmpo_num_entities(c) + 4 * c->num_agents + (is_rotfam(c) ? 2 * c->num_agents : 0) + (c->scenario == GMPE_SCENARIO_TWO_PHASE ? 2 : 0) +
           (gmpo_num_entities(c) + 31) / 32;
}

/* ------------------------------------------------------------------ create / fields */
#define ALLOC(p, n, T) do { (p) = (T*)calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(T)); if (!(p)) return GMPE_ERR_INVALID_ARG; } while (0)

int gmpo_create(const gmpe_config* cfg, gmpo** out) {
    if (!cfg || !out || cfg->abi_version != GMPE_ABI_VERSION) { snprintf(g_err, sizeof g_err, "bad config/abi"); return GMPE_ERR_INVALID_ARG; }
    if (cfg->formation_type < GMPE_FORMATION_POINT || cfg->formation_type > GMPE_FORMATION_CIRCLE) { snprintf(g_err, sizeof g_err, "bad formation_type"); return GMPE_ERR_UNSUPPORTED; }
    if (cfg->num_agents < 1 || cfg->num_agents > GMPE_MAX_AGENTS || cfg->num_landmarks < cfg->num_agents ||
        cfg->num_envs < 1 || cfg->num_walls > GMPE_MAX_WALLS || gmpo_num_entities(cfg) > GMPE_MAX_ENTITIES) {
        snprintf(g_err, sizeof g_err, "config out of range"); return GMPE_ERR_INVALID_ARG;
    }
    gmpo* h = (gmpo*)calloc(1, sizeof(gmpo));
    h->c = *cfg;
    h->N = cfg->num_envs; h->A = cfg->num_agents; h->L = cfg->num_landmarks; h->O = cfg->num_obstacles;
    h->E = gmpo_num_entities(cfg); h->D = gmpo_obs_dim(cfg); h->F = gmpo_node_feats(cfg);
    const size_t NA = (size_t)h->N * h->A;
    ALLOC(h->x, NA, double); ALLOC(h->y, NA, double); ALLOC(h->s2, NA, double); ALLOC(h->s3, NA, double);
    ALLOC(h->p_dist, NA, double); ALLOC(h->time, NA, double); ALLOC(h->prev_proj, NA, double); ALLOC(h->status, NA, uint8_t);
    ALLOC(h->prev_phase, NA, int32_t); ALLOC(h->phase_reached, NA, int32_t); ALLOC(h->cooldown, NA, int32_t);
    ALLOC(h->goal_tracker, NA, int32_t); ALLOC(h->current_step, h->N, int32_t); ALLOC(h->rng_ctr, h->N, int64_t);
    ALLOC(h->tube, (size_t)h->N * GMPE_TUBE_STRIDE, double); ALLOC(h->landmarks, (size_t)h->N * h->L * 2, double);
    ALLOC(h->obstacles, (size_t)h->N * h->O * 2, double);
    ALLOC(h->times_required, NA, int32_t); ALLOC(h->dists_to_goal, NA, int32_t); ALLOC(h->dist_left, NA, int32_t);
    ALLOC(h->goal_reached, NA, int32_t); ALLOC(h->n_agent_coll, NA, int32_t); ALLOC(h->n_obst_coll, NA, int32_t);
    ALLOC(h->spacing_viol, NA, int32_t); ALLOC(h->steps_in_corr, NA, int32_t); ALLOC(h->conformance, NA, int32_t);
    ALLOC(h->goal_min_time, NA, double); ALLOC(h->delta_spacing, h->N, double); ALLOC(h->error_flags, h->N, int32_t);
    ALLOC(h->dist, (size_t)h->N * h->E * h->E, double);
    h->W = gmpo_entity_table_width(cfg);
    ALLOC(h->etab, (size_t)h->N * h->W, double);
    for (size_t i = 0; i < NA; ++i) { h->goal_tracker[i] = -1; h->times_required[i] = -1; h->dists_to_goal[i] = -1; h->dist_left[i] = -1; h->goal_reached[i] = -1; }
    *out = h;
    return GMPE_OK;
}
int gmpo_destroy(gmpo* h) {
    if (!h) return GMPE_OK;
    void* ps[] = {h->prev_proj, h->x, h->y, h->s2, h->s3, h->p_dist, h->time, h->status, h->prev_phase, h->phase_reached,
                  h->cooldown, h->goal_tracker, h->current_step, h->rng_ctr, h->tube, h->landmarks, h->obstacles,
                  h->times_required, h->dists_to_goal, h->dist_left, h->goal_reached, h->n_agent_coll,
                  h->n_obst_coll, h->spacing_viol, h->steps_in_corr, h->conformance, h->goal_min_time,
                  h->delta_spacing, h->error_flags, h->dist, h->etab};
    for (size_t i = 0; i < sizeof ps / sizeof ps[0]; ++i) free(ps[i]);
    free(h);
    return GMPE_OK;
}
static int field_ptr(gmpo* h, int f, void** p, size_t* bytes) {
    const size_t N = h->N, NA = (size_t)h->N * h->A;
    switch (f) {
        case GMPE_F_X: *p = h->x; *bytes = NA * 8; break;
        case GMPE_F_Y: *p = h->y; *bytes = NA * 8; break;
        case GMPE_F_S2: *p = h->s2; *bytes = NA * 8; break;
        case GMPE_F_S3: *p = h->s3; *bytes = NA * 8; break;
        case GMPE_F_P_DIST: *p = h->p_dist; *bytes = NA * 8; break;
        case GMPE_F_TIME: *p = h->time; *bytes = NA * 8; break;
        case GMPE_F_STATUS: *p = h->status; *bytes = NA; break;
        case GMPE_F_PREV_PHASE: *p = h->prev_phase; *bytes = NA * 4; break;
        case GMPE_F_PHASE_REACHED: *p = h->phase_reached; *bytes = NA * 4; break;
        case GMPE_F_COOLDOWN: *p = h->cooldown; *bytes = NA * 4; break;
        case GMPE_F_GOAL_TRACKER: *p = h->goal_tracker; *bytes = NA * 4; break;
        case GMPE_F_CURRENT_STEP: *p = h->current_step; *bytes = N * 4; break;
        case GMPE_F_RNG_CTR: *p = h->rng_ctr; *bytes = N * 8; break;
        case GMPE_F_TUBE: *p = h->tube; *bytes = N * GMPE_TUBE_STRIDE * 8; break;
        case GMPE_F_LANDMARKS: *p = h->landmarks; *bytes = N * h->L * 2 * 8; break;
        case GMPE_F_OBSTACLES: *p = h->obstacles; *bytes = N * h->O * 2 * 8; break;
        case GMPE_F_TIMES_REQUIRED: *p = h->times_required; *bytes = NA * 4; break;
        case GMPE_F_DISTS_TO_GOAL: *p = h->dists_to_goal; *bytes = NA * 4; break;
        case GMPE_F_DIST_LEFT: *p = h->dist_left; *bytes = NA * 4; break;
        case GMPE_F_GOAL_REACHED: *p = h->goal_reached; *bytes = NA * 4; break;
        case GMPE_F_N_AGENT_COLL: *p = h->n_agent_coll; *bytes = NA * 4; break;
        case GMPE_F_N_OBST_COLL: *p = h->n_obst_coll; *bytes = NA * 4; break;
        case GMPE_F_SPACING_VIOL: *p = h->spacing_viol; *bytes = NA * 4; break;
        case GMPE_F_STEPS_IN_CORR: *p = h->steps_in_corr; *bytes = NA * 4; break;
        case GMPE_F_CONFORMANCE: *p = h->conformance; *bytes = NA * 4; break;
        case GMPE_F_GOAL_MIN_TIME: *p = h->goal_min_time; *bytes = NA * 8; break;
        case GMPE_F_DELTA_SPACING: *p = h->delta_spacing; *bytes = N * 8; break;
        case GMPE_F_ERROR_FLAGS: *p = h->error_flags; *bytes = N * 4; break;
        case GMPE_F_PREV_PROJ: *p = h->prev_proj; *bytes = NA * 8; break;
        default: snprintf(g_err, sizeof g_err, "unknown field %d", f); return GMPE_ERR_INVALID_ARG;
    }
    return GMPE_OK;
}
int gmpo_field_bytes(gmpo* h, int f, size_t* bytes) { void* p; return field_ptr(h, f, &p, bytes); }
int gmpo_get_field(gmpo* h, int f, void* dst, size_t bytes) {
    void* p; size_t b; int rc = field_ptr(h, f, &p, &b); if (rc) return rc;
    if (b != bytes) { snprintf(g_err, sizeof g_err, "field %d: %zu bytes expected, got %zu", f, b, bytes); return GMPE_ERR_INVALID_ARG; }
    memcpy(dst, p, b); return GMPE_OK;
}
int gmpo_set_field(gmpo* h, int f, const void* src, size_t bytes) {
    void* p; size_t b; int rc = field_ptr(h, f, &p, &b); if (rc) return rc;
    if (b != bytes) { snprintf(g_err, sizeof g_err, "field %d: %zu bytes expected, got %zu", f, b, bytes); return GMPE_ERR_INVALID_ARG; }
    memcpy(p, src, b); return GMPE_OK;
}
int gmpo_set_rng_tape(gmpo* h, const double* tape, int64_t len_per_env) { h->tape = tape; h->tape_len = len_per_env; return GMPE_OK; }
/* World.step integrates safe_action_list instead of raw_action_list where the filter intervened (core.py:692-736) */
int gmpo_set_control_override(gmpo* h, const double* ctrl, const uint8_t* use) { h->ovr_ctrl = ctrl; h->ovr_use = ctrl ? use : NULL; return GMPE_OK; }
static void filtered_control(const gmpo* h, int n, int i, double* u) {
    if (!h->ovr_ctrl) return;
    const size_t na = (size_t)n * h->c.num_agents + i;
    if (h->ovr_use && !h->ovr_use[na]) return;
    u[0] = h->ovr_ctrl[2 * na]; u[1] = h->ovr_ctrl[2 * na + 1];
}
/* entity table of the last step / reset, [N,W] (what a rank ships instead of node_obs; tests compare the engine's table and feed the gloo rehearsal) */
int gmpo_get_entity_table(gmpo* h, double* dst) { memcpy(dst, h->etab, (size_t)h->N * h->W * 8); return GMPE_OK; }
/* world.cached_dist_mag after in-place masking, [N,E,E] */
int gmpo_get_dist_cache(gmpo* h, double* dst) { memcpy(dst, h->dist, (size_t)h->N * h->E * h->E * 8); return GMPE_OK; }

/* ------------------------------------------------------------------ per-env view */
typedef struct envv {
    gmpo* h; int n, A, L, O, E;
    double *x, *y, *s2, *s3, *p_dist, *time, *prev_proj; uint8_t* status;
    int32_t *prev_phase, *phase_reached, *cooldown, *goal_tracker;
    double *tube, *lm, *ob, *dist;
    int32_t *times_required, *dists_to_goal, *dist_left, *goal_reached, *n_agent_coll, *n_obst_coll,
        *spacing_viol, *steps_in_corr, *conformance;
    double* goal_min_time;
} envv;
static envv view(gmpo* h, int n) {
    envv v; const size_t o = (size_t)n * h->A;
    v.h = h; v.n = n; v.A = h->A; v.L = h->L; v.O = h->O; v.E = h->E;
    v.x = h->x + o; v.y = h->y + o; v.s2 = h->s2 + o; v.s3 = h->s3 + o; v.p_dist = h->p_dist + o; v.time = h->time + o; v.prev_proj = h->prev_proj + o;
    v.status = h->status + o; v.prev_phase = h->prev_phase + o; v.phase_reached = h->phase_reached + o;
    v.cooldown = h->cooldown + o; v.goal_tracker = h->goal_tracker + o;
    v.tube = h->tube + (size_t)n * GMPE_TUBE_STRIDE; v.lm = h->landmarks + (size_t)n * h->L * 2;
    v.ob = h->obstacles + (size_t)n * h->O * 2; v.dist = h->dist + (size_t)n * h->E * h->E;
    v.times_required = h->times_required + o; v.dists_to_goal = h->dists_to_goal + o; v.dist_left = h->dist_left + o;
    v.goal_reached = h->goal_reached + o; v.n_agent_coll = h->n_agent_coll + o; v.n_obst_coll = h->n_obst_coll + o;
    v.spacing_viol = h->spacing_viol + o; v.steps_in_corr = h->steps_in_corr + o; v.conformance = h->conformance + o;
    v.goal_min_time = h->goal_min_time + o;
    return v;
}
/* tube record layout (GMPE_F_TUBE) */
enum { T_ANGLE = 0, T_ENTX, T_ENTY, T_EXX, T_EXY, T_EX, T_EY, T_NX, T_NY, T_L, T_HALFW, T_WIDTH };

/* entity k position / velocity (entity order: agents, landmarks, obstacles — core.py:575-582) */
static void ent_pos(const envv* v, int k, double* px, double* py) {
    if (k < v->A) { *px = v->x[k]; *py = v->y[k]; }
    else if (k < v->A + v->L) { *px = v->lm[2 * (k - v->A)]; *py = v->lm[2 * (k - v->A) + 1]; }
    else { *px = v->ob[2 * (k - v->A - v->L)]; *py = v->ob[2 * (k - v->A - v->L) + 1]; }
}
/* p_vel: air_taxi core.py:281-286 (speed*cos, speed*sin); double integrator core.py:191-193 */
static void agent_vel(const envv* v, int k, double* vx, double* vy) {
    if (is_kinematic(v->h)) { *vx = v->s3[k] * cos(v->s2[k]); *vy = v->s3[k] * sin(v->s2[k]); }
    else { *vx = v->s2[k]; *vy = v->s3[k]; }
}

/* Entity table rows of env v (layout: include/gmpe.h). stage 0: positions + the agents' velocities BEFORE the step's reward loop (the value an ego sees for an
 * agent with a larger index, environment.py:1036-1053); stage 1: the velocities AFTER it (reset_velocity on a goal reach, core.py:324-333), the post-reward heading's
 * cos / sin and two_phase's exit. A reset calls both stages back to back (nothing moves between them). */
static void table_stage(envv* v, int stage) {
    gmpo* h = v->h; const int A = v->A, E = v->E;
    double* T = h->etab + (size_t)v->n * h->W;
    double* vo = T + 2 * E; double* vn = vo + 2 * A;
    if (stage == 0) {
        for (int k = 0; k < E; ++k) ent_pos(v, k, &T[k], &T[E + k]);
        for (int a = 0; a < A; ++a) agent_vel(v, a, &vo[a], &vo[A + a]);
    } else {
        for (int a = 0; a < A; ++a) agent_vel(v, a, &vn[a], &vn[A + a]);
        if (is_rotfam(&h->c)) for (int a = 0; a < A; ++a) { vn[2 * A + a] = cos(v->s2[a]); vn[3 * A + a] = sin(v->s2[a]); }
        const int nmw = (E + 31) / 32;
        if (h->c.scenario == GMPE_SCENARIO_TWO_PHASE) { T[h->W - nmw - 2] = v->tube[T_EXX]; T[h->W - nmw - 1] = v->tube[T_EXY]; }
        /* adjacency mask of this step (graph_observation's `mask:` block above: done agents, reached landmarks), 32 node bits per double */
        for (int j = 0; j < nmw; ++j) {
            uint32_t bits = 0;
            for (int b = 0; b < 32 && 32 * j + b < E; ++b) {
                const int k = 32 * j + b;
                int off = 0;
                if (k < A) off = v->status[k];
                else if (k < A + v->L) { for (int a = 0; a < A; ++a) if (v->goal_tracker[a] == k - A) off = 1; }
                bits |= (uint32_t)(off ? 1 : 0) << b;
            }
            T[h->W - nmw + j] = (double)bits;
        }
    }
}

/* World.calculate_distances, core.py:600-624: upper triangle delta, mirrored negated; norm(axis=2). */
static void calculate_distances(envv* v) {
    const int E = v->E;
    for (int a = 0; a < E; ++a) {
        double ax, ay; ent_pos(v, a, &ax, &ay);
        v->dist[a * E + a] = 0.0;
        for (int b = a + 1; b < E; ++b) {
            double bx, by; ent_pos(v, b, &bx, &by);
            const double dx = ax - bx, dy = ay - by;
            const double d = sqrt(dx * dx + dy * dy);
            v->dist[a * E + b] = d; v->dist[b * E + a] = d;
        }
    }
}

/* Scenario.is_obstacle_collision, …_july.py:864-890 */
static int is_obstacle_collision(const envv* v, double px, double py, double size) {
    const gmpe_config* c = &v->h->c;
    for (int o = 0; o < v->O; ++o) {
        const double dx = v->ob[2 * o] - px, dy = v->ob[2 * o + 1] - py;
        if (norm2(dx, dy) < 2.0 * (c->entity_size + size)) return 1;
    }
    for (int w = 0; w < c->num_walls; ++w) {
        const gmpe_wall* wl = &c->walls[w];
        const double band = 1.5 * size;
        const double perp = wl->orient == 0 ? py : px, prll = wl->orient == 0 ? px : py;
        if (wl->axis_pos - band <= perp && perp <= wl->axis_pos + band)
            if (wl->end0 - band <= prll && prll <= wl->end1 + band) return 1;
    }
    return 0;
}
/* Scenario.is_collision, …_july.py:907-914 */
static int is_collision(const envv* v, int a1, int a2) {
    if (v->status[a1] || v->status[a2]) return 0;
    return norm2(v->x[a1] - v->x[a2], v->y[a1] - v->y[a2]) < v->h->c.sep_dist;
}

/* ------------------------------------------------------------------ dynamics */
/* AirTaxiXYState.update_state, core.py:300-316. The reference integrates dstate (289-297) with
 * scipy RK45; the exact solution for constant (w, a) is used here (<=3.1e-9 from RK45 per step,
 * pinned by tests/golden/rk45_airtaxi.npz). Position uses the UNclamped v(t); clamp afterwards. */
void gmpo_kinematic_step(double s[4], double w, double a, double dt, double v_min, double v_max, double* p_dist, double* tm) {
    const double th0 = s[2], v0 = s[3];
    const double th1 = th0 + w * dt, v1 = v0 + a * dt;
    if (w != 0.0) {
        const double s0 = sin(th0), c0 = cos(th0), s1 = sin(th1), c1 = cos(th1);
        s[0] += (v1 * s1 - v0 * s0) / w + a * (c1 - c0) / (w * w);
        s[1] += (-v1 * c1 + v0 * c0) / w + a * (s1 - s0) / (w * w);
    } else {
        const double d = (v0 + 0.5 * a * dt) * dt;
        s[0] += d * cos(th0); s[1] += d * sin(th0);
    }
    s[2] = th1;
    double v = v1;
    if (v > v_max) v = v_max;
    if (v < v_min) v = v_min;
    s[3] = v;
    if (p_dist) *p_dist += v * dt;
    if (tm) *tm += dt;
}

/* get_wall_collision_force, core.py:909-964 (classic twin mpe/core.py:289-335). Returns 0 if None. */
static int wall_force(const gmpe_wall* wl, double px, double py, double size, int ghost, double kf, double km, double f[2]) {
    if (ghost && !wl->hard) return 0;
    const int prll_dim = wl->orient == 0 ? 0 : 1;
    const double p[2] = {px, py};
    const double prll = p[prll_dim], perp = p[1 - prll_dim];
    double theta, dist_min;
    if (prll < wl->end0 - size || prll > wl->end1 + size) return 0;
    else if (prll < wl->end0 || prll > wl->end1) {
        const double dist_past_end = prll < wl->end0 ? prll - wl->end0 : prll - wl->end1;
        theta = asin(dist_past_end / size);
        dist_min = cos(theta) * size + 0.5 * wl->width;
    } else { theta = 0; dist_min = size + 0.5 * wl->width; }
    const double delta = perp - wl->axis_pos;
    const double dist = fabs(delta);
    const double pen = logaddexp0(-(dist - dist_min) / km) * km;
    const double fm = kf * delta / dist * pen;
    f[1 - prll_dim] = cos(theta) * fm;
    f[prll_dim] = sin(theta) * fabs(fm);
    return 1;
}

/* Force-based World.step core: apply_action_force + apply_environment_force + integrate_state.
 *   DI flavour  (multiagent/core.py:766-845, 872-906): d_min = const COLLISION_DISTANCE, mass ratio 1,
 *               side skipped when status==True (899-900), separate wall constants, p_dist/time odometers.
 *   classic     (onpolicy/envs/mpe/core.py:205-335): d_min = size_a+size_b, walls share contact params.
 * pos/vel: [n_ent][2]; force_in: [n_agents][2] already = mass*accel*u (NULL rows impossible: agents are
 * movable). Accumulation order is the reference's (a outer, b inner, then walls of a). */
int gmpo_force_step(int n_ent, int n_agents, double* pos, double* vel, const double* act_force,
                    const uint8_t* movable, const uint8_t* collide, const double* size, double d_min_const,
                    const uint8_t* status, const double* mass, const double* max_speed, int n_walls,
                    const gmpe_wall* walls, double dt, double damping, double kf, double km, double wkf,
                    double wkm, double* p_dist, double* tm) {
    if (n_ent > 256) return GMPE_ERR_INVALID_ARG;
    double F[256][2]; uint8_t has[256];
    for (int i = 0; i < n_ent; ++i) { F[i][0] = F[i][1] = 0; has[i] = 0; }
    for (int i = 0; i < n_agents; ++i) if (movable[i]) { F[i][0] = act_force[2 * i]; F[i][1] = act_force[2 * i + 1]; has[i] = 1; }
    for (int a = 0; a < n_ent; ++a) {
        for (int b = a + 1; b < n_ent; ++b) {
            if (!collide[a] || !collide[b]) continue;
            if (!movable[a] && !movable[b]) continue;
            const double dx = pos[2 * a] - pos[2 * b], dy = pos[2 * a + 1] - pos[2 * b + 1];
            const double dist = sqrt(dx * dx + dy * dy);
            const double dmin = size ? size[a] + size[b] : d_min_const;
            const double pen = logaddexp0(-(dist - dmin) / km) * km;
            const double fx = kf * dx / dist * pen, fy = kf * dy / dist * pen;
            int fa, fb; double rax = fx, ray = fy, rbx = -fx, rby = -fy;
            if (movable[a] && movable[b]) {
                const double ratio = mass ? mass[b] / mass[a] : 1.0;
                rax = ratio * fx; ray = ratio * fy; rbx = -(1 / ratio) * fx; rby = -(1 / ratio) * fy;
                fa = !(status && a < n_agents && status[a]); fb = !(status && b < n_agents && status[b]);
            } else { fa = movable[a]; fb = movable[b]; }
            if (fa) { F[a][0] = rax + F[a][0]; F[a][1] = ray + F[a][1]; has[a] = 1; }
            if (fb) { F[b][0] = rbx + F[b][0]; F[b][1] = rby + F[b][1]; has[b] = 1; }
        }
        if (movable[a])
            for (int w = 0; w < n_walls; ++w) {
                double wf[2];
                const double sz = size ? size[a] : 0.06;
                if (wall_force(&walls[w], pos[2 * a], pos[2 * a + 1], sz, 0, wkf, wkm, wf)) {
                    F[a][0] = F[a][0] + wf[0]; F[a][1] = F[a][1] + wf[1]; has[a] = 1;
                }
            }
    }
    for (int i = 0; i < n_ent; ++i) {           /* integrate_state, core.py:827-845 */
        if (!movable[i]) continue;
        double vx = vel[2 * i] * (1 - damping), vy = vel[2 * i + 1] * (1 - damping);
        if (has[i]) { const double m = mass ? mass[i] : 1.0; vx += (F[i][0] / m) * dt; vy += (F[i][1] / m) * dt; }
        if (max_speed && !isnan(max_speed[i])) {
            const double sp = sqrt(vx * vx + vy * vy);
            if (sp > max_speed[i]) { const double q = sqrt(vx * vx + vy * vy); vx = vx / q * max_speed[i]; vy = vy / q * max_speed[i]; }
        }
        vel[2 * i] = vx; vel[2 * i + 1] = vy;
        pos[2 * i] += vx * dt; pos[2 * i + 1] += vy * dt;
        if (p_dist && i < n_agents) { const double ax = vx * dt, ay = vy * dt; p_dist[i] += sqrt(ax * ax + ay * ay); }
        if (tm && i < n_agents) tm[i] += dt;
    }
    return GMPE_OK;
}

/* ------------------------------------------------------------------ tube geometry (July) */
/* _tube_coords, …_july.py:621-627: pos rounded to fp32 BEFORE subtracting the fp64 entrance. */
static void tube_coords(const envv* v, double px, double py, double* s, double* yy) {
    const double rx = (double)(float)px - v->tube[T_ENTX], ry = (double)(float)py - v->tube[T_ENTY];
    *s = rx * v->tube[T_EX] + ry * v->tube[T_EY];
    *yy = rx * v->tube[T_NX] + ry * v->tube[T_NY];
}
/* get_agent_phase, …_july.py:683-733 — NOTE: mutates cooldown and previous_phase. */
static int get_agent_phase(envv* v, int i) {
    const double eps = 0.05, L = v->tube[T_L], hw = v->tube[T_HALFW];
    double s, yy; tube_coords(v, v->x[i], v->y[i], &s, &yy);
    const int in_tube = (-eps <= s && s <= L + eps) && (fabs(yy) <= hw + eps);
    const double tdx = v->tube[T_EXX] - v->tube[T_ENTX], tdy = v->tube[T_EXY] - v->tube[T_ENTY];
    const double tn = sqrt(tdx * tdx + tdy * tdy);
    const double ux = tdx / tn, uy = tdy / tn;
    const int passed = ((v->x[i] - v->tube[T_EXX]) * ux + (v->y[i] - v->tube[T_EXY]) * uy) > 0;
    const double gate_front = 0.08 * L, gate_back = 0.02 * L;           /* 612-613, 633-637 */
    const int valid_entrance = (-gate_back - eps <= s && s <= gate_front + eps) && (fabs(yy) <= hw + eps);
    if (v->cooldown[i] > 0) v->cooldown[i] -= 1;
    if (!in_tube && !passed) return 0;
    else if (in_tube) {
        if (v->prev_phase[i] == 0) return valid_entrance ? 1 : 0;
        return 1;
    } else {
        if (v->prev_phase[i] == 1) { if (passed) { v->prev_phase[i] = 2; return 2; } }
        else if (v->prev_phase[i] == 2 && passed) return 2;
        return 0;
    }
}

static double goal_block(envv* v, int i);
static double collision_block(envv* v, int i);

/* ------------------------------------------------------------------ rot_inv geometry + phase
 * nav_graph_metered_single_corridor_rot_inv.py:639-669 (gates), 675-739 (get_agent_phase). */
static double entrance_gate_distance(double s, double y, double hw) {       /* :646-652 */
    const double cy = clipd(y, -hw, hw);
    return hypot(fabs(s), y - cy);
}
static double exit_gate_distance(double s, double y, double L, double hw) { /* :660-669, penalize_backward=False */
    const double cy = clipd(y, -hw, hw);
    const double ds = (L - s) > 0.0 ? (L - s) : 0.0;
    return hypot(ds, y - cy);
}
static int in_tube_rect(double s, double y, double L, double hw) { const double eps = 0.05; return (-eps <= s && s <= L + eps) && (fabs(y) <= hw + eps); }
static int in_entrance_gate(double s, double y, double L, double hw) {
    const double eps = 0.05, gf = 0.08 * L, gb = 0.02 * L;
    return (-gb - eps <= s && s <= gf + eps) && (fabs(y) <= hw + eps);
}
static int in_exit_gate(double s, double y, double L, double hw, double back_ratio) {   /* :654-658; exit_back_ratio 0.05 (rot_inv:619) / 0.02 (two_phase_graph.py:589) */
    const double eps = 0.05, eb = back_ratio * L, ef = 0.08 * L;
    return (L - eb - eps <= s && s <= L + ef + eps) && (fabs(y) <= hw + eps);
}
/* rot_inv get_agent_phase: mutates only the cooldown; depends on previous_phase AND phase_reached. */
static int get_agent_phase_rot(envv* v, int i) {
    const gmpe_config* c = &v->h->c;
    const double L = v->tube[T_L], hw = v->tube[T_HALFW];
    double s, yy; tube_coords(v, v->x[i], v->y[i], &s, &yy);
    const int in_tube = in_tube_rect(s, yy, L, hw), passed = s > L;
    const int valid_entrance = in_entrance_gate(s, yy, L, hw), valid_exit = in_exit_gate(s, yy, L, hw, is_phasefam(c) ? 0.02 : 0.05);
    if (v->cooldown[i] > 0) v->cooldown[i] -= 1;
    if (!in_tube && !passed) return 0;
    else if (in_tube) {
        if (v->prev_phase[i] == 0) return valid_entrance ? 1 : 0;
        if (is_phasefam(c) && v->prev_phase[i] == 1 && valid_exit) return 2;                       /* two_phase_graph.py:679-685 */
        if (c->scenario == GMPE_SCENARIO_THREE_PHASE && v->prev_phase[i] == 2 && valid_exit) return 2;  /* three_phase_graph.py:681-683 */
        return 1;
    }
    if (passed) {
        if (v->phase_reached[i] >= 1) {
            if (v->prev_phase[i] == 1 && valid_exit) return 2;
            else if (v->prev_phase[i] == 2) return 2;
            return 0;
        }
    }
    return 0;
}
/* get_rotated_position_from_relative (…rot_inv.py:91-97): [[c, s], [-s, c]] @ v */
static void rotate(double th, double vx, double vy, double* ox, double* oy) {
    const double c = cos(th), s = sin(th);
    *ox = c * vx + s * vy; *oy = -s * vx + c * vy;
}
#define F32(x) ((double)(float)(x))
/* Scenario.observation, …rot_inv.py:1453-1548 — 13 floats, every entry rounded to float32 like the reference. */
/* np.float64 % float (npy_divmod): fmod, then the sign of the divisor */
static double pymod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) { if ((b < 0) != (m < 0)) m += b; } else m = copysign(0.0, b);
    return m;
}
/* (theta - corridor_heading + pi) % 2pi - pi with corridor_heading = arctan2(e[1], e[0]) (two_phase_graph.py:1060-1063, 1213-1215) */
static double heading_error_signed(const envv* v, double th) {
    const double ch = atan2(v->tube[T_EY], v->tube[T_EX]);
    return pymod(th - ch + M_PI, 2 * M_PI) - M_PI;
}
static void observation_rot(envv* v, int i, double* o) {
    const gmpe_config* c = &v->h->c;
    const int A = v->A, phasefam = is_phasefam(c);
    const double px = v->x[i], py = v->y[i], th = v->s2[i];
    double gx, gy;
    if (c->scenario == GMPE_SCENARIO_TWO_PHASE) rotate(th, F32(v->tube[T_EXX]) - px, F32(v->tube[T_EXY]) - py, &gx, &gy);   /* float32(exit) - pos (two_phase_graph.py:1196-1198) */
    else rotate(th, v->lm[2 * i] - px, v->lm[2 * i + 1] - py, &gx, &gy);
    int b1 = -1, b2 = -1; double d1 = 0, d2 = 0;
    for (int k = 0; k < A; ++k) {
        if (k == i) continue;
        if (phasefam && v->status[k]) continue;                    /* completed agents are "ghosts" (two_phase_graph.py:1174-1176) */
        const double d = norm2(v->x[k] - px, v->y[k] - py);
        if (b1 < 0 || d < d1) { b2 = b1; d2 = d1; b1 = k; d1 = d; }
        else if (b2 < 0 || d < d2) { b2 = k; d2 = d; }
    }
    double n1x = 0, n1y = 0, n2x = 0, n2y = 0;
    if (b1 >= 0) rotate(th, F32(v->x[b1] - px), F32(v->y[b1] - py), &n1x, &n1y);   /* rel vec cast to float32 BEFORE the rotation */
    if (b2 >= 0) rotate(th, F32(v->x[b2] - px), F32(v->y[b2] - py), &n2x, &n2y);
    const int phase = get_agent_phase_rot(v, i);
    const double L = v->tube[T_L], hw = v->tube[T_HALFW];
    double s, yy; tube_coords(v, px, py, &s, &yy);
    o[0] = F32(cos(th)); o[1] = F32(sin(th)); o[2] = F32(v->s3[i]);
    o[3] = F32(gx); o[4] = F32(gy); o[5] = F32(n1x); o[6] = F32(n1y); o[7] = F32(n2x); o[8] = F32(n2y);
    o[9] = F32(clipd(s / L, -2.0, 2.0)); o[10] = F32(clipd(yy / (hw + 1e-9), -2.0, 2.0));
    o[11] = F32(exit_gate_distance(s, yy, L, hw) / (L + 1e-9));
    if (phasefam) { const double he = heading_error_signed(v, th); o[12] = F32(cos(he)); o[13] = F32(sin(he)); o[14] = (double)phase; }
    else o[12] = (double)phase;
}
/* Scenario.reward, …rot_inv.py:1122-1338 */
static double reward_rot(envv* v, int i) {
    const gmpe_config* c = &v->h->c;
    const int A = v->A;
    double rew = 0;
    const int sc = c->scenario, two = sc == GMPE_SCENARIO_TWO_PHASE, three = sc == GMPE_SCENARIO_THREE_PHASE;
    int cp = get_agent_phase_rot(v, i);
    if (sc == GMPE_SCENARIO_ROT_INV) rew += collision_block(v, i);
    else if (three) { for (int a = 0; a < A; ++a) { if (a == i) continue; if (is_collision(v, a, i)) rew -= c->collision_rew; } }   /* three_phase_graph.py:965-970; two_phase: none */
    const double tdx = v->tube[T_EXX] - v->tube[T_ENTX], tdy = v->tube[T_EXY] - v->tube[T_ENTY];
    const double tlen = sqrt(tdx * tdx + tdy * tdy);
    const double px = v->x[i], py = v->y[i];
    const double th_pre = v->s2[i];
    const double hx = cos(th_pre), hy = sin(th_pre);
    const double L = v->tube[T_L], hw = v->tube[T_HALFW];
    double s, yy; tube_coords(v, px, py, &s, &yy);
    int front = -1, back = -1; double fproj = 0, bproj = 0;
    for (int k = 0; k < A; ++k) {
        if (k == i) continue;
        const double pj = (v->x[k] - px) * hx + (v->y[k] - py) * hy;
        if (pj > 0) { if (front < 0 || pj < fproj) { front = k; fproj = pj; } }
        else { if (back < 0 || pj > bproj) { back = k; bproj = pj; } }
    }
    if (cp == 2 && cp > v->prev_phase[i] + 1) rew -= c->goal_rew;
    const double ux = tdx / tlen, uy = tdy / tlen;
    const double proj = (px - v->tube[T_ENTX]) * ux + (py - v->tube[T_ENTY]) * uy;
    if (cp == v->prev_phase[i] + 1 && v->phase_reached[i] == cp - 1) {
        if (cp == 1 && in_entrance_gate(s, yy, L, hw) && v->cooldown[i] == 0) {
            rew += c->goal_rew;
            v->cooldown[i] = (two || three) ? c->episode_length : (int32_t)((double)c->episode_length / 10);   /* float assigned into an int32 array (:1200, :228); two_phase_graph.py:228 */
            v->phase_reached[i] = 1;
        } else if (cp == 2) {
            rew += c->goal_rew; v->phase_reached[i] = 2;
            if (two && !v->status[i]) {                                      /* two_phase_graph.py:1040-1044: the episode ends for this agent at the exit gate */
                v->status[i] = 1;
                v->s2[i] = uniform(v->h, v->n, 0.0, 2 * M_PI); v->s3[i] = c->v_min;
                rew += c->goal_rew * 5;
            }
        }
    }
    const double herr = fabs(heading_error_signed(v, th_pre));               /* pre-reward heading (read at :984 before any redraw) */
    if (cp == 0) {
        const double de = entrance_gate_distance(s, yy, hw);
        rew -= de;
        if ((two || three) && de < c->world_size * 0.1) rew -= herr * c->formation_rew * 0.5;   /* two_phase_graph.py:1057-1067 */
    }
    else if (cp == 1) {
        double err = 0;
        if (front >= 0) { const double diff = norm2(v->x[front] - px, v->y[front] - py) - c->sep_dist; err += diff < 0 ? fabs(diff) : 0; }
        if (back >= 0) { const double diff = norm2(v->x[back] - px, v->y[back] - py) - c->sep_dist; err += diff < 0 ? fabs(diff) : 0; }
        if (err > 0) v->spacing_viol[i] += 1;
        rew -= err * c->formation_rew;
        rew -= exit_gate_distance(s, yy, L, hw);
        if (!(two || three)) {
            const double progress_gain = c->goal_rew / (c->world_size * 0.8 * 10);   /* :522 goal_rew / (tube_length*10) */
            const double dproj = proj - v->prev_proj[i];
            rew += progress_gain * (dproj > -0.05 ? dproj : -0.05);
            v->prev_proj[i] = F32(proj);                                        /* prev_proj is a float32 array (:374) */
        } else rew -= herr * c->formation_rew * 0.1;                            /* two_phase_graph.py:1101-1106 */
        if (!two) v->h->delta_spacing[v->n] += err;                             /* two_phase_graph.py never appends (its Delta_spacing is 0) */
        v->steps_in_corr[i] += 1;
    } else if (!(two || three) && cp == 2 && v->phase_reached[i] == 0) cp = 0;
    else if (cp == 2 && !two) {
        if (three) {                                                            /* three_phase_graph.py:1110-1123: as goal_block, but goal_tracker stays untouched */
            const double d = norm2(px - v->lm[2 * i], py - v->lm[2 * i + 1]);
            if (d < c->goal_thresh) {
                if (!v->status[i]) { v->status[i] = 1; v->s2[i] = uniform(v->h, v->n, 0.0, 2 * M_PI); v->s3[i] = c->v_min; rew += c->goal_rew * 5; }
            } else rew -= d;
        } else rew += goal_block(v, i);
    }
    if (v->phase_reached[i] == 1 && cp == 0) v->conformance[i] += 1;
    if (cp > v->phase_reached[i]) v->phase_reached[i] = cp;
    if (cp < v->prev_phase[i]) rew -= c->collision_rew;
    if (cp < v->phase_reached[i]) rew -= c->collision_rew;
    v->prev_phase[i] = cp;
    if (in_tube_rect(s, yy, L, hw) && cp != 1 && !(three && in_exit_gate(s, yy, L, hw, 0.02))) rew -= c->collision_rew;   /* three_phase_graph.py:1145 */
    if (s > L && v->phase_reached[i] < 1) rew -= c->goal_rew;
    return clipd(rew, -4 * c->collision_rew, c->goal_rew * 5);
}
/* _get_entity_feat_relative, …rot_inv.py:1690-1766: float32 positions/velocities, differences in float32, rotated by
 * the ego heading in float64, rounded to float32. node: [E,7]. The adjacency part is shared with the July file. */
static void node_features_rot(envv* v, int i, double* node) {
    const int A = v->A, L = v->L, E = v->E;
    double evx, evy; agent_vel(v, i, &evx, &evy);
    const float apx = (float)v->x[i], apy = (float)v->y[i], avx = (float)evx, avy = (float)evy;
    const double th = v->s2[i];
    for (int k = 0; k < E; ++k) {
        double kx, ky, kvx = 0.0, kvy = 0.0; ent_pos(v, k, &kx, &ky);
        if (k < A) agent_vel(v, k, &kvx, &kvy);
        const float rpx = (float)kx - apx, rpy = (float)ky - apy, rvx = (float)kvx - avx, rvy = (float)kvy - avy;
        double* r = node + 7 * k;
        double ox, oy;
        rotate(th, (double)rvx, (double)rvy, &ox, &oy); r[0] = F32(ox); r[1] = F32(oy);
        rotate(th, (double)rpx, (double)rpy, &ox, &oy); r[2] = F32(ox); r[3] = F32(oy);
        if (k < A) {
            const int two = v->h->c.scenario == GMPE_SCENARIO_TWO_PHASE;          /* two_phase_graph.py:1405: every agent's goal node feature is the corridor exit */
            const float gx = (float)(two ? v->tube[T_EXX] : v->lm[2 * k]) - apx, gy = (float)(two ? v->tube[T_EXY] : v->lm[2 * k + 1]) - apy;
            rotate(th, (double)gx, (double)gy, &ox, &oy); r[4] = F32(ox); r[5] = F32(oy); r[6] = 0.0;
        } else { r[4] = r[2]; r[5] = r[3]; r[6] = k < A + L ? 1.0 : 2.0; }
    }
}

/* ------------------------------------------------------------------ observation */
/* Scenario.observation, …_july.py:1337-1463 (navigation_graph: slice [0:13]). */
static void observation(envv* v, int i, double* o) {
    const int A = v->A;
    double vx, vy; agent_vel(v, i, &vx, &vy);
    const double px = v->x[i], py = v->y[i];
    const double gx = v->lm[2 * i] - px, gy = v->lm[2 * i + 1] - py;
    o[0] = px; o[1] = py; o[2] = vx; o[3] = vy; o[4] = gx; o[5] = gy; o[6] = 0.0; o[7] = gx; o[8] = gy;
    /* two nearest other agents, stable sort by distance (1398-1417) */
    int b1 = -1, b2 = -1; double d1 = 0, d2 = 0;
    for (int k = 0; k < A; ++k) {
        if (k == i) continue;
        const double d = norm2(v->x[k] - px, v->y[k] - py);
        if (b1 < 0 || d < d1) { b2 = b1; d2 = d1; b1 = k; d1 = d; }
        else if (b2 < 0 || d < d2) { b2 = k; d2 = d; }
    }
    o[9] = b1 >= 0 ? v->x[b1] - px : 0.0; o[10] = b1 >= 0 ? v->y[b1] - py : 0.0;
    o[11] = b2 >= 0 ? v->x[b2] - px : 0.0; o[12] = b2 >= 0 ? v->y[b2] - py : 0.0;
    if (v->h->c.scenario == GMPE_SCENARIO_TUBE_JULY) {
        o[13] = v->tube[T_ENTX] - px; o[14] = v->tube[T_ENTY] - py;
        o[15] = v->tube[T_EXX] - px; o[16] = v->tube[T_EXY] - py;
        o[17] = v->tube[T_WIDTH];
        o[18] = (double)get_agent_phase(v, i);          /* first phase call of the step (1447) */
    }
}

/* ------------------------------------------------------------------ reward */
/* goal block, …_july.py:1185-1194; returns reward delta. reset_velocity(): core.py:324-333 / 223-225 */
static double goal_block(envv* v, int i) {
    const gmpe_config* c = &v->h->c;
    const double d = norm2(v->x[i] - v->lm[2 * i], v->y[i] - v->lm[2 * i + 1]);
    if (d < c->goal_thresh) {
        if (!v->status[i]) {
            v->status[i] = 1;
            if (is_kinematic(v->h)) { v->s2[i] = uniform(v->h, v->n, 0.0, 2 * M_PI); v->s3[i] = c->v_min; }
            else { v->s2[i] = 0.0; v->s3[i] = 0.0; }
            v->goal_tracker[i] = i;
            return c->goal_rew * 5;
        }
        return 0.0;
    }
    return -d;
}
static double collision_block(envv* v, int i) {
    const gmpe_config* c = &v->h->c;
    double rew = 0;
    for (int a = 0; a < v->A; ++a) { if (a == i) continue; if (is_collision(v, a, i)) rew -= c->collision_rew * 4; }
    if (is_obstacle_collision(v, v->x[i], v->y[i], c->entity_size)) rew -= c->collision_rew * 3;
    return rew;
}
/* Scenario.reward, …_july.py:1105-1221. `spacing_out` gets this step's spacing error (phase 1). */
static double reward_july(envv* v, int i) {
    const gmpe_config* c = &v->h->c;
    const int A = v->A;
    double rew = 0;
    int cp = get_agent_phase(v, i);                      /* second phase call (1113) */
    rew += collision_block(v, i);
    const double tdx = v->tube[T_EXX] - v->tube[T_ENTX], tdy = v->tube[T_EXY] - v->tube[T_ENTY];
    const double tlen = sqrt(tdx * tdx + tdy * tdy);
    const double px = v->x[i], py = v->y[i];
    const double hx = cos(v->s2[i]), hy = sin(v->s2[i]);
    int front = -1, back = -1; double fproj = 0, bproj = 0;
    for (int k = 0; k < A; ++k) {                        /* 1136-1143: min over proj>0, max over proj<=0, first wins ties */
        if (k == i) continue;
        const double proj = (v->x[k] - px) * hx + (v->y[k] - py) * hy;
        if (proj > 0) { if (front < 0 || proj < fproj) { front = k; fproj = proj; } }
        else { if (back < 0 || proj > bproj) { back = k; bproj = proj; } }
    }
    if (cp == 2 && cp > v->prev_phase[i] + 1) rew -= c->goal_rew * 3;
    const double tn = sqrt(tdx * tdx + tdy * tdy);
    const double ux = tdx / tn, uy = tdy / tn;
    const double qx = px - v->tube[T_ENTX], qy = py - v->tube[T_ENTY];
    const double proj = qx * ux + qy * uy;
    const double entrance_dist = norm2(qx - proj * tdx, qy - proj * tdy);   /* un-normalised tdir: 1154 */
    if (cp == v->prev_phase[i] + 1 && v->phase_reached[i] == cp - 1) {
        if (cp == 1 && 0 <= proj && proj < 0.1 * tlen && entrance_dist < 0.2 * tlen) rew += c->goal_rew * 3;
        else if (cp == 2) rew += c->goal_rew * 3;
    }
    if (cp == 0) rew -= norm2(v->tube[T_ENTX] - px, v->tube[T_ENTY] - py);
    else if (cp == 1) {
        double err = 0;
        if (front >= 0) { const double diff = norm2(v->x[front] - px, v->y[front] - py) - c->sep_dist; err += diff < 0 ? fabs(diff) : 0; }
        if (back >= 0) { const double diff = norm2(v->x[back] - px, v->y[back] - py) - c->sep_dist; err += diff < 0 ? fabs(diff) : 0; }
        if (err > 0) v->spacing_viol[i] += 1;
        rew -= err * c->formation_rew;
        rew -= norm2(v->tube[T_EXX] - px, v->tube[T_EXY] - py);
        v->h->delta_spacing[v->n] += err;
        v->steps_in_corr[i] += 1;
    } else if (cp == 2 && v->phase_reached[i] == 0) cp = 0;
    else rew += goal_block(v, i);
    if (v->phase_reached[i] == 1 && cp == 0) v->conformance[i] += 1;
    if (cp > v->phase_reached[i]) v->phase_reached[i] = cp;
    if (cp < v->prev_phase[i]) rew -= c->collision_rew * 3;
    if (cp < v->phase_reached[i]) rew -= c->collision_rew;
    v->prev_phase[i] = cp;
    rew = clipd(rew, -4 * c->collision_rew, c->goal_rew * 5);
    return clipd(rew, c->min_reward, c->max_reward);
}
/* navigation_graph reward = collision block (…_july.py:1117-1124) + goal block (1185-1194) + clips. */
static double reward_nav(envv* v, int i) {
    const gmpe_config* c = &v->h->c;
    double rew = collision_block(v, i);
    rew += goal_block(v, i);
    rew = clipd(rew, -4 * c->collision_rew, c->goal_rew * 5);
    return clipd(rew, c->min_reward, c->max_reward);
}

/* ------------------------------------------------------------------ graph observation */
/* graph_observation + _get_entity_feat_relative, …_july.py:1584-1649, 1694-1771.
 * node: [E,8] for ego i. The adjacency is world.cached_dist_mag itself, masked IN PLACE. */
static void graph_observation(envv* v, int i, double* node) {
    const int A = v->A, L = v->L, E = v->E;
    if (v->h->c.graph_feat_type == 1) {                  /* _get_entity_feat_global, …_july.py:1672-1691 (same code in rot_inv.py:1668-1687, two_phase_graph.py,
                                                            three_phase_graph.py): [vel, pos, goal, type] in world coordinates */
        for (int k = 0; k < E; ++k) {
            double kx, ky, kvx = 0.0, kvy = 0.0; ent_pos(v, k, &kx, &ky);
            if (k < A) agent_vel(v, k, &kvx, &kvy);
            double* r = node + 7 * k;
            r[0] = kvx; r[1] = kvy; r[2] = kx; r[3] = ky;
            if (k < A) { r[4] = v->lm[2 * k]; r[5] = v->lm[2 * k + 1]; r[6] = 0.0; }
            else { r[4] = kx; r[5] = ky; r[6] = k < A + L ? 1.0 : 2.0; }
        }
        goto mask;
    }
    if (is_rotfam(&v->h->c)) { node_features_rot(v, i, node); goto mask; }
    {
    double evx, evy; agent_vel(v, i, &evx, &evy);
    const double px = v->x[i], py = v->y[i];
    for (int k = 0; k < E; ++k) {
        double kx, ky, kvx = 0.0, kvy = 0.0; ent_pos(v, k, &kx, &ky);
        if (k < A) agent_vel(v, k, &kvx, &kvy);
        double* r = node + 8 * k;
        r[0] = kvx - evx; r[1] = kvy - evy; r[2] = kx - px; r[3] = ky - py;
        if (k < A) { r[4] = v->lm[2 * k] - px; r[5] = v->lm[2 * k + 1] - py; r[6] = 0.0; r[7] = 0.0; }
        else { r[4] = r[2]; r[5] = r[3]; r[6] = 1.0; r[7] = k < A + L ? 1.0 : 2.0; }
    }
    }
mask:
    for (int k = 0; k < A + L; ++k) {                    /* mask sized to E: obstacle rows never masked */
        int off;
        if (k < A) off = v->status[k];
        else { off = 0; for (int a = 0; a < A; ++a) if (v->goal_tracker[a] == k - A) off = 1; }
        if (off) for (int j = 0; j < E; ++j) { v->dist[k * E + j] = 0.0; v->dist[j * E + k] = 0.0; }
    }
}

/* ------------------------------------------------------------------ info */
/* info_callback, …_july.py:741-829. out[17] in INFO_KEYS order (tests/golden/make_fixtures.py). */
static void info_callback(envv* v, int i, double rew, double* out) {
    const gmpe_config* c = &v->h->c;
    const int A = v->A;
    int nearest = 0; double dmin = 0;
    for (int l = 0; l < v->L; ++l) {
        const double d = norm2(v->x[i] - v->lm[2 * l], v->y[i] - v->lm[2 * l + 1]);
        if (l == 0 || d < dmin) { dmin = d; nearest = l; }
    }
    const double thr = c->goal_thresh;
    const int32_t tnow = (int32_t)((double)v->h->current_step[v->n] * c->dt);
    if (dmin < thr && (nearest != v->goal_reached[i] && v->goal_reached[i] != -1)) { v->goal_reached[i] = nearest; v->dist_left[i] = (int32_t)dmin; }
    if (dmin < thr && v->times_required[i] == -1) {
        v->times_required[i] = tnow; v->dists_to_goal[i] = (int32_t)v->p_dist[i]; v->dist_left[i] = (int32_t)dmin; v->goal_reached[i] = nearest;
    }
    if (v->times_required[i] == -1) { v->dists_to_goal[i] = (int32_t)v->p_dist[i]; v->dist_left[i] = (int32_t)dmin; }
    if (dmin > thr && v->times_required[i] != -1) { v->dists_to_goal[i] = (int32_t)v->p_dist[i]; v->times_required[i] = tnow; v->dist_left[i] = (int32_t)dmin; }
    if (dmin < thr && nearest == v->goal_reached[i]) { v->dist_left[i] = (int32_t)dmin; v->goal_reached[i] = nearest; }
    if (is_obstacle_collision(v, v->x[i], v->y[i], c->entity_size)) v->n_obst_coll[i] += 1;
    for (int a = 0; a < A; ++a) { if (a == i) continue; if (is_collision(v, i, a)) v->n_agent_coll[i] += 1; }
    double dm = 0, tm = 0;
    for (int a = 0; a < A; ++a) { dm += v->dists_to_goal[a]; tm += v->times_required[a]; }
    dm /= A; tm /= A;
    double dv = 0, tv = 0;
    for (int a = 0; a < A; ++a) { const double p = v->dists_to_goal[a] - dm, q = v->times_required[a] - tm; dv += p * p; tv += q * q; }
    const double ds = sqrt(dv / A), ts = sqrt(tv / A);
    double svsum = 0; for (int a = 0; a < A; ++a) svsum += v->spacing_viol[a];
    out[0] = rew; out[1] = v->dist_left[i]; out[2] = v->times_required[i]; out[3] = v->n_agent_coll[i];
    out[4] = v->n_obst_coll[i]; out[5] = dm; out[6] = ds; out[7] = dm / (ds + 0.0001);
    out[8] = v->dists_to_goal[i]; out[9] = v->times_required[i]; out[10] = tm; out[11] = ts; out[12] = tm / (ts + 0.0001);
    out[13] = (double)v->conformance[i] / c->episode_length;
    out[14] = v->h->delta_spacing[v->n] / (svsum != 0 ? svsum : 1);
    out[15] = (double)v->spacing_viol[i] / (v->steps_in_corr[i] != 0 ? v->steps_in_corr[i] : 1);
    out[16] = v->goal_min_time[i];
    out[17] = v->phase_reached[i];                        /* 'Phase_reached' (rot_inv.py:835) */
}

/* ------------------------------------------------------------------ reset */
static void reset_counters(envv* v) {             /* reset_world, …_july.py:339-374, 391-392 */
    for (int i = 0; i < v->A; ++i) {
        v->times_required[i] = -1; v->dists_to_goal[i] = -1; v->dist_left[i] = -1; v->n_obst_coll[i] = 0;
        v->n_agent_coll[i] = 0; v->goal_reached[i] = -1; v->goal_tracker[i] = -1; v->conformance[i] = 0;
        v->spacing_viol[i] = 0; v->steps_in_corr[i] = 0; v->phase_reached[i] = 0; v->cooldown[i] = 0;
        v->p_dist[i] = 0.0; v->time[i] = 0.0; v->prev_proj[i] = 0.0;
    }
    v->h->delta_spacing[v->n] = 0.0;
}
static void min_times(envv* v) {                  /* min_time, …_july.py:941-951 */
    const double ms = v->h->c.max_speed;
    for (int i = 0; i < v->A; ++i) {
        const double dx = v->x[i] - v->lm[2 * i], dy = v->y[i] - v->lm[2 * i + 1];
        v->goal_min_time[i] = ms > 0 ? sqrt(dx * dx + dy * dy) / ms : 0.0;
    }
}
/* reset_world (July): RNG draw order of SURVEY.md §3.3 — wall_length(1), tube angle(1), per placement
 * attempt jitter(2), on accept heading(1). …_july.py:339-420, 440-515, 518-613; utils.py:165-193. */
static void reset_world_july(envv* v) {
    gmpo* h = v->h; const gmpe_config* c = &h->c; const int n = v->n;
    const double ws = c->world_size, size = c->entity_size;
    reset_counters(v);
    (void)uniform(h, n, 0.2, 0.8);                                    /* wall_length, value unused (368) */
    const double a = 3 * size * 2.5, b = ws * 0.15;
    const double width = a > b ? a : b;                               /* 525-528 */
    const double angle = uniform(h, n, -M_PI / 2, M_PI / 2);          /* 530 */
    double tl = ws * 0.8;
    if (is_phasefam(c)) tl += uniform(h, n, -ws * 0.3, ws * 0.1);      /* two_phase_graph.py:506 */
    const double ca = cos(angle), sa = sin(angle);
    const double be = tl / 4, bx = -tl / 4;
    const double entx = ca * 0 + sa * be, enty = -sa * 0 + ca * be;   /* R @ [0, +len/4] (545-553) */
    const double exx = ca * 0 + sa * bx, exy = -sa * 0 + ca * bx;
    const double dx = exx - entx, dy = exy - enty;
    const double L = sqrt(dx * dx + dy * dy) + 1e-9;                  /* 600 */
    const double ex = dx / L, ey = dy / L;
    double* t = v->tube;
    t[T_ANGLE] = angle; t[T_ENTX] = entx; t[T_ENTY] = enty; t[T_EXX] = exx; t[T_EXY] = exy; t[T_EX] = ex; t[T_EY] = ey;
    t[T_NX] = (double)(float)(-ey); t[T_NY] = (double)(float)ex;      /* n stored float32 (602) */
    t[T_L] = L; t[T_HALFW] = width * 0.5; t[T_WIDTH] = width;
    int k = 0, tries = 0;
    while (k < v->A) {                                                /* random_scenario 452-486 */
        const double u0 = draw(h, n), u1 = draw(h, n);
        const int rot = is_rotfam(c);                                        /* rot_inv.py:463, 469: 0.3 and /3 */
        const double jf = rot ? 0.3 : 0.2;
        const double jx = jf * (-ws + (ws - (-ws)) * u0), jy = jf * (-ws + (ws - (-ws)) * u1);
        const double dfe = rot ? (ws + k) / 3 : (ws + k) / 5;
        const double px = entx + dfe * sa + jx, py = enty + dfe * ca + jy;
        int bad = is_obstacle_collision(v, px, py, size);
        for (int q = 0; q < k && !bad; ++q) if (norm2(v->x[q] - px, v->y[q] - py) < c->sep_dist) bad = 1;   /* 895-904 */
        if (bad && ++tries < MAX_TRIES) continue;
        if (bad) h->error_flags[n] |= 2;                              /* the reference would spin forever */
        v->x[k] = px; v->y[k] = py;
        v->s2[k] = uniform(h, n, 0.0, 2 * M_PI); v->s3[k] = c->v_min;  /* reset_velocity (core.py:324-333) */
        v->status[k] = 0;
        ++k; tries = 0;
    }
    /* landmarks by args.formation_type (…_july.py:492-497; the same call sites in rot_inv.py:493-498, two_phase_graph.py:464-469,
     * three_phase_graph.py:459-464). The landmarks' reset_velocity() draws nothing (DoubleIntegratorXYState, core.py:400). */
    if (c->formation_type == GMPE_FORMATION_LINE) {
        /* set_landmarks_in_line(start = (-ws/2, -ws/2), end = (ws/2, -ws/2)) -> np.linspace(start, end, L) (utils.py:77-130). The y step is 0, so
         * linspace takes its "any_step_zero" branch for BOTH coordinates: y = (i / div) * delta + start, last row = stop exactly; L = 1: start. */
        const int num = v->L, div = num - 1;
        const double sx0 = -ws / 2, sy0 = -ws / 2, ex0 = ws / 2, ey0 = -ws / 2;
        const double ddx = ex0 - sx0, ddy = ey0 - sy0;
        for (int l = 0; l < num; ++l) {
            const double f = div > 0 ? (double)l / (double)div : (double)l;
            double lx = f * ddx + sx0, ly = f * ddy + sy0;
            if (num > 1 && l == num - 1) { lx = ex0; ly = ey0; }
            if (is_obstacle_collision(v, lx, ly, size)) h->error_flags[n] |= 2;     /* the reference raises ValueError (utils.py:113-114) */
            v->lm[2 * l] = lx; v->lm[2 * l + 1] = ly;
        }
    } else if (c->formation_type == GMPE_FORMATION_CIRCLE) {
        /* set_landmarks_in_circle(center = (0, exit_y + ws/5), radius = ws/3) (utils.py:231-267) */
        const double cx = 0.0, cy = exy + ws / 5, radius = ws / 3;
        const double angle_step = 2 * M_PI / v->L;
        for (int l = 0; l < v->L; ++l) {
            const double ang = l * angle_step;
            v->lm[2 * l] = cx + radius * cos(ang); v->lm[2 * l + 1] = cy + radius * sin(ang);
        }
    } else {
        /* set_landmarks_in_point (utils.py:165-193): every landmark at exit + R(angle) @ [0, -ws/3] */
        const double rel = -ws / 3;
        const double rx = ca * 0.0 + sa * rel, ry = -sa * 0.0 + ca * rel;
        for (int l = 0; l < v->L; ++l) { v->lm[2 * l] = exx + rx; v->lm[2 * l + 1] = exy + ry; }
    }
    min_times(v);
}
/* navigation_graph reset — this project's own composition: uniform placement in
 * 0.8*[-ws/2, ws/2]^2 with rejection, order obstacles -> agents -> landmarks, 2 draws per attempt. */
static void reset_world_nav(envv* v) {
    gmpo* h = v->h; const gmpe_config* c = &h->c; const int n = v->n;
    const double ws = c->world_size, size = c->entity_size;
    reset_counters(v);
    for (int o = 0, tries = 0; o < v->O;) {
        const double px = 0.8 * uniform(h, n, -ws / 2, ws / 2), py = 0.8 * uniform(h, n, -ws / 2, ws / 2);
        int bad = 0;
        for (int q = 0; q < o && !bad; ++q) if (norm2(v->ob[2 * q] - px, v->ob[2 * q + 1] - py) < 2.0 * (size + size)) bad = 1;
        if (bad && ++tries < MAX_TRIES) continue;
        if (bad) h->error_flags[n] |= 2;
        v->ob[2 * o] = px; v->ob[2 * o + 1] = py; ++o; tries = 0;
    }
    const int O_all = v->O;
    for (int k = 0, tries = 0; k < v->A;) {
        const double px = 0.8 * uniform(h, n, -ws / 2, ws / 2), py = 0.8 * uniform(h, n, -ws / 2, ws / 2);
        int bad = is_obstacle_collision(v, px, py, size);
        for (int q = 0; q < k && !bad; ++q) if (norm2(v->x[q] - px, v->y[q] - py) < c->sep_dist) bad = 1;
        if (bad && ++tries < MAX_TRIES) continue;
        if (bad) h->error_flags[n] |= 2;
        v->x[k] = px; v->y[k] = py; v->s2[k] = 0.0; v->s3[k] = 0.0; v->status[k] = 0; ++k; tries = 0;
    }
    (void)O_all;
    for (int l = 0, tries = 0; l < v->L;) {
        const double px = 0.8 * uniform(h, n, -ws / 2, ws / 2), py = 0.8 * uniform(h, n, -ws / 2, ws / 2);
        int bad = is_obstacle_collision(v, px, py, size);
        for (int q = 0; q < l && !bad; ++q) if (norm2(v->lm[2 * q] - px, v->lm[2 * q + 1] - py) < c->sep_dist) bad = 1;
        if (bad && ++tries < MAX_TRIES) continue;
        if (bad) h->error_flags[n] |= 2;
        v->lm[2 * l] = px; v->lm[2 * l + 1] = py; ++l; tries = 0;
    }
    min_times(v);
}

/* MultiAgentGraphEnv.reset, environment.py:1066-1081. Outputs may be NULL. adj: [E,E] (one matrix). */
static void env_reset(envv* v, double* obs, int32_t* ids, double* node, double* adj) {
    gmpo* h = v->h; const int A = v->A, E = v->E, D = h->D;
    h->current_step[v->n] = 0;
    if (is_tube(&h->c)) reset_world_july(v); else reset_world_nav(v);
    calculate_distances(v);                     /* initialize_min_time_distance_graph (735-739) */
    table_stage(v, 0); table_stage(v, 1);
    double otmp[32], ntmp[GMPE_MAX_ENTITIES * 8];
    const int F = h->F, rotinv = is_rotfam(&h->c);
    for (int i = 0; i < A; ++i) {
        if (rotinv) observation_rot(v, i, otmp); else observation(v, i, otmp);
        graph_observation(v, i, ntmp);
        if (obs) memcpy(obs + (size_t)i * D, otmp, sizeof(double) * D);
        if (ids) ids[i] = i;
        if (node) memcpy(node + (size_t)i * E * F, ntmp, sizeof(double) * E * F);
    }
    if (adj) memcpy(adj, v->dist, sizeof(double) * E * E);
}

int gmpo_reset(gmpo* h, const uint8_t* mask, double* obs, int32_t* ids, double* node, double* adj) {
    const int A = h->A, E = h->E, D = h->D;
    for (int n = 0; n < h->N; ++n) {
        if (mask && !mask[n]) continue;
        envv v = view(h, n);
        env_reset(&v, obs ? obs + (size_t)n * A * D : NULL, ids ? ids + (size_t)n * A : NULL,
                  node ? node + (size_t)n * A * E * h->F : NULL, adj ? adj + (size_t)n * E * E : NULL);
    }
    return GMPE_OK;
}

/* update_graph, …_july.py:1651-1670: (d <= max_edge_dist) & (d > 0), row-major (csr->coo order). */
int gmpo_update_graph(gmpo* h, int n, int32_t* edges /*[2,cap]*/, double* weights, int cap) {
    const int E = h->E; const double* d = h->dist + (size_t)n * E * E; int m = 0;
    for (int r = 0; r < E; ++r) for (int c = 0; c < E; ++c) {
        const double x = d[r * E + c];
        if (x <= h->c.coord_range && x > 0) { if (m < cap) { edges[m] = r; edges[cap + m] = c; if (weights) weights[m] = x; } ++m; }
    }
    return m;
}

/* ------------------------------------------------------------------ step */
/* _set_action (environment.py:336-475): discrete index -> control, x sensitivity. */
static void decode_action(const gmpe_config* c, int idx, double u[2]) {
    if (c->dynamics == GMPE_DYN_DOUBLE_INTEGRATOR) {
        if (c->n_actions == 5) {                 /* 399-405: u = [a1-a2, a3-a4] on the one-hot */
            u[0] = (idx == 1 ? 1.0 : 0.0) - (idx == 2 ? 1.0 : 0.0);
            u[1] = (idx == 3 ? 1.0 : 0.0) - (idx == 4 ? 1.0 : 0.0);
        } else {                                 /* 382-392 action_map */
            static const double m[9][2] = {{0, 0}, {-1, 0}, {-0.71, -0.71}, {0, -1}, {0.71, -0.71}, {1, 0}, {0.71, 0.71}, {0, 1}, {-0.71, 0.71}};
            u[0] = m[idx][0]; u[1] = m[idx][1];
        }
    } else {                                     /* 437-449 */
        const int wi = idx / 5, ai = idx - wi * 5;
        u[0] = c->ang_rate_opt[wi]; u[1] = c->accel_opt[ai];
    }
    u[0] *= c->sensitivity; u[1] *= c->sensitivity;
}

/* One env: MultiAgentGraphEnv.step (environment.py:1021-1063) then graphworker's auto-reset
 * (env_wrappers.py:865-870). Double outputs; adj is the single [E,E] matrix. Returns 1 if reset. */
static int env_step(envv* v, const int32_t* act, double* obs, int32_t* ids, double* node, double* adj,
                    double* rew, uint8_t* done, double* info, int auto_reset) {
    gmpo* h = v->h; const gmpe_config* c = &h->c; const int A = v->A, E = v->E, D = h->D, n = v->n;
    h->current_step[n] += 1;
    if (is_kinematic(h)) {                       /* World.step -> update_agent_state (core.py:819-826) */
        for (int i = 0; i < A; ++i) {
            if (v->status[i]) continue;
            double u[2]; decode_action(c, act[i], u); filtered_control(h, n, i, u);
            double s[4] = {v->x[i], v->y[i], v->s2[i], v->s3[i]};
            gmpo_kinematic_step(s, u[0], u[1], c->dt, c->v_min, c->v_max, &v->p_dist[i], &v->time[i]);
            v->x[i] = s[0]; v->y[i] = s[1]; v->s2[i] = s[2]; v->s3[i] = s[3];
        }
    } else {                                     /* force path on agents+landmarks+obstacles */
        double pos[GMPE_MAX_ENTITIES * 2], vel[GMPE_MAX_ENTITIES * 2], F[GMPE_MAX_AGENTS * 2], ms[GMPE_MAX_ENTITIES];
        uint8_t mov[GMPE_MAX_ENTITIES], col[GMPE_MAX_ENTITIES];
        const int classic = c->contact_family == 1;          /* onpolicy/envs/mpe/core.py constants: d_min = size_a + size_b, per-entity mass */
        double size[GMPE_MAX_ENTITIES], mass[GMPE_MAX_ENTITIES];
        for (int k = 0; k < E; ++k) {
            size[k] = k < A ? c->agent_size : c->collider_size; mass[k] = k < A ? c->agent_mass : 1.0;
            ent_pos(v, k, &pos[2 * k], &pos[2 * k + 1]);
            vel[2 * k] = k < A ? v->s2[k] : 0.0; vel[2 * k + 1] = k < A ? v->s3[k] : 0.0;
            mov[k] = k < A; col[k] = (k < A) || (k >= A + v->L);     /* landmarks collide=False (…_july.py:298) */
            ms[k] = (k < A && c->max_speed > 0) ? c->max_speed : NAN;
        }
        for (int i = 0; i < A; ++i) {
            double u[2]; decode_action(c, act[i], u); filtered_control(h, n, i, u);
            const double sc = classic ? c->action_force_scale : 1.0;       /* mpe/core.py:211-213: mass * accel (or mass) */
            F[2 * i] = sc * u[0]; F[2 * i + 1] = sc * u[1];
        }
        gmpo_force_step(E, A, pos, vel, F, mov, col, classic ? size : NULL, c->sep_dist, classic ? NULL : v->status, classic ? mass : NULL, ms,
                        c->num_walls, c->walls, c->dt, c->damping, c->contact_force, c->contact_margin, c->wall_contact_force,
                        c->wall_contact_margin, v->p_dist, v->time);
        for (int i = 0; i < A; ++i) { v->x[i] = pos[2 * i]; v->y[i] = pos[2 * i + 1]; v->s2[i] = vel[2 * i]; v->s3[i] = vel[2 * i + 1]; }
    }
    calculate_distances(v);
    table_stage(v, 0);
    double otmp[32], ntmp[GMPE_MAX_ENTITIES * 8], itmp[GMPE_INFO_KEYS], rsum = 0;
    int all_done = 1;
    for (int i = 0; i < A; ++i) {                /* environment.py:1036-1053, IN ORDER */
        if (is_rotfam(c)) observation_rot(v, i, otmp); else observation(v, i, otmp);
        const double r = c->scenario == GMPE_SCENARIO_TUBE_JULY ? reward_july(v, i) : (is_rotfam(c) ? reward_rot(v, i) : reward_nav(v, i));
        graph_observation(v, i, ntmp);
        const int dn = v->status[i] || h->current_step[n] >= c->episode_length;   /* _get_done 264-271 */
        info_callback(v, i, r, itmp);
        if (obs) memcpy(obs + (size_t)i * D, otmp, sizeof(double) * D);
        if (ids) ids[i] = i;
        if (node) memcpy(node + (size_t)i * E * h->F, ntmp, sizeof(double) * E * h->F);
        if (rew) rew[i] = r;
        if (done) done[i] = (uint8_t)dn;
        if (info) memcpy(info + (size_t)i * GMPE_INFO_KEYS, itmp, sizeof itmp);
        rsum += r; all_done &= dn;
    }
    if (c->collaborative && rew) for (int i = 0; i < A; ++i) rew[i] = rsum;   /* environment.py:1056-1061 */
    if (adj) memcpy(adj, v->dist, sizeof(double) * E * E);
    table_stage(v, 1);
    if (all_done && auto_reset) { env_reset(v, obs, ids, node, adj); return 1; }
    return 0;
}

/* auto_reset = 0 returns the terminal obs too and leaves the reset to the caller (gmpo_reset with a
 * mask) — used by the golden tests, which hold both the pre- and post-reset outputs. */
int gmpo_step(gmpo* h, const int32_t* act, double* obs, int32_t* ids, double* node, double* adj,
              double* rew, uint8_t* done, double* info, uint8_t* did_reset, int auto_reset) {
    const int A = h->A, E = h->E, D = h->D;
    for (int n = 0; n < h->N; ++n) {
        envv v = view(h, n);
        const int r = env_step(&v, act + (size_t)n * A, obs ? obs + (size_t)n * A * D : NULL,
                               ids ? ids + (size_t)n * A : NULL, node ? node + (size_t)n * A * E * h->F : NULL,
                               adj ? adj + (size_t)n * E * E : NULL, rew ? rew + (size_t)n * A : NULL,
                               done ? done + (size_t)n * A : NULL, info ? info + (size_t)n * A * GMPE_INFO_KEYS : NULL, auto_reset);
        if (did_reset) did_reset[n] = (uint8_t)r;
    }
    return GMPE_OK;
}

