/* asan_driver.c — runs the oracle under AddressSanitizer + UBSan (CPU build only; the GPU pool has no ASan).
 * Built and run by `make -C oracle asan-run` and tests/test_host_logic.py::test_oracle_under_sanitizers. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/gmpe.h"

typedef struct gmpo gmpo;
int gmpo_create(const gmpe_config*, gmpo**);
int gmpo_destroy(gmpo*);
int gmpo_reset(gmpo*, const uint8_t*, double*, int32_t*, double*, double*);
int gmpo_step(gmpo*, const int32_t*, double*, int32_t*, double*, double*, double*, uint8_t*, double*, uint8_t*, int);
int gmpo_update_graph(gmpo*, int, int32_t*, double*, int);

static void fill(gmpe_config* c, int scen, int N, int A, int O, int W) {
    memset(c, 0, sizeof *c);
    c->abi_version = GMPE_ABI_VERSION; c->scenario = scen; c->dynamics = scen != GMPE_SCENARIO_NAVIGATION_GRAPH ? GMPE_DYN_AIR_TAXI : GMPE_DYN_DOUBLE_INTEGRATOR;
    c->num_envs = N; c->num_agents = A; c->num_landmarks = A; c->num_obstacles = O; c->num_walls = W; c->episode_length = 6;
    c->n_actions = scen != GMPE_SCENARIO_NAVIGATION_GRAPH ? 25 : 5; c->seed = 7; c->world_size = 4.0; c->max_speed = 2.0;
    c->collision_rew = 5; c->formation_rew = 1; c->goal_rew = 5; c->min_reward = -40; c->max_reward = 50;
    if (scen != GMPE_SCENARIO_NAVIGATION_GRAPH) { c->dt = 1.0; c->v_min = 0.03086664; c->v_max = 0.0900277; c->goal_thresh = 0.35; c->sep_dist = 0.4572; c->coord_range = 4.82802;
        for (int i = 0; i < 5; ++i) { c->ang_rate_opt[i] = -0.1 + 0.05 * i; c->accel_opt[i] = -0.001 + 0.00075 * i; } }
    else { c->dt = 0.1; c->v_max = 1.0; c->goal_thresh = 0.2; c->sep_dist = 0.5; c->coord_range = 5; }
    c->sensitivity = 5; c->entity_size = 0.06; c->damping = 0.25; c->contact_force = 300; c->contact_margin = 0.02;
    c->wall_contact_force = 220; c->wall_contact_margin = 0.024;
    for (int w = 0; w < W; ++w) { c->walls[w].orient = w / 2; c->walls[w].hard = 1; c->walls[w].axis_pos = (w % 2 ? -2.0 : 2.0); c->walls[w].end0 = -2; c->walls[w].end1 = 2; c->walls[w].width = 0.1; }
}

static int run(int scen, int N, int A, int O, int W) {
    gmpe_config c; fill(&c, scen, N, A, O, W);
    gmpo* h = NULL;
    if (gmpo_create(&c, &h)) return 1;
    const int E = 2 * A + O, D = scen == GMPE_SCENARIO_TUBE_JULY ? 19 : 13;
    double* obs = malloc(sizeof(double) * N * A * D); int32_t* ids = malloc(4 * N * A);
    double* node = malloc(sizeof(double) * N * A * E * 8); double* adj = malloc(sizeof(double) * N * E * E);
    double* rew = malloc(sizeof(double) * N * A); uint8_t* done = malloc(N * A); double* info = malloc(sizeof(double) * N * A * GMPE_INFO_KEYS);
    uint8_t* did = malloc(N); int32_t* act = malloc(4 * N * A); int32_t* edges = malloc(4 * 2 * E * E);
    gmpo_reset(h, NULL, obs, ids, node, adj);
    unsigned s = 1;
    for (int t = 0; t < 20; ++t) {
        for (int q = 0; q < N * A; ++q) { s = s * 1664525u + 1013904223u; act[q] = (int)((s >> 16) % (unsigned)c.n_actions); }
        gmpo_step(h, act, obs, ids, node, adj, rew, done, info, did, 1);
        gmpo_update_graph(h, 0, edges, NULL, E * E);
    }
    double chk = 0; for (int q = 0; q < N * A * D; ++q) chk += obs[q];
    printf("scenario %d A=%d O=%d walls=%d checksum %.6f\n", scen, A, O, W, chk);
    free(obs); free(ids); free(node); free(adj); free(rew); free(done); free(info); free(did); free(act); free(edges);
    gmpo_destroy(h);
    return 0;
}

int main(void) {
    return run(GMPE_SCENARIO_TUBE_JULY, 5, 10, 0, 0) | run(GMPE_SCENARIO_TUBE_JULY, 3, 3, 0, 0) | run(GMPE_SCENARIO_ROT_INV, 4, 6, 0, 0) |
           run(GMPE_SCENARIO_NAVIGATION_GRAPH, 4, 6, 3, 4) | run(GMPE_SCENARIO_NAVIGATION_GRAPH, 2, 64, 0, 0);
}
