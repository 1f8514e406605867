/* gmpe.h — C ABI of the MI355X batched GraphMPE step engine (libgmpe.so).
 *
 * This is the drop-in boundary for ONE hot path of Jaroan/Contracts-MARL-AAM-Corridors: the
 * per-environment particle-world step + graph observation, batched over N environments.
 * It replaces what `GraphSubprocVecEnv` (onpolicy/envs/env_wrappers.py:959-1037) obtains from its
 * N worker processes, each of which runs `MultiAgentGraphEnv.step/reset`
 * (multiagent/environment.py:1021-1081) on `World.step` (multiagent/core.py:687-756) with the
 * scenario callbacks of multiagent/custom_scenarios/nav_metered_one_goal_graph_rotate_tube_july.py.
 *
 * Conventions
 *  - plain C: POD structs, raw pointers, sizes. No torch / C++ types cross this boundary.
 *  - every function returns 0 on success or a negative gmpe_status; gmpe_last_error() gives text.
 *    No exception crosses the ABI.
 *  - "dev" pointers are device (HBM) pointers owned by the CALLER (e.g. torch tensors'
 *    data_ptr()); "host" pointers are ordinary host memory. The handle owns only the persistent
 *    SoA world state and scratch (a few KB; plus one [N,E,E] float matrix for handles on the split
 *    big-E path, allocated by gmpe_create); step/reset never allocate.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream). step/reset are
 *    asynchronous on it; get/set_field synchronise the handle's stream themselves.
 *  - one handle per GPU; handles are not thread-safe.
 *
 * Shapes: N envs, A agents, L landmarks, O obstacles, E = A+L+O graph nodes, F = 8 node features,
 * D = observation width (19 tube_july, 13 navigation_graph, 13 rot_inv); F = 8 (7 for rot_inv and for graph_feat_type 'global').
 */
#ifndef GMPE_H
#define GMPE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMPE_ABI_VERSION 3
#define GMPE_NODE_FEATS 8          /* …_july.py:1771  [rel_vel2, rel_pos2, rel_goal2, occupied, type]; rot_inv: 7 (gmpe_node_feats) */
#define GMPE_INFO_KEYS 18          /* …_july.py:806-828 + 'individual_reward' (environment.py:1048) + 'Phase_reached' (rot_inv:835) */
#define GMPE_MAX_AGENTS 64         /* one wavefront lane per agent in the sequential-semantics pass   */
#define GMPE_MAX_ENTITIES 160
#define GMPE_MAX_WALLS 8
#define GMPE_TUBE_STRIDE 12

typedef enum gmpe_status {
    GMPE_OK = 0,
    GMPE_ERR_INVALID_ARG = -1,
    GMPE_ERR_HIP = -2,              /* a HIP runtime call failed (text in gmpe_last_error)            */
    GMPE_ERR_NO_DEVICE = -3,
    GMPE_ERR_UNSUPPORTED = -4,
    GMPE_ERR_TAPE_EXHAUSTED = -5,   /* parity mode: an env needed more uniform draws than the tape has */
    GMPE_ERR_PLACEMENT = -6         /* reset: rejection sampler hit GMPE_MAX_PLACEMENT_TRIES           */
} gmpe_status;

/* Scenarios (multiagent/custom_scenarios/<name>.py). */
typedef enum gmpe_scenario {
    GMPE_SCENARIO_NAVIGATION_GRAPH = 0, /* not shipped by the reference (train_mpe.py:72-73 default only):
                                           restated from extant blocks, see DESIGN.md §navigation_graph */
    GMPE_SCENARIO_TUBE_JULY = 1,        /* nav_metered_one_goal_graph_rotate_tube_july.py               */
    GMPE_SCENARIO_ROT_INV = 2,          /* nav_graph_metered_single_corridor_rot_inv.py: rotation-
                                           invariant 13-d obs, 7 node features, exit gate + progress reward, armed cooldown */
    GMPE_SCENARIO_TWO_PHASE = 3,        /* two_phase_graph.py: rot_inv family, 15-d obs (exit vector, heading alignment), random
                                           tube length, episode ends for an agent at the exit gate, no collision reward term  */
    GMPE_SCENARIO_THREE_PHASE = 4       /* three_phase_graph.py: two_phase + post-tube goal phase, -collision_rew per contact   */
} gmpe_scenario;

/* Dynamics (multiagent/core.py:23-26 EntityDynamicsType). */
typedef enum gmpe_dynamics {
    GMPE_DYN_DOUBLE_INTEGRATOR = 0,     /* force path: core.py:766-845, 872-964                       */
    GMPE_DYN_UNICYCLE = 1,              /* kinematic, UnicycleVehicleConfig constants                 */
    GMPE_DYN_AIR_TAXI = 2               /* kinematic: core.py:231-340, 819-826                        */
} gmpe_dynamics;

/* args.formation_type (…_july.py:192, 492-497): where random_scenario puts the landmarks (goal i belongs to agent i). */
typedef enum gmpe_formation {
    GMPE_FORMATION_POINT = 0,           /* set_landmarks_in_point (custom_scenarios/utils.py:165-193): all at exit + R(angle) @ [0, -ws/3]      */
    GMPE_FORMATION_LINE = 1,            /* set_landmarks_in_line (utils.py:77-130) called with start (-ws/2, -ws/2), end (ws/2, -ws/2):
                                           np.linspace(start, end, L) — the zero y-step takes linspace's (i / div) * delta branch             */
    GMPE_FORMATION_CIRCLE = 2           /* set_landmarks_in_circle (utils.py:231-267): centre (0, exit_y + ws/5), radius ws/3, angle i * 2 pi / L */
} gmpe_formation;

typedef struct gmpe_wall {              /* multiagent/core.py:354-373 Wall                             */
    int32_t orient;                     /* 0 = 'H' (lies along x at y = axis_pos), 1 = 'V'            */
    int32_t hard;
    double axis_pos;
    double end0, end1;
    double width;
} gmpe_wall;

/* Everything the reference reads from `args` (…_july.py:155-192,206-224,248-259,274,315,326) and
 * from multiagent/config.py, frozen into one POD. The Python host fills it (gmpe/config.py). */
typedef struct gmpe_config {
    int32_t abi_version;                /* GMPE_ABI_VERSION                                            */
    int32_t scenario;                   /* gmpe_scenario                                               */
    int32_t dynamics;                   /* gmpe_dynamics                                               */
    int32_t num_envs;                   /* N on this handle                                            */
    int32_t num_agents;                 /* A  (== num_landmarks: goal i belongs to agent i, :356)      */
    int32_t num_landmarks;              /* L                                                           */
    int32_t num_obstacles;              /* O                                                           */
    int32_t num_walls;                  /* physics only, never graph nodes                             */
    int32_t episode_length;             /* world.world_length (environment.py:264-271)                 */
    int32_t env_id_base;                /* global id of env 0 of this handle: RNG key = seed, id       */
    int32_t n_actions;                  /* 25 (5x5 motion primitives) or 5 / 9 (double integrator)     */
    int32_t collaborative;              /* shared reward (environment.py:1056-1061)                    */
    uint64_t seed;
    double world_size;
    double max_speed;                   /* args.max_speed (agent.max_speed; <=0 means None)            */
    double collision_rew, formation_rew, goal_rew;
    double min_reward, max_reward;      /* RewardWeightConfig.MIN/MAX_REWARD (config.py:135-136)       */
    /* multiagent/config.py constants of the chosen dynamics */
    double dt;
    double v_min, v_max;                /* kinematic speed clamp (core.py:309-312); DI: v_max = VX_MAX */
    double goal_thresh;                 /* DISTANCE_TO_GOAL_THRESHOLD                                  */
    double sep_dist;                    /* COLLISION_DISTANCE (= SEPARATION_DISTANCE for air_taxi)     */
    double coord_range;                 /* COORDINATION_RANGE: update_graph max_edge_dist (:242)       */
    double ang_rate_opt[5];             /* np.linspace(-ANGULAR_RATE_MAX, +, 5) (environment.py:441)   */
    double accel_opt[5];                /* np.linspace(ACCEL_MIN, ACCEL_MAX, 5)   (environment.py:440) */
    double sensitivity;                 /* 5.0 (environment.py:460-463)                                */
    double entity_size;                 /* Entity.size = 0.06 (core.py:385)                            */
    /* force path (core.py:542-548) */
    double damping, contact_force, contact_margin, wall_contact_force, wall_contact_margin;
    gmpe_wall walls[GMPE_MAX_WALLS];
    /* ---- ABI 2 ---- */
    int32_t graph_feat_type;            /* 0 'relative' (…_july.py:1694-1771, F = 8; rot_inv family F = 7); 1 'global' (…_july.py:1672-1691,
                                           rot_inv.py:1668-1687: [vel, pos, goal, type] in world coordinates, F = 7, every scenario)  */
    int32_t contact_family;             /* force path constants: 0 = multiagent/core.py:872-906 (d_min = COLLISION_DISTANCE, no force on a
                                           done side, separate wall constants); 1 = classic MPE, onpolicy/envs/mpe/core.py:273-286
                                           (d_min = size_a + size_b, every collider side gets its force, walls share the contact
                                           constants). navigation_graph only                                                      */
    double agent_size, collider_size;   /* classic: Entity.size of agents / of obstacles (mpe/core.py:44)                          */
    double agent_mass;                  /* Entity.mass (mpe/core.py:52 initial_mass = 1.0): v += F / mass * dt                     */
    double action_force_scale;          /* apply_action_force (mpe/core.py:205-214): mass * accel if accel is not None else mass   */
    /* ---- ABI 3 ---- */
    int32_t formation_type;             /* gmpe_formation: landmark placement of the tube scenarios' reset (…_july.py:492-497, rot_inv.py:493-498,
                                           two_phase_graph.py:464-469, three_phase_graph.py:459-464); navigation_graph does not read it        */
    int32_t reserved0;
} gmpe_config;

/* Persistent per-env state, addressable for checkpoint / parity injection. */
typedef enum gmpe_field {
    GMPE_F_X = 0,            /* f64 [N,A]                                                               */
    GMPE_F_Y,                /* f64 [N,A]                                                               */
    GMPE_F_S2,               /* f64 [N,A]  air_taxi/unicycle: theta        double_integrator: v_x       */
    GMPE_F_S3,               /* f64 [N,A]  air_taxi/unicycle: speed        double_integrator: v_y       */
    GMPE_F_P_DIST,           /* f64 [N,A]  odometer (core.py:315)                                       */
    GMPE_F_TIME,             /* f64 [N,A]  (core.py:316)                                                */
    GMPE_F_STATUS,           /* u8  [N,A]  agent.status (done)                                          */
    GMPE_F_PREV_PHASE,       /* i32 [N,A]  agent.previous_phase — survives resets (…_july.py:708-710)   */
    GMPE_F_PHASE_REACHED,    /* i32 [N,A]                                                               */
    GMPE_F_COOLDOWN,         /* i32 [N,A]  entry_reward_cooldown (never armed in the July file)         */
    GMPE_F_GOAL_TRACKER,     /* i32 [N,A]                                                               */
    GMPE_F_CURRENT_STEP,     /* i32 [N]                                                                 */
    GMPE_F_RNG_CTR,          /* i64 [N]    uniform draws consumed so far by this env                    */
    GMPE_F_TUBE,             /* f64 [N,12] angle, ent.xy, exit.xy, e.xy, n.xy(fp32-rounded), L, half_w, width */
    GMPE_F_LANDMARKS,        /* f64 [N,L,2]                                                             */
    GMPE_F_OBSTACLES,        /* f64 [N,O,2]                                                             */
    /* info counters (…_july.py:741-829); the int-typed ones reproduce the reference's np.full(n,-1)
       int64 arrays, into which floats are truncated on assignment */
    GMPE_F_TIMES_REQUIRED,   /* i32 [N,A] */
    GMPE_F_DISTS_TO_GOAL,    /* i32 [N,A] */
    GMPE_F_DIST_LEFT,        /* i32 [N,A] */
    GMPE_F_GOAL_REACHED,     /* i32 [N,A] */
    GMPE_F_N_AGENT_COLL,     /* i32 [N,A] */
    GMPE_F_N_OBST_COLL,      /* i32 [N,A] */
    GMPE_F_SPACING_VIOL,     /* i32 [N,A] */
    GMPE_F_STEPS_IN_CORR,    /* i32 [N,A] */
    GMPE_F_CONFORMANCE,      /* i32 [N,A] */
    GMPE_F_GOAL_MIN_TIME,    /* f64 [N,A] agent.goal_min_time (…_july.py:941-951)                       */
    GMPE_F_DELTA_SPACING,    /* f64 [N]   running sum of the delta_spacing list (:1180, :802)           */
    GMPE_F_ERROR_FLAGS,      /* i32 [N]   sticky: bit0 tape exhausted, bit1 placement gave up           */
    GMPE_F_PREV_PROJ,        /* f64 [N,A] rot_inv: prev_proj (a float32 array in the reference, :374, 1268-1276) */
    GMPE_F_COUNT
} gmpe_field;

typedef struct gmpe_handle gmpe_handle;

/* Output buffers of one step/reset, all caller-owned DEVICE memory. Any pointer may be NULL to
 * skip that output. Layout = what GraphSubprocVecEnv.step_wait stacks (env_wrappers.py:996-1004),
 * narrowed to the dtypes GraphReplayBuffer stores (onpolicy/utils/graph_buffer.py:84-114). */
typedef struct gmpe_outputs {
    float*   obs;        /* [N,A,D]                                                                    */
    int32_t* agent_id;   /* [N,A,1]                    (…_july.py:1554-1555)                           */
    float*   node_obs;   /* [N,A,E,F]  F = gmpe_node_feats (…_july.py:1584-1624, 1694-1771)            */
    float*   adj;        /* adj_compact ? [N,E,E] : [N,A,E,E]  (…_july.py:1625-1648)                   */
    float*   reward;     /* [N,A]   (step only)        (…_july.py:1105-1221)                           */
    uint8_t* done;       /* [N,A]   (step only)        (environment.py:264-271)                        */
    float*   info;       /* [N,A,GMPE_INFO_KEYS] (step only; optional) (…_july.py:741-829)             */
    int32_t  adj_compact;/* 1: write the single E×E matrix every ego shares            */
    int32_t  reserved;
    /* ---- ABI 3 ---- */
    double*  entity_table;/* [N,W] f64, W = gmpe_entity_table_width(cfg), or NULL: the per-entity state the node_obs rows of this step are a pure function of —
                            what a rank SHIPS instead of the [N,A,E,F] rows (SURVEY §8e "state + one E×E per env"); gmpe_expand_node_obs rebuilds the rows
                            bit for bit on the learner. Layout per env: x[E], y[E] of every entity (agents after the move / after a reset), then per agent
                            vox[A], voy[A] (velocity BEFORE this step's reward loop), vnx[A], vny[A] (AFTER it: re-drawn heading on a goal reach, core.py:324-333);
                            rot_inv family: + cos[A], sin[A] of the post-reward heading; two_phase_graph: + exit x, exit y (its goal node feature); last, ceil(E / 32)
                            words of this step's adjacency mask (bit k of word k / 32 = node k's rows / columns are zeroed, …_july.py:1627-1648), 32 bits per double   */
} gmpe_outputs;

int gmpe_abi_version(void);
const char* gmpe_last_error(void);

/* Observation width D, node feature count F and node count E for a config (no device needed). */
int gmpe_obs_dim(const gmpe_config* cfg);
int gmpe_node_feats(const gmpe_config* cfg);   /* 8; 7 for the rot_inv family and for graph_feat_type = 1 */
int gmpe_num_entities(const gmpe_config* cfg);
int gmpe_entity_table_width(const gmpe_config* cfg);   /* W of gmpe_outputs.entity_table: 2E + 4A (+ 2A rot_inv family) (+ 2 two_phase_graph) + ceil(E / 32) doubles per env */

/* Create the engine on HIP device `device`. Replaces N x `GraphMPEEnv(args)` + `env.seed(seed +
 * rank*1000)` (multiagent/MPE_env.py:56-84, onpolicy/scripts/train_mpe.py:21-43). */
int gmpe_create(const gmpe_config* cfg, int device, gmpe_handle** out);
int gmpe_destroy(gmpe_handle* h);

/* Parity mode: replay the reference's np.random draws. `tape_dev` is DEVICE memory, f64
 * [N, len_per_env] of [0,1) samples; draw k of env n is tape[n*len+k]. NULL returns to the
 * counter-based Philox4x32-10 stream keyed by (seed, env_id_base+n, k). */
int gmpe_set_rng_tape(gmpe_handle* h, const double* tape_dev, int64_t len_per_env);

/* Replaces GraphSubprocVecEnv.reset (env_wrappers.py:1006-1013 → environment.py:1066-1081 →
 * …_july.py:339-420). `env_mask_dev` (u8 [N], device) selects envs; NULL = all. */
int gmpe_reset(gmpe_handle* h, const uint8_t* env_mask_dev, const gmpe_outputs* out, void* stream);

/* Replaces GraphSubprocVecEnv.step (env_wrappers.py:991-1004 → graphworker :851-873 →
 * environment.py:1021-1063), including the worker's auto-reset: envs whose agents are all done
 * return their POST-reset obs/agent_id/node_obs/adj with the terminal reward/done.
 * `action_idx_dev`: i32 [N,A] discrete action index (argmax of the runner's one-hot,
 * environment.py:446). */
int gmpe_step(gmpe_handle* h, const int32_t* action_idx_dev, const gmpe_outputs* out, void* stream);

/* The same step for the envs [env_lo, env_hi) only; `action_idx_dev` and `out` are the WHOLE-batch arrays (rows outside the range are neither
 * read nor written). Envs are independent (one OS process each in the reference, env_wrappers.py:968-975), so ranges may be stepped on different
 * streams at different times — e.g. a runner that double-buffers two halves of the batch: while the policy works on one half's observations the
 * other half steps, and one half's latency chain runs under the other half's store drain. Not on the split big-E path (GMPE_ERR_UNSUPPORTED). */
int gmpe_step_envs(gmpe_handle* h, const int32_t* action_idx_dev, const gmpe_outputs* out, int32_t env_lo, int32_t env_hi, void* stream);
/* `num_steps` steps of `parts` (1..4) equal env ranges, each range on a side stream of its own (forked from / joined into `stream` once per call): the launch
 * shape of a runner that double-buffers ranges of the batch, with the open-loop action source of gmpe_step_many. Same results as gmpe_step_many. */
int gmpe_step_many_envs(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets, const gmpe_outputs* out,
                        int32_t parts, void* stream);

/* `num_steps` consecutive steps enqueued by one call (no host round trip between steps; one launch, see gmpe_rollout_steps): step k uses
 * action set k % num_action_sets of `actions_dev` (i32 [num_action_sets, N, A]). Outputs are overwritten
 * by every step (same buffers), exactly as a host loop over gmpe_step would. Used for open-loop rollouts
 * (random-action benchmarking, scripted policies). */
int gmpe_step_many(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets,
                   const gmpe_outputs* out, void* stream);
/* The same steps as ONE KERNEL LAUNCH PER STEP (the closed-loop launch shape, enqueued without host round trips; replays a graph
 * recorded by gmpe_step_many_prepare when there is one). gmpe_step_many falls back to this on the split big-E path, where the steps' chunk
 * pipelines are chained (gmpe_tuning.xstep): the side streams fork before the first step and join the caller's stream after the last one. */
int gmpe_step_many_launches(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets,
                            const gmpe_outputs* out, void* stream);

/* Open-loop rollout in ONE launch (round 2): the K steps run inside a persistent kernel — each workgroup keeps the state of its
 * envs in LDS / registers from step to step (no reload, one write-back at the end) and the graph stores of step k drain under
 * step k+1's arithmetic. Results are bit-identical to K calls of gmpe_step (auto-resets included).
 * This is what the reference's collect loop does with a fixed action source: graph_mpe_runner.py:57-103 (`for step in
 * range(self.episode_length)`: envs.step -> GraphReplayBuffer.insert, onpolicy/utils/graph_buffer.py:168-251), minus the policy.
 * Output placement: step k writes slot (first_slot + k) % num_slots; slot s of an output lies `stride_*` ELEMENTS after slot 0
 * (the pointers in `slot0`). num_slots = 1 with zero strides = "every step overwrites the same buffers" (gmpe_step_many).
 * `masks` / `active_masks` (optional, f32 [slots][N,A], stride_masks apart) receive GraphReplayBuffer.insert's mask rules for the
 * step (see gmpe_masks_from_dones). Not available for handles on the split big-E path (gmpe_tuning.split): returns
 * GMPE_ERR_UNSUPPORTED there — use gmpe_step_many.
 * Performance note (round 3, profiles/r03_notes.md): with one slot every persistent workgroup rewrites its own output block each step; where a step's
 * outputs are far larger than the 256 MiB Infinity Cache that is markedly slower than slot-per-step storage (c4, 6 GB per step: 1122 us per step with one
 * slot, 929 with 4 slots, 868 with 26) — give big configurations the [T, ...] storage a rollout buffer has anyway. Outputs that fit the cache (c2 / c3:
 * 98 MB per step) are faster with one slot (absorbed by the cache) but are then not paid in DRAM writes. */
typedef struct gmpe_rollout {
    int32_t num_steps;          /* K >= 1                                                                  */
    int32_t num_action_sets;    /* S: step k uses action set k % S of actions_dev (i32 [S,N,A])            */
    int32_t num_slots;          /* >= 1                                                                    */
    int32_t first_slot;         /* in [0, num_slots)                                                       */
    int64_t stride_obs, stride_agent_id, stride_node_obs, stride_adj, stride_reward, stride_done, stride_info, stride_masks;
    float*  masks;              /* slot 0 of the masks, or NULL                                            */
    float*  active_masks;       /* slot 0 of the active_masks, or NULL                                     */
    int64_t stride_entity_table;/* ABI 3: elements (doubles) between consecutive slots of gmpe_outputs.entity_table */
} gmpe_rollout;
int gmpe_rollout_steps(gmpe_handle* h, const int32_t* actions_dev, const gmpe_rollout* plan, const gmpe_outputs* slot0, void* stream);

/* Optional: record the `num_steps` launches of gmpe_step_many(actions_dev, num_steps, num_action_sets, out) into a
 * hipGraph once (capture on a private stream + instantiate: milliseconds, not on the step path). Later
 * gmpe_step_many calls with the SAME pointers and counts replay it with one hipGraphLaunch on the caller's stream
 * (kernel-to-kernel dispatch overhead 3.6 -> 1.6 us at this launch shape, profiles/README.md); any other call takes
 * the plain launch loop. The action / output BUFFERS are baked in, their contents are read at replay time.
 * Since round 2 gmpe_step_many runs the rollout kernel above by default (gmpe_tuning.roll); a prepared graph is what it falls back to
 * when that is off (GMPE_ROLL=0) or unavailable (split path, per-launch timing). */
int gmpe_step_many_prepare(gmpe_handle* h, const int32_t* actions_dev, int32_t num_steps, int32_t num_action_sets,
                           const gmpe_outputs* out);

/* Same, taking the runner's float one-hot [N,A,n_actions] (graph_mpe_runner.py:375-377); the
 * argmax (np.argmax: first maximum) is fused into the step kernel. */
int gmpe_step_onehot(gmpe_handle* h, const float* onehot_dev, const gmpe_outputs* out, void* stream);

/* Safety-filter hook slot (multiagent/core.py:505-534, 692-736): in the reference `World.step` hands every agent's raw control
 * [omega, accel] (or [a_x, a_y]) to `safety_handle.apply_safety_filter` between `get_action()` and `update_agent_state` and
 * integrates the FILTERED control. The HJ / CBF filter itself is out of scope (value-function data and jax / cvxpy are absent); this
 * keeps its place in the step: when `ctrl_dev` (f64 [N,A,2], device, caller-owned) is set, agent (n,a) integrates ctrl_dev[n,a,:]
 * instead of its decoded action wherever `use_dev` (u8 [N,A]; NULL = everywhere) is non-zero — the `filtered` flag of the reference.
 * Units: the decoded control AFTER the x5 sensitivity (environment.py:460-463), i.e. what `agent.action.u` holds. An external
 * filter kernel reads the state through gmpe_field_device_ptr and runs on the same stream before gmpe_step. NULL removes the hook. */
int gmpe_set_control_override(gmpe_handle* h, const double* ctrl_dev, const uint8_t* use_dev);
/* Device pointer of a state field (layout in gmpe_field), valid for the life of the handle; for on-device consumers such as the
 * filter above. Reading it is ordered with the engine's launches only through the stream they share. */
int gmpe_field_device_ptr(gmpe_handle* h, int field, void** ptr_out);

/* Host <-> engine state copies (whole field, `bytes` must equal the field's size). */
int gmpe_field_bytes(const gmpe_handle* h, int field, size_t* bytes);
int gmpe_get_field(gmpe_handle* h, int field, void* host_dst, size_t bytes);
int gmpe_set_field(gmpe_handle* h, int field, const void* host_src, size_t bytes);

/* Learner-side edge set of onpolicy/algorithms/utils/gnn_new.py:329-358 (process_adj):
 * mask = (adj < max_edge_dist) & (adj > 0) on fp32, edges in (batch,row,col) lexicographic order,
 * node ids offset by batch*E. adj_dev: f32 [B,E,E]. Outputs: edge_index i32 [2,cap] (row 0 = src,
 * row 1 = dst), edge_attr f32 [cap], n_edges i32 [1] (device). If more than `cap` edges exist only
 * the first `cap` are written and n_edges still holds the true count. */
int gmpe_edges_from_adj(gmpe_handle* h, const float* adj_dev, int32_t batch, int32_t num_nodes,
                        float max_edge_dist, int32_t inclusive, int32_t* edge_index_dev,
                        float* edge_attr_dev, int32_t cap, int32_t* n_edges_dev, void* stream);

/* The same edge set computed from the COMPACT adjacency the engine writes with gmpe_outputs.adj_compact (adj_compact_dev: f32
 * [num_envs,E,E]): the `copies` (= A) per-agent graphs of an env are identical up to the id shift, so each matrix is read once and
 * the A copies of its edge list are emitted — output identical to gmpe_edges_from_adj on the materialised [num_envs*copies,E,E]
 * tensor, batch b = env*copies + copy. index64 != 0: edge_index is int64 [2,cap] (what torch.nonzero / PyG message passing use,
 * gnn_new.py:329-358), else int32. n_edges saturates at INT32_MAX. */
int gmpe_edges_from_adj_compact(gmpe_handle* h, const float* adj_compact_dev, int32_t num_envs, int32_t copies, int32_t num_nodes,
                                float max_edge_dist, int32_t inclusive, int32_t index64, void* edge_index_dev, float* edge_attr_dev,
                                int64_t cap, int32_t* n_edges_dev, void* stream);

/* Learner side of the compact rollout gather (replaces what GraphSubprocVecEnv.step_wait receives pickled from its workers, onpolicy/envs/env_wrappers.py:996-1004, for
 * the node features consumed by GraphReplayBuffer.insert, onpolicy/utils/graph_buffer.py:168-251): rebuild node_obs rows from entity tables
 * (gmpe_outputs.entity_table) with the engine's own arithmetic — same operations in the same order, -ffp-contract=off — so the result is BIT-IDENTICAL to the
 * node_obs the engine would have written (_get_entity_feat_relative …_july.py:1694-1771 / rot_inv.py:1690-1766, _get_entity_feat_global …_july.py:1672-1691,
 * including the ordered-visibility rule for re-drawn velocities). Needs no handle (the learner rank may own no envs): `cfg` supplies scenario, feature type and sizes.
 *   table_dev     f64 [num_blocks, envs_per_block, W]           (e.g. one rank's [T+1, N, W] rollout section)
 *   node_obs_dev  f32 [num_blocks, out_envs_per_block, A, E, F]  block t, env n -> out env out_env_offset + n (a rank's env range inside the global batch) */
int gmpe_expand_node_obs(const gmpe_config* cfg, int device, const double* table_dev, int64_t num_blocks, int64_t envs_per_block,
                         float* node_obs_dev, int64_t out_envs_per_block, int64_t out_env_offset, void* stream);
/* The same for the adjacency: the E x E matrix of every env-step from its entity table (positions + mask words) — f32(sqrt(dx^2 + dy^2)) with the engine's own
 * expression (World.calculate_distances, core.py:600-624), masked rows / columns zeroed (…_july.py:1627-1648) — bit-identical to gmpe_outputs.adj, so a rank need not
 * ship the matrix at all. adj_dev: f32 [num_blocks, out_envs_per_block, copies, E, E]; copies = 1: the compact form, copies = A: the materialised [.., A, E, E]. */
int gmpe_expand_adj(const gmpe_config* cfg, int device, const double* table_dev, int64_t num_blocks, int64_t envs_per_block,
                    float* adj_dev, int64_t out_envs_per_block, int64_t out_env_offset, int32_t copies, void* stream);

/* Rollout-buffer masks from a step's dones (GraphReplayBuffer.insert: onpolicy/utils/graph_buffer.py:223-251 with the runner's
 * rules graph_mpe_runner.py:85-90, 395-405): masks f32 [N,A] = 0 where done; active_masks f32 [N,A] = 0 where done unless all agents of
 * the env are done. Either output may be NULL. */
int gmpe_masks_from_dones(gmpe_handle* h, const uint8_t* done_dev, float* masks_dev, float* active_masks_dev, void* stream);

/* What gmpe_create chose for this handle (recorded by bench.py next to every number). Environment variables override the heuristics —
 * GMPE_G / GMPE_BLOCK (step tile shape), GMPE_GROLL (rollout tile shape), GMPE_AP=0 (run-time-size instead of exact-size kernels),
 * GMPE_NT / GMPE_ROLLNT (nontemporal graph stores of step / rollout launches), GMPE_SPEC (wave specialisation), GMPE_SPLIT / GMPE_CHUNKS
 * / GMPE_RAMP / GMPE_AHEAD / GMPE_XSTEP (split big-E path, its chunk count, quarter + half first chunks, run-ahead bound, chained steps), GMPE_ROLL (gmpe_step_many as one rollout launch), GMPE_FUSE — none of them changes results
 * (tests/test_gpu_instantiations.py, tests/test_gpu_rollout_kernel.py). */
typedef struct gmpe_tuning {
    int32_t G;                  /* envs per workgroup (tile)                                               */
    int32_t block;              /* threads per workgroup: 64, 128 or 256                                   */
    int32_t nt;                 /* 1: nontemporal graph stores (launch output > Infinity Cache)            */
    int32_t spec;               /* 1: wave-specialised tiles                                               */
    int32_t split;              /* 1: big-E path — fused kernel writes the compact [N,E,E] matrix into the handle's scratch,
                                      k_adj_expand materialises the A ego copies                           */
    int32_t roll;               /* 1: gmpe_step_many runs the persistent rollout kernel                    */
    int32_t ap;                 /* exact-size instantiation (0: run-time sizes)                            */
    int32_t lds_bytes;          /* dynamic LDS per tile of the step kernels                                */
    int32_t diag_build;         /* 1: library built with -DGMPE_DIAG (ablations honoured): never for results */
    int32_t G_roll, block_roll; /* tile shape of the rollout kernel (its own register budget, hence its own residency)  */
    int32_t chunks, ahead;      /* split path: env chunks per step; how many chunks the fused kernel may run ahead of the expansion (0: unbounded) */
    int32_t xstep;              /* split path: gmpe_step_many chains the steps' chunk pipelines (no join between open-loop steps)          */
    int32_t chunks_x, ahead_x;  /* ... with this chunking / run-ahead bound                                                                */
    int32_t lds_bytes_roll;     /* dynamic LDS per tile of the rollout kernel (G_roll envs; + the pair-force buffer of fused navigation_graph rollouts)     */
} gmpe_tuning;
int gmpe_get_tuning(const gmpe_handle* h, gmpe_tuning* out);

/* Timing hooks used by bench.py: HIP events on the handle's launch stream around every step
 * kernel, so the dominant kernel's duration is measured live (not via torch's current stream). */
int gmpe_timing_enable(gmpe_handle* h, int32_t enable);          /* one event pair around EVERY launch (perturbs back-to-back launches by ~5 us each) */
int gmpe_timing_read(gmpe_handle* h, double* total_ms, int64_t* launches, int32_t reset_counters);
/* one event pair around a REGION of launches: mark(0) before the first, mark(1) after the last, both on
 * the launch stream; region_ms synchronises on the second event. */
int gmpe_timing_mark(gmpe_handle* h, int32_t which, void* stream);
int gmpe_timing_region_ms(gmpe_handle* h, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* GMPE_H */
