"""Host-side constants and the POD config handed through the C ABI (include/gmpe.h: gmpe_config).

Mirrors multiagent/config.py (class-level constants per dynamics model) and the `args` fields the
scenario reads in make_world (nav_metered_one_goal_graph_rotate_tube_july.py:155-192, 206-224,
248-259, 274, 315, 326).
"""
import ctypes as C
import math

import numpy as np

ABI_VERSION = 3
NODE_FEATS = 8
INFO_KEYS = ["individual_reward", "Dist_to_goal", "Time_req_to_goal", "Num_agent_collisions",
             "Num_obst_collisions", "Distance_mean", "Distance_variance", "Mean_by_variance",
             "Dists_traveled", "Time_taken", "Time_mean", "Time_stddev", "Time_mean_by_stddev",
             "Conformance", "Delta_spacing", "Spacing_violations", "Min_time_to_goal", "Phase_reached"]
MAX_WALLS = 8
TUBE_STRIDE = 12

SCENARIO_NAVIGATION_GRAPH = 0
SCENARIO_TUBE_JULY = 1
SCENARIO_ROT_INV = 2
SCENARIO_TWO_PHASE = 3
SCENARIO_THREE_PHASE = 4
ROT_FAMILY = (SCENARIO_ROT_INV, SCENARIO_TWO_PHASE, SCENARIO_THREE_PHASE)      # rotated-frame float32 features, F = 7
SCENARIOS = {
    "navigation_graph": SCENARIO_NAVIGATION_GRAPH,
    "nav_metered_one_goal_graph_rotate_tube_july": SCENARIO_TUBE_JULY,
    "nav_graph_metered_single_corridor_rot_inv": SCENARIO_ROT_INV,
    "two_phase_graph": SCENARIO_TWO_PHASE,
    "three_phase_graph": SCENARIO_THREE_PHASE,
}
FORMATIONS = {"point": 0, "line": 1, "circle": 2}      # gmpe_formation (…_july.py:492-497; custom_scenarios/utils.py:77-130, 165-193, 231-267)
DYN_DOUBLE_INTEGRATOR, DYN_UNICYCLE, DYN_AIR_TAXI = 0, 1, 2
DYNAMICS = {"double_integrator": DYN_DOUBLE_INTEGRATOR, "unicycle_vehicle": DYN_UNICYCLE,
            "air_taxi": DYN_AIR_TAXI}


class AirTaxiConfig:                      # multiagent/config.py:4-33
    V_MIN = 60 * 0.514444 * 0.001
    V_MAX = 175 * 0.514444 * 0.001
    ACCEL_MIN = -0.001
    ACCEL_MAX = 0.002
    ANGULAR_RATE_MAX = 0.1
    DT = 1.0
    DISTANCE_TO_GOAL_THRESHOLD = 0.35
    SEPARATION_DISTANCE = 1500 * 0.0003048
    COLLISION_DISTANCE = SEPARATION_DISTANCE
    COORDINATION_RANGE = 3 * 1.60934


class UnicycleVehicleConfig:              # multiagent/config.py:36-53
    V_MIN = 0.4
    V_MAX = 0.75
    ACCEL_MIN = -0.5
    ACCEL_MAX = 0.5
    ANGULAR_RATE_MAX = 0.5
    DT = 0.1
    DISTANCE_TO_GOAL_THRESHOLD = 0.2
    COLLISION_DISTANCE = 0.4
    COORDINATION_RANGE = 5               # reference spells it COMMUNICATION_RANGE (config.py:53)


class DoubleIntegratorConfig:             # multiagent/config.py:94-116
    V_MIN = 0.0
    V_MAX = 1.0                          # VX_MAX: DoubleIntegratorXYState.max_speed (core.py:165)
    ACCEL_MIN = -1.0
    ACCEL_MAX = 1.0
    ANGULAR_RATE_MAX = 0.0
    DT = 0.1
    DISTANCE_TO_GOAL_THRESHOLD = 0.2
    COLLISION_DISTANCE = 0.5
    COORDINATION_RANGE = 5               # COMMUNICATION_RANGE (config.py:114)


_DYN_CFG = {DYN_AIR_TAXI: AirTaxiConfig, DYN_UNICYCLE: UnicycleVehicleConfig,
            DYN_DOUBLE_INTEGRATOR: DoubleIntegratorConfig}


class GmpeWall(C.Structure):
    _fields_ = [("orient", C.c_int32), ("hard", C.c_int32), ("axis_pos", C.c_double),
                ("end0", C.c_double), ("end1", C.c_double), ("width", C.c_double)]


class GmpeConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("scenario", C.c_int32), ("dynamics", C.c_int32),
        ("num_envs", C.c_int32), ("num_agents", C.c_int32), ("num_landmarks", C.c_int32),
        ("num_obstacles", C.c_int32), ("num_walls", C.c_int32), ("episode_length", C.c_int32),
        ("env_id_base", C.c_int32), ("n_actions", C.c_int32), ("collaborative", C.c_int32),
        ("seed", C.c_uint64),
        ("world_size", C.c_double), ("max_speed", C.c_double),
        ("collision_rew", C.c_double), ("formation_rew", C.c_double), ("goal_rew", C.c_double),
        ("min_reward", C.c_double), ("max_reward", C.c_double),
        ("dt", C.c_double), ("v_min", C.c_double), ("v_max", C.c_double),
        ("goal_thresh", C.c_double), ("sep_dist", C.c_double), ("coord_range", C.c_double),
        ("ang_rate_opt", C.c_double * 5), ("accel_opt", C.c_double * 5),
        ("sensitivity", C.c_double), ("entity_size", C.c_double),
        ("damping", C.c_double), ("contact_force", C.c_double), ("contact_margin", C.c_double),
        ("wall_contact_force", C.c_double), ("wall_contact_margin", C.c_double),
        ("walls", GmpeWall * MAX_WALLS),
        ("graph_feat_type", C.c_int32), ("contact_family", C.c_int32),
        ("agent_size", C.c_double), ("collider_size", C.c_double), ("agent_mass", C.c_double), ("action_force_scale", C.c_double),
        ("formation_type", C.c_int32), ("reserved0", C.c_int32),
    ]

    # ---- derived sizes (gmpe_obs_dim / gmpe_num_entities)
    @property
    def num_entities(self):
        return self.num_agents + self.num_landmarks + self.num_obstacles

    @property
    def obs_dim(self):
        return 19 if self.scenario == SCENARIO_TUBE_JULY else (15 if self.scenario in (SCENARIO_TWO_PHASE, SCENARIO_THREE_PHASE) else 13)

    @property
    def entity_table_width(self):
        """gmpe_entity_table_width: doubles per env of the entity table (x[E], y[E], vox / voy / vnx / vny [A], + cos / sin [A] rot_inv family, + exit two_phase,
        + ceil(E / 32) adjacency-mask words)."""
        return (2 * self.num_entities + 4 * self.num_agents + (2 * self.num_agents if self.scenario in ROT_FAMILY else 0)
                + (2 if self.scenario == SCENARIO_TWO_PHASE else 0) + (self.num_entities + 31) // 32)

    @property
    def node_feats(self):
        return 7 if (self.scenario in ROT_FAMILY or self.graph_feat_type == 1) else NODE_FEATS


def default_walls(world_size, num_walls):
    """SURVEY.md §8(d) wall set for navigation_graph: H at y=±ws/2, V at x=±ws/2, width 0.1."""
    h = world_size / 2.0
    spec = [(0, +h), (0, -h), (1, +h), (1, -h)]
    return [dict(orient=o, axis_pos=a, end0=-h, end1=+h, width=0.1, hard=1) for o, a in spec[:num_walls]]


def make_config(scenario_name="nav_metered_one_goal_graph_rotate_tube_july", dynamics_type=None,
                num_envs=1, num_agents=3, num_landmarks=None, num_obstacles=0, num_walls=0,
                world_size=4.0, episode_length=25, max_speed=2.0, collision_rew=5.0,
                formation_rew=1.0, goal_rew=5.0, collaborative=False, total_actions=5, seed=1,
                env_id_base=0, walls=None, graph_feat_type="relative", contact_family="multiagent", agent_size=None, collider_size=None,
                agent_mass=1.0, agent_accel=None, formation_type="point"):
    """Build the POD from keyword arguments with the reference's defaults.

    formation_type: 'point' | 'line' | 'circle' — landmark placement at reset of the tube scenarios (…_july.py:492-497; navigation_graph
    places its goals at random and does not read it).
    graph_feat_type: 'relative' (default) or 'global' (…_july.py:1672-1691; July / navigation_graph only).
    contact_family (navigation_graph only): 'multiagent' = multiagent/core.py:542-548, 872-906 (contact 300 / 0.02, walls 220 / 0.024,
    d_min = COLLISION_DISTANCE, done sides get no agent-agent force) or 'classic' = onpolicy/envs/mpe/core.py:125-130, 273-286
    (contact 100 / 1e-3 for entities AND walls, d_min = size_a + size_b with agent_size / collider_size, v += F / agent_mass * dt,
    action force mass * agent_accel (or mass), sensitivity = agent_accel or 5.0 as in the classic _set_action)."""
    if scenario_name not in SCENARIOS:
        raise NotImplementedError("scenario %r is not built by this engine (have: %s)"
                                  % (scenario_name, sorted(SCENARIOS)))
    scen = SCENARIOS[scenario_name]
    if dynamics_type is None:
        dynamics_type = "double_integrator" if scen == SCENARIO_NAVIGATION_GRAPH else "air_taxi"
    if dynamics_type not in DYNAMICS:
        raise NotImplementedError("dynamics_type %r" % (dynamics_type,))
    dyn = DYNAMICS[dynamics_type]
    if scen != SCENARIO_NAVIGATION_GRAPH and dyn == DYN_DOUBLE_INTEGRATOR:
        raise NotImplementedError("the tube scenarios are kinematic (air_taxi / unicycle_vehicle)")
    if scen == SCENARIO_NAVIGATION_GRAPH and dyn != DYN_DOUBLE_INTEGRATOR:
        raise NotImplementedError("navigation_graph uses the double_integrator force path")
    dc = _DYN_CFG[dyn]
    c = GmpeConfig()
    c.abi_version = ABI_VERSION
    c.scenario, c.dynamics = scen, dyn
    c.num_envs, c.num_agents = int(num_envs), int(num_agents)
    c.num_landmarks = int(num_agents if num_landmarks is None else num_landmarks)
    c.num_obstacles, c.num_walls = int(num_obstacles), int(num_walls)
    c.episode_length, c.env_id_base = int(episode_length), int(env_id_base)
    c.collaborative = int(bool(collaborative))
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.world_size = float(world_size)
    c.max_speed = float(max_speed) if max_speed is not None else -1.0
    c.collision_rew, c.formation_rew, c.goal_rew = float(collision_rew), float(formation_rew), float(goal_rew)
    c.min_reward, c.max_reward = -40.0, 50.0          # RewardWeightConfig (config.py:135-136)
    c.dt, c.v_min, c.v_max = dc.DT, dc.V_MIN, dc.V_MAX
    c.goal_thresh, c.sep_dist = dc.DISTANCE_TO_GOAL_THRESHOLD, dc.COLLISION_DISTANCE
    c.coord_range = dc.COORDINATION_RANGE
    w = np.linspace(-dc.ANGULAR_RATE_MAX, dc.ANGULAR_RATE_MAX, 5)      # environment.py:441
    a = np.linspace(dc.ACCEL_MIN, dc.ACCEL_MAX, 5)                     # environment.py:440
    for i in range(5):
        c.ang_rate_opt[i] = float(w[i])
        c.accel_opt[i] = float(a[i])
    c.n_actions = 25 if dyn != DYN_DOUBLE_INTEGRATOR else int(total_actions)
    if c.n_actions not in (5, 9, 25):
        raise NotImplementedError("total_actions=%d" % c.n_actions)
    c.sensitivity = 5.0                                # environment.py:460 (agent.accel is None)
    c.entity_size = 0.06                               # core.py:385
    c.damping, c.contact_force, c.contact_margin = 0.25, 3e2, 2e-2     # core.py:542-547
    c.wall_contact_force, c.wall_contact_margin = 2.2e2, 2.4e-2        # core.py:545, 548
    if graph_feat_type not in ("relative", "global"):
        raise NotImplementedError("graph_feat_type %r" % (graph_feat_type,))
    c.graph_feat_type = 1 if graph_feat_type == "global" else 0
    if contact_family not in ("multiagent", "classic"):
        raise NotImplementedError("contact_family %r" % (contact_family,))
    c.contact_family = 0
    c.agent_size = float(c.entity_size if agent_size is None else agent_size)
    c.collider_size = float(c.entity_size if collider_size is None else collider_size)
    c.agent_mass, c.action_force_scale = 1.0, 1.0
    if contact_family == "classic":
        if scen != SCENARIO_NAVIGATION_GRAPH:
            raise NotImplementedError("contact_family='classic' applies to the force path (navigation_graph)")
        c.contact_family = 1
        c.contact_force, c.contact_margin = 1e2, 1e-3                  # onpolicy/envs/mpe/core.py:128-130
        c.wall_contact_force, c.wall_contact_margin = c.contact_force, c.contact_margin   # get_wall_collision_force uses the same pair (:330-332)
        c.agent_mass = float(agent_mass)
        c.action_force_scale = c.agent_mass * float(agent_accel) if agent_accel is not None else c.agent_mass   # :211-213
        if agent_accel is not None:
            c.sensitivity = float(agent_accel)
    if formation_type not in FORMATIONS:
        raise NotImplementedError("formation_type %r (have: %s)" % (formation_type, sorted(FORMATIONS)))
    c.formation_type = FORMATIONS[formation_type]
    if walls is None:
        walls = default_walls(c.world_size, c.num_walls)
    if len(walls) != c.num_walls or c.num_walls > MAX_WALLS:
        raise ValueError("walls list does not match num_walls (max %d)" % MAX_WALLS)
    for i, wl in enumerate(walls):
        c.walls[i].orient = int(wl["orient"]); c.walls[i].hard = int(wl.get("hard", 1))
        c.walls[i].axis_pos = float(wl["axis_pos"]); c.walls[i].end0 = float(wl["end0"])
        c.walls[i].end1 = float(wl["end1"]); c.walls[i].width = float(wl["width"])
    return c


def config_from_args(args, num_envs=None, env_id_base=0):
    """`args` is the Namespace GraphMPEEnv(args) takes (multiagent/MPE_env.py:56-84)."""
    g = lambda k, d=None: getattr(args, k, d)
    if g("num_scripted_agents", 0):
        raise NotImplementedError("scripted agents are not supported")
    if not g("discrete_action", True):
        raise NotImplementedError("only the discrete action space is supported")
    return make_config(
        scenario_name=g("scenario_name"), dynamics_type=g("dynamics_type"),
        num_envs=num_envs if num_envs is not None else g("n_rollout_threads", 1),
        num_agents=g("num_agents"), num_landmarks=g("num_landmarks", g("num_agents")),
        num_obstacles=g("num_obstacles", 0), num_walls=g("num_walls", 0),
        world_size=g("world_size", 2), episode_length=g("episode_length", 25),
        max_speed=g("max_speed", 2), collision_rew=g("collision_rew", 5),
        formation_rew=g("formation_rew", 1), goal_rew=g("goal_rew", 5),
        collaborative=g("collaborative", False), total_actions=g("total_actions", 5),
        seed=g("seed", 1), env_id_base=env_id_base, graph_feat_type=g("graph_feat_type", "relative"),
        formation_type=g("formation_type", "point"))


# Field table of include/gmpe.h (gmpe_field): name -> (id, dtype, shape-fn(cfg))
def _na(c): return (c.num_envs, c.num_agents)
def _n(c): return (c.num_envs,)

FIELDS = {}
for _i, (_name, _dt, _shape) in enumerate([
        ("x", np.float64, _na), ("y", np.float64, _na), ("s2", np.float64, _na), ("s3", np.float64, _na),
        ("p_dist", np.float64, _na), ("time", np.float64, _na), ("status", np.uint8, _na),
        ("prev_phase", np.int32, _na), ("phase_reached", np.int32, _na), ("cooldown", np.int32, _na),
        ("goal_tracker", np.int32, _na), ("current_step", np.int32, _n), ("rng_ctr", np.int64, _n),
        ("tube", np.float64, lambda c: (c.num_envs, TUBE_STRIDE)),
        ("landmarks", np.float64, lambda c: (c.num_envs, c.num_landmarks, 2)),
        ("obstacles", np.float64, lambda c: (c.num_envs, c.num_obstacles, 2)),
        ("times_required", np.int32, _na), ("dists_to_goal", np.int32, _na), ("dist_left", np.int32, _na),
        ("goal_reached", np.int32, _na), ("n_agent_coll", np.int32, _na), ("n_obst_coll", np.int32, _na),
        ("spacing_viol", np.int32, _na), ("steps_in_corr", np.int32, _na), ("conformance", np.int32, _na),
        ("goal_min_time", np.float64, _na), ("delta_spacing", np.float64, _n), ("error_flags", np.int32, _n),
        ("prev_proj", np.float64, _na)]):
    FIELDS[_name] = (_i, np.dtype(_dt), _shape)


def algorithmic_bytes_per_env_step(cfg, adj_compact=False):
    """B = 4*A*(E^2 + E*F + D + 2) + A + 4*A + 2*A*S, S = 48 B/agent. With the compact adjacency the engine
    writes ONE E x E matrix per env instead of A copies, and only that is counted. (The optional `info` output, 72 B/agent, is
    never counted.)"""
    A, E, F, D = cfg.num_agents, cfg.num_entities, cfg.node_feats, cfg.obs_dim
    adj = E * E if adj_compact else A * E * E
    return 4 * (adj + A * (E * F + D + 2)) + A + 4 * A + 2 * A * 48
