"""Generate the golden vectors under tests/golden/ by RUNNING the reference in this container.

    python tests/golden/make_fixtures.py            # writes tests/golden/*.npz

Vectors (data only — inputs and the reference's outputs):
  rotinv_A{3,6,10}_s{seed}.npz  the same for nav_graph_metered_single_corridor_rot_inv (float32 obs, 7 node features).
  julyglobal_A{3,6}_s{seed}.npz  the July rollouts with graph_feat_type='global' (7 node features in world coordinates).
  {july,rotinv,twophase,threephase}{line,circle}_A*_s{seed}_guided.npz  guided rollouts with formation_type 'line' / 'circle' (distinct landmarks).
  july_A{3,10}_s{seed}.npz   end-to-end rollouts of MultiAgentGraphEnv (July tube scenario, air_taxi),
                             driven like graphworker does (env_wrappers.py:851-873: step, auto-reset
                             when all agents are done), with the uniform-sample tape that replays the
                             reference's np.random draws.
  rk45_airtaxi.npz           AirTaxiXYState.update_state (core.py:300-316, scipy RK45) on random
                             states x all 25 controls.
  force_classic.npz          onpolicy/envs/mpe/core.py World.step (live force path) on random worlds.
  force_di.npz               multiagent/core.py dead-code force methods called in the order
                             calculate_distances -> apply_action_force -> apply_environment_force ->
                             integrate_state on a DoubleIntegrator world.
  misc.npz                   linspace control tables, config constants.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

INFO_KEYS = ["individual_reward", "Dist_to_goal", "Time_req_to_goal", "Num_agent_collisions",
             "Num_obst_collisions", "Distance_mean", "Distance_variance", "Mean_by_variance",
             "Dists_traveled", "Time_taken", "Time_mean", "Time_stddev", "Time_mean_by_stddev",
             "Conformance", "Delta_spacing", "Spacing_violations", "Min_time_to_goal"]

STATE_KEYS = ["x", "y", "theta", "speed", "status", "prev_phase", "phase_reached", "goal_tracker",
              "p_dist", "time", "current_step", "cooldown", "prev_proj"]
TUBE_KEYS = ["tube_angle", "entrance", "exit", "tube_e", "tube_n", "tube_L", "half_w", "width",
             "landmarks"]


def pad_edges(edge_list, m_max):
    out = -np.ones((2, m_max), dtype=np.int32)
    out[:, :edge_list.shape[1]] = edge_list
    return out


def _inject_queue(env, sc, w, rng):
    """Overwrite agent states (through the reference's own objects) so that the rollout exercises
    the gate / tube / exit / goal branches: a queue of agents in front of the tube entrance."""
    tp = w.tube_params
    e = np.array(tp["e"], dtype=np.float64)
    n = np.array([-e[1], e[0]])
    head = np.arctan2(e[1], e[0])
    inj = []
    for k, a in enumerate(w.agents):
        pos = tp["entrance"] - e * (0.12 + 0.27 * k) + n * rng.uniform(-0.12, 0.12)
        th = head + rng.uniform(-0.3, 0.3)
        sp = rng.uniform(0.05, 0.09)
        a.state.p_pos = np.array(pos)
        a.state.theta = th
        a.state.speed = sp
        inj.append([pos[0], pos[1], th, sp])
    return np.array(inj)


def _guided_action(w, sc, rng, w_opt, a_opt):
    tp = w.tube_params
    e = np.array(tp["e"], dtype=np.float64)
    acts = []
    for i, a in enumerate(w.agents):
        if rng.rand() < 0.2:
            acts.append(rng.randint(0, 25)); continue
        p = a.state.p_pos
        s = float(np.dot(p - tp["entrance"], e))
        wp = tp["entrance"] if s < -0.02 else (tp["exit"] + 0.1 * e if s < tp["L"] + 0.05 else sc.landmark_poses[i])
        des = np.arctan2(wp[1] - p[1], wp[0] - p[0])
        err = (des - a.state.theta + np.pi) % (2 * np.pi) - np.pi
        j = int(np.argmin(np.abs(5.0 * w_opt - err)))
        k = 4 if rng.rand() < 0.8 else int(rng.randint(0, 5))
        acts.append(j * 5 + k)
    return np.array(acts)


def july_rollout(num_agents, seed, T, world_size=4.0, episode_length=25, guided=False,
                 scenario_name="nav_metered_one_goal_graph_rotate_tube_july", graph_feat_type="relative", formation_type="point"):
    np.random.seed(seed)
    args = H.july_args(num_agents, world_size=world_size, episode_length=episode_length, scenario_name=scenario_name,
                       graph_feat_type=graph_feat_type, formation_type=formation_type)
    info_keys = INFO_KEYS + (["Phase_reached"] if ("rot_inv" in scenario_name or "phase_graph" in scenario_name) else [])
    with H.UniformTape() as tape:
        env, sc, w = H.make_july_env(args)
        A = num_agents
        E = len(w.entities)
        n_ctor = len(tape.samples)
        pre = H.snapshot(env, sc, w)
        out = {"A": A, "E": E, "T": T, "seed": seed, "world_size": world_size,
               "episode_length": episode_length, "collision_rew": args.collision_rew,
               "formation_rew": args.formation_rew, "goal_rew": args.goal_rew,
               "max_speed": args.max_speed, "init_prev_phase": pre["prev_phase"], "formation_type": formation_type}
        tape.samples.clear()
        o, ids, nd, ad = env.reset(0)
        assert all(a is ad[0] for a in ad)
        st = H.snapshot(env, sc, w)
        out.update({"reset0_obs": np.array(o), "reset0_id": np.array(ids), "reset0_node": np.array(nd),
                    "reset0_adj": np.array(ad[0])})
        out.update({"reset0_" + k: st[k] for k in STATE_KEYS + TUBE_KEYS})
        rng_inj = np.random.RandomState(5000 + seed)
        injs = []
        if guided:
            injs.append(_inject_queue(env, sc, w, rng_inj))
        w_opt = np.linspace(-0.1, 0.1, 5); a_opt = np.linspace(-0.001, 0.002, 5)
        tape_pos = [len(tape.samples)]
        rec = {k: [] for k in ["act", "obs", "node", "adj", "rew", "done", "info", "did_reset",
                               "ret_obs", "ret_node", "ret_adj", "edges", "n_edges"]}
        srec = {k: [] for k in STATE_KEYS}
        rrec = {k: [] for k in STATE_KEYS + TUBE_KEYS}
        rng_act = np.random.RandomState(1000 + seed)     # separate stream: does not touch global RNG
        m_max = E * E
        for t in range(T):
            idx = _guided_action(w, sc, rng_act, w_opt, a_opt) if guided else rng_act.randint(0, 25, A)
            onehot = np.eye(25)[idx]
            o, ids, nd, ad, rw, dn, info = env.step([onehot[i] for i in range(A)])
            assert all(a is ad[0] for a in ad)           # one aliased matrix per env
            rec["act"].append(idx.astype(np.int32))
            rec["obs"].append(np.array(o)); rec["node"].append(np.array(nd))
            rec["adj"].append(np.array(ad[0])); rec["rew"].append(np.array(rw, dtype=np.float64))
            rec["done"].append(np.array(dn, dtype=bool))
            rec["info"].append(np.array([[float(info[i][k]) for k in info_keys] for i in range(A)]))
            rec["edges"].append(pad_edges(w.edge_list, m_max)); rec["n_edges"].append(w.edge_list.shape[1])
            st = H.snapshot(env, sc, w)
            for k in STATE_KEYS:
                srec[k].append(st[k])
            if np.all(dn):                               # graphworker auto-reset (env_wrappers.py:865-870)
                o, ids, nd, ad = env.reset(t)
                rs = H.snapshot(env, sc, w)
                for k in STATE_KEYS + TUBE_KEYS:
                    rrec[k].append(rs[k])
                rec["did_reset"].append(True)
                if guided:
                    injs.append(_inject_queue(env, sc, w, rng_inj))
            else:
                rec["did_reset"].append(False)
            rec["ret_obs"].append(np.array(o)); rec["ret_node"].append(np.array(nd))
            rec["ret_adj"].append(np.array(ad[0]))
            tape_pos.append(len(tape.samples))
        out["tape"] = np.array(tape.samples, dtype=np.float64)
        out["tape_pos"] = np.array(tape_pos, dtype=np.int64)
        out["n_ctor_draws"] = n_ctor
        out["guided"] = guided
        out["inject"] = np.array(injs) if guided else np.zeros((0, A, 4))
    for k, v in rec.items():
        out[k] = np.array(v)
    for k, v in srec.items():
        out["st_" + k] = np.array(v)
    for k, v in rrec.items():
        out["rs_" + k] = np.array(v)
    out["info_keys"] = np.array(info_keys)
    out["scenario_name"] = scenario_name
    return out


def rk45_fixture(n=40, seed=5):
    H.install_stubs()
    from multiagent.core import AirTaxiXYState
    from multiagent.config import AirTaxiConfig as C
    rng = np.random.RandomState(seed)
    w_opt = np.linspace(-C.ANGULAR_RATE_MAX, C.ANGULAR_RATE_MAX, 5)
    a_opt = np.linspace(C.ACCEL_MIN, C.ACCEL_MAX, 5)
    s_in, u_in, s_out, pd = [], [], [], []
    for _ in range(n):
        s0 = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(-7, 7),
                       rng.uniform(C.V_MIN * 0.9, C.V_MAX * 1.1)])
        for idx in range(25):
            u = 5.0 * np.array([w_opt[idx // 5], a_opt[idx % 5]])
            st = AirTaxiXYState(C.V_MIN, C.V_MAX)
            st.values = s0.copy()
            st.update_state(u, C.DT)
            s_in.append(s0); u_in.append(u); s_out.append(np.array(st.values)); pd.append(st.p_dist)
    return dict(s_in=np.array(s_in), u_in=np.array(u_in), s_out=np.array(s_out), p_dist=np.array(pd),
                w_opt=w_opt, a_opt=a_opt, v_min=C.V_MIN, v_max=C.V_MAX, dt=C.DT)


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def force_classic_fixture(n_worlds=12, seed=11):
    """Live classic MPE force step: onpolicy/envs/mpe/core.py:183-335 (NumPy only, loaded by path)."""
    core = _load_by_path("ref_mpe_core", os.path.join(H.REF, "onpolicy/envs/mpe/core.py"))
    rng = np.random.RandomState(seed)
    recs = []
    for wi in range(n_worlds):
        A, L = int(rng.randint(2, 7)), int(rng.randint(1, 5))
        w = core.World()
        w.agents = [core.Agent() for _ in range(A)]
        w.landmarks = [core.Landmark() for _ in range(L)]
        nw = int(rng.randint(0, 3))
        walls = []
        for k in range(nw):
            walls.append(core.Wall(orient="H" if k % 2 == 0 else "V", axis_pos=float(rng.uniform(-1, 1)),
                                   endpoints=(-0.6, 0.7), width=0.1))
        w.walls = walls
        for i, a in enumerate(w.agents):
            a.name = "agent %d" % i; a.collide = True; a.silent = True
            a.size = float(rng.choice([0.05, 0.15])); a.max_speed = float(rng.choice([0.3, 1.0]))
            a.accel = float(rng.choice([3.0, 5.0])) if rng.rand() < 0.7 else None
            a.state.p_pos = rng.uniform(-0.8, 0.8, 2) if i else np.array([-0.62, walls[0].axis_pos + 0.02]) if nw else rng.uniform(-0.8, 0.8, 2)
            a.state.p_vel = rng.uniform(-0.5, 0.5, 2)
            a.state.c = np.zeros(w.dim_c)
            a.action.u = rng.uniform(-1, 1, 2); a.action.c = np.zeros(w.dim_c)
        for i, l in enumerate(w.landmarks):
            l.name = "landmark %d" % i; l.collide = bool(i % 2 == 0); l.movable = False
            l.size = 0.2
            l.state.p_pos = rng.uniform(-0.8, 0.8, 2); l.state.p_vel = np.zeros(2)
        # make a couple of agents overlap so contact forces are exercised
        if A >= 2:
            w.agents[1].state.p_pos = w.agents[0].state.p_pos + np.array([0.03, -0.021])
        ent = w.entities
        rec = dict(A=A, L=L,
                   pos=np.array([e.state.p_pos for e in ent]), vel=np.array([e.state.p_vel for e in ent]),
                   size=np.array([e.size for e in ent]), collide=np.array([e.collide for e in ent]),
                   movable=np.array([e.movable for e in ent]),
                   mass=np.array([e.mass for e in ent]),
                   max_speed=np.array([np.nan if e.max_speed is None else e.max_speed for e in ent]),
                   accel=np.array([np.nan if a.accel is None else a.accel for a in w.agents]),
                   u=np.array([a.action.u for a in w.agents]),
                   wall_orient=np.array([0 if wl.orient == "H" else 1 for wl in walls], dtype=np.int32),
                   wall_axis=np.array([wl.axis_pos for wl in walls]),
                   wall_ends=np.array([wl.endpoints for wl in walls]).reshape(-1, 2),
                   wall_width=np.array([wl.width for wl in walls]),
                   dt=w.dt, damping=w.damping, contact_force=w.contact_force, contact_margin=w.contact_margin)
        out_pos, out_vel = [], []
        for _ in range(3):
            w.step()
            out_pos.append(np.array([e.state.p_pos for e in w.entities]))
            out_vel.append(np.array([e.state.p_vel for e in w.entities]))
        rec["out_pos"] = np.array(out_pos); rec["out_vel"] = np.array(out_vel)
        recs.append(rec)
    flat = {"n_worlds": n_worlds}
    for wi, r in enumerate(recs):
        for k, v in r.items():
            flat["w%d_%s" % (wi, k)] = v
    return flat


def force_di_fixture(n_worlds=12, seed=13):
    """multiagent/core.py dead-code force path on a DoubleIntegrator world."""
    H.install_stubs()
    from multiagent import core
    from multiagent.config import DoubleIntegratorConfig as DC
    # harness-side shim for the renamed attribute — touches the imported class only
    if not hasattr(DC, "COORDINATION_RANGE"):
        DC.COORDINATION_RANGE = DC.COMMUNICATION_RANGE
    rng = np.random.RandomState(seed)
    flat = {"n_worlds": n_worlds}
    for wi in range(n_worlds):
        A, L, O = int(rng.randint(2, 7)), int(rng.randint(1, 4)), int(rng.randint(0, 3))
        nw = int(rng.randint(0, 3))
        w = core.World(core.EntityDynamicsType.DoubleIntegratorXY)
        w.agents = [core.Agent(core.EntityDynamicsType.DoubleIntegratorXY) for _ in range(A)]
        w.landmarks = [core.Landmark() for _ in range(L)]
        w.obstacles = [core.Landmark() for _ in range(O)]
        w.walls = [core.Wall(orient="H" if k % 2 == 0 else "V", axis_pos=float(rng.uniform(-1, 1)),
                             endpoints=(-0.6, 0.7), width=0.1) for k in range(nw)]
        for k, wl in enumerate(w.walls):
            wl.id = k; wl.name = "wall %d" % k; wl.collide = True; wl.movable = False
            wl.ghost = False
        for i, a in enumerate(w.agents):
            a.id = i; a.name = "agent %d" % i; a.collide = True; a.silent = True
            a.max_speed = float(rng.choice([0.5, 2.0])); a.status = bool(rng.rand() < 0.25)
            a.state.p_pos = rng.uniform(-1.5, 1.5, 2); a.state.p_vel = rng.uniform(-0.8, 0.8, 2)
            a.action.u = 5.0 * rng.choice([-1.0, 0.0, 1.0], 2); a.action.c = np.zeros(2)
        if A >= 2:
            w.agents[1].state.p_pos = w.agents[0].state.p_pos + np.array([0.21, -0.12])
        if nw:
            w.agents[0].state.p_pos = np.array([-0.63, w.walls[0].axis_pos + 0.05])
            if A >= 2:
                w.agents[1].state.p_pos = w.agents[0].state.p_pos + np.array([0.21, -0.12])
        for i, l in enumerate(w.landmarks):
            l.id = i; l.name = "landmark %d" % i; l.collide = False; l.movable = False
            l.state.p_pos = rng.uniform(-1.5, 1.5, 2)
        for i, o in enumerate(w.obstacles):
            o.name = "obstacle %d" % i; o.collide = True; o.movable = False
            o.state.p_pos = w.agents[-1].state.p_pos + rng.uniform(-0.3, 0.3, 2)
        ent = w.entities
        n_phys = A + L + O          # walls are entities too in the reference's list; keep them out of pos arrays
        pre = dict(A=A, L=L, O=O,
                   pos=np.array([e.state.p_pos for e in ent[:n_phys]]),
                   vel=np.array([e.state.p_vel for e in ent[:n_phys]]),
                   status=np.array([a.status for a in w.agents]),
                   max_speed=np.array([a.max_speed for a in w.agents]),
                   u=np.array([a.action.u for a in w.agents]),
                   wall_orient=np.array([0 if wl.orient == "H" else 1 for wl in w.walls], dtype=np.int32),
                   wall_axis=np.array([wl.axis_pos for wl in w.walls]),
                   wall_ends=np.array([wl.endpoints for wl in w.walls]).reshape(-1, 2),
                   wall_width=np.array([wl.width for wl in w.walls]),
                   dt=w.dt, damping=w.damping, contact_force=w.contact_force,
                   contact_margin=w.contact_margin, wall_contact_force=w.wall_contact_force,
                   wall_contact_margin=w.wall_contact_margin, d_min=DC.COLLISION_DISTANCE,
                   size=ent[0].size)
        out_pos, out_vel, out_pd = [], [], []
        for _ in range(3):
            w.calculate_distances()
            p_force = [None] * len(w.entities)
            p_force = w.apply_action_force(p_force)
            p_force = w.apply_environment_force(p_force)
            w.integrate_state(p_force)
            out_pos.append(np.array([e.state.p_pos for e in w.entities[:n_phys]]))
            out_vel.append(np.array([e.state.p_vel for e in w.entities[:n_phys]]))
            out_pd.append(np.array([a.state.p_dist for a in w.agents]))
        pre["out_pos"] = np.array(out_pos); pre["out_vel"] = np.array(out_vel); pre["out_pdist"] = np.array(out_pd)
        for k, v in pre.items():
            flat["w%d_%s" % (wi, k)] = v
    return flat


def misc_fixture():
    H.install_stubs()
    from multiagent.config import AirTaxiConfig as C, DoubleIntegratorConfig as DC
    return dict(w_opt=np.linspace(-C.ANGULAR_RATE_MAX, C.ANGULAR_RATE_MAX, 5),
                a_opt=np.linspace(C.ACCEL_MIN, C.ACCEL_MAX, 5),
                airtaxi=np.array([C.V_MIN, C.V_MAX, C.DT, C.DISTANCE_TO_GOAL_THRESHOLD,
                                  C.COLLISION_DISTANCE, C.COORDINATION_RANGE]),
                di=np.array([DC.VX_MAX, DC.DT, DC.DISTANCE_TO_GOAL_THRESHOLD, DC.COLLISION_DISTANCE,
                             DC.COMMUNICATION_RANGE]))


def main_july():
    jobs = [(3, 0, 60, 4.0, 25, False), (3, 1, 60, 4.0, 25, False), (10, 0, 55, 4.0, 25, False),
            (3, 2, 130, 2.0, 60, True), (3, 3, 130, 2.0, 60, True), (6, 4, 130, 3.0, 70, True),
            (10, 5, 100, 4.0, 90, True)]
    for A, seed, T, ws, el, guided in jobs:
        d = july_rollout(A, seed, T, world_size=ws, episode_length=el, guided=guided)
        p = os.path.join(HERE, "july_A%d_s%d%s.npz" % (A, seed, "_guided" if guided else ""))
        np.savez_compressed(p, **d)
        print(p, os.path.getsize(p), "resets", int(d["did_reset"].sum()),
              "steps with a done agent", int(d["st_status"].any(axis=1).sum()),
              "max phase", int(d["obs"][:, :, 18].max()), "phase_reached", d["st_phase_reached"].max(axis=0))


def main_rot(ROT="nav_graph_metered_single_corridor_rot_inv", prefix="rotinv", seed=5, phase_col=12):
    # rot_inv. The reference crashes at env construction for some seeds (an agent placed inside the
    # tube before `previous_phase` exists: AttributeError in get_agent_phase, rot_inv.py:712) — such seeds are skipped.
    # The two/three-phase variants (same family, D = 15) reuse this generator with their own prefix.
    for A, T, ws, el, guided in [(3, 60, 4.0, 25, False), (10, 40, 4.0, 25, False), (3, 130, 2.0, 60, True),
                                 (3, 130, 2.0, 60, True), (6, 130, 3.0, 70, True), (10, 100, 4.0, 90, True)]:
        while True:
            seed += 1
            try:
                d = july_rollout(A, seed, T, world_size=ws, episode_length=el, guided=guided, scenario_name=ROT)
                break
            except AttributeError as e:
                print("seed", seed, "reference crashed:", str(e)[:70])
        p = os.path.join(HERE, "%s_A%d_s%d%s.npz" % (prefix, A, seed, "_guided" if guided else ""))
        np.savez_compressed(p, **d)
        print(p, os.path.getsize(p), "resets", int(d["did_reset"].sum()), "steps with a done agent", int(d["st_status"].any(axis=1).sum()),
              "max phase", int(d["obs"][:, :, phase_col].max()), "phase_reached", d["st_phase_reached"].max(axis=0), "cooldown max", d["st_cooldown"].max())


def main_global(scenario_name="nav_metered_one_goal_graph_rotate_tube_july", prefix="julyglobal", seed=5,
                jobs=((3, 60, 4.0, 25, False), (6, 120, 3.0, 60, True))):
    # graph_feat_type='global' (_get_entity_feat_global, …_july.py:1672-1691; rot_inv.py:1668-1687): node rows [vel, pos, goal, type] in world coordinates, F = 7
    for A, T, ws, el, guided in jobs:
        while True:                                   # the reference crashes at construction for some seeds (see main_rot)
            seed += 1
            try:
                d = july_rollout(A, seed, T, world_size=ws, episode_length=el, guided=guided, graph_feat_type="global", scenario_name=scenario_name)
                break
            except AttributeError as e:
                print("seed", seed, "reference crashed:", str(e)[:70])
        d["graph_feat_type"] = "global"
        p = os.path.join(HERE, "%s_A%d_s%d%s.npz" % (prefix, A, seed, "_guided" if guided else ""))
        np.savez_compressed(p, **d)
        print(p, os.path.getsize(p), "node row width", d["node"].shape[-1], "resets", int(d["did_reset"].sum()),
              "steps with a done agent", int(d["st_status"].any(axis=1).sum()), "phase_reached", d["st_phase_reached"].max(axis=0))


def main_formation():
    # formation_type 'line' / 'circle' (…_july.py:492-495 -> custom_scenarios/utils.py:77-130, 231-267; the same call sites in the rot_inv family's
    # files): DISTINCT landmark positions, so the landmark x landmark adjacency block, per-agent goals, landmark masks on distinct rows and
    # info_callback's nearest-landmark logic are pinned against the reference in their general form ('point' makes every landmark coincide).
    # Guided rollouts: the agents fly gate -> tube -> exit -> their OWN landmark.
    JULY = "nav_metered_one_goal_graph_rotate_tube_july"
    jobs = [(JULY, "july", "line", 3, 2, 130, 2.0, 60), (JULY, "july", "line", 6, 4, 140, 3.0, 70),
            (JULY, "july", "circle", 3, 2, 130, 2.0, 60), (JULY, "july", "circle", 6, 4, 140, 3.0, 70),
            ("nav_graph_metered_single_corridor_rot_inv", "rotinv", "line", 4, 80, 130, 2.4, 60),
            ("nav_graph_metered_single_corridor_rot_inv", "rotinv", "circle", 6, 84, 140, 3.0, 70),
            ("three_phase_graph", "threephase", "line", 3, 90, 130, 2.0, 60), ("three_phase_graph", "threephase", "circle", 5, 94, 140, 3.0, 70),
            ("two_phase_graph", "twophase", "line", 3, 100, 90, 2.0, 45)]
    for name, prefix, form, A, seed, T, ws, el in jobs:
        seed -= 1
        while True:                                   # the reference crashes at construction for some seeds (see main_rot)
            seed += 1
            try:
                d = july_rollout(A, seed, T, world_size=ws, episode_length=el, guided=True, scenario_name=name, formation_type=form)
                break
            except AttributeError as e:
                print("seed", seed, "reference crashed:", str(e)[:70])
        p = os.path.join(HERE, "%s%s_A%d_s%d_guided.npz" % (prefix, form, A, seed))
        np.savez_compressed(p, **d)
        lm = d["reset0_landmarks"]
        print(p, os.path.getsize(p), "resets", int(d["did_reset"].sum()), "agents that reached their goal", d["st_status"].max(axis=0).astype(int),
              "goal_tracker", d["st_goal_tracker"].max(axis=0), "distinct landmarks", len({tuple(np.round(q, 9)) for q in lm}))


def main_blocks():
    np.savez_compressed(os.path.join(HERE, "rk45_airtaxi.npz"), **rk45_fixture())
    np.savez_compressed(os.path.join(HERE, "force_classic.npz"), **force_classic_fixture())
    np.savez_compressed(os.path.join(HERE, "force_di.npz"), **force_di_fixture())
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc_fixture())


def main(which):
    """`python make_fixtures.py [july|rot|phase|blocks ...]` regenerates the named groups (default: all)."""
    H.selfcheck_uniform_patch()
    which = which or ["july", "global", "globalrot", "rot", "phase", "formation", "blocks"]
    if "july" in which:
        main_july()
    if "global" in which:
        main_global()
    if "globalrot" in which:
        main_global("nav_graph_metered_single_corridor_rot_inv", "rotinvglobal", seed=60, jobs=((4, 110, 2.4, 50, True),))
        main_global("two_phase_graph", "twophaseglobal", seed=70, jobs=((3, 70, 4.0, 25, False),))
    if "rot" in which:
        main_rot()
    if "phase" in which:
        main_rot("two_phase_graph", "twophase", seed=20, phase_col=14)
        main_rot("three_phase_graph", "threephase", seed=40, phase_col=14)
    if "formation" in which:
        main_formation()
    if "blocks" in which:
        main_blocks()
    print("done")


if __name__ == "__main__":
    main(sys.argv[1:])
