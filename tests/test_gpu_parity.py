"""GPU parity: the HIP engine (through the C ABI) vs the CPU oracle and vs the golden vectors.

Bars (BASELINE.json north_star): positions/velocities/observations within 1e-5 in fp32, graph edge
indices and dones bit-exact.
"""
import glob
import os

import numpy as np
import pytest

import gmpe
from gmpe.config import INFO_KEYS
import oracle_lib as ol

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
JULY = [q for pre in ("july", "julyglobal", "rotinv", "twophase", "threephase", "rotinvglobal", "twophaseglobal",
                       "julyline", "julycircle", "rotinvline", "rotinvcircle", "twophaseline", "threephaseline", "threephasecircle") for q in sorted(glob.glob(os.path.join(GOLD, pre + "_A*_s*.npz")))]
ROT = "nav_graph_metered_single_corridor_rot_inv"
ROTFAM = [ROT, "two_phase_graph", "three_phase_graph"]
TOL = 1e-5


def _engine(cfg, **kw):
    from gmpe.engine import GmpeEngine
    return GmpeEngine(cfg, device=0, **kw)


def _np(t):
    return t.detach().cpu().numpy()


def edges_numpy(adj32, d, inclusive=False):
    """process_adj rule (gnn_new.py:329-358) on a [B,E,E] float32 array."""
    m = ((adj32 <= d) if inclusive else (adj32 < d)) & (adj32 > 0)
    b, r, c = np.nonzero(m)
    E = adj32.shape[-1]
    return np.stack([b * E + r, b * E + c]).astype(np.int32), adj32[b, r, c]


def _july_cfg(d, scenario_name="nav_metered_one_goal_graph_rotate_tube_july", **kw):
    return gmpe.make_config(scenario_name=scenario_name,
                            num_envs=1, num_agents=int(d["A"]), world_size=float(d["world_size"]),
                            episode_length=int(d["episode_length"]), max_speed=float(d["max_speed"]),
                            collision_rew=float(d["collision_rew"]), formation_rew=float(d["formation_rew"]),
                            goal_rew=float(d["goal_rew"]), graph_feat_type=str(d["graph_feat_type"]) if "graph_feat_type" in d else "relative",
                            formation_type=str(d["formation_type"]) if "formation_type" in d else "point", **kw)


@pytest.mark.parametrize("path", JULY, ids=[os.path.basename(p)[:-4] for p in JULY])
def test_golden_replay_on_gpu(path):
    """The reference's own rollouts (incl. its np.random draws via the tape) replayed on the GPU."""
    d = np.load(path)
    A, E, T = int(d["A"]), int(d["E"]), int(d["T"])
    rot = not os.path.basename(path).startswith("july")              # the rot_inv family: float32 rotated features, F = 7
    eng = _engine(_july_cfg(d, str(d["scenario_name"])) if rot else _july_cfg(d))     # the July fixtures predate the name field
    assert _np(eng.out.node_obs).shape[-1] == (7 if (rot or "global" in os.path.basename(path)) else 8)
    eng.set("prev_phase", d["init_prev_phase"][None])
    eng.set_tape(d["tape"][None])
    o = eng.reset()
    np.testing.assert_allclose(_np(o.obs)[0], d["reset0_obs"], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.node_obs)[0], d["reset0_node"], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.adj)[0], np.broadcast_to(d["reset0_adj"], (A, E, E)), rtol=0, atol=TOL)
    np.testing.assert_array_equal(_np(o.agent_id)[0], d["reset0_id"])
    guided = bool(d["guided"])
    n_inj = 0

    def inject():
        nonlocal n_inj
        inj = d["inject"][n_inj]; n_inj += 1
        eng.set("x", inj[None, :, 0]); eng.set("y", inj[None, :, 1]); eng.set("s2", inj[None, :, 2]); eng.set("s3", inj[None, :, 3])

    if guided:
        inject()
    import torch
    for t in range(T):
        o = eng.step(torch.as_tensor(d["act"][t][None].astype(np.int32)))
        np.testing.assert_allclose(_np(o.reward)[0], d["rew"][t], rtol=0, atol=TOL, err_msg="rew t=%d" % t)
        np.testing.assert_array_equal(_np(o.done)[0].astype(bool), d["done"][t], err_msg="done t=%d" % t)
        np.testing.assert_allclose(_np(o.obs)[0], d["ret_obs"][t], rtol=0, atol=TOL, err_msg="obs t=%d" % t)
        np.testing.assert_allclose(_np(o.node_obs)[0], d["ret_node"][t], rtol=0, atol=TOL, err_msg="node t=%d" % t)
        adj = _np(o.adj)[0]
        np.testing.assert_allclose(adj, np.broadcast_to(d["ret_adj"][t], (A, E, E)), rtol=0, atol=TOL, err_msg="adj t=%d" % t)
        np.testing.assert_array_equal(adj == 0, np.broadcast_to(d["ret_adj"][t] == 0, (A, E, E)))
        K = d["info"].shape[-1]                       # 17 keys in the July file, 18 (+Phase_reached) in rot_inv
        np.testing.assert_allclose(_np(o.info)[0][:, :K], d["info"][t], rtol=2e-6, atol=2e-5, err_msg="info t=%d" % t)
        if not d["did_reset"][t]:
            if rot:
                np.testing.assert_array_equal(eng.get("cooldown")[0], d["st_cooldown"][t], err_msg="cooldown t=%d" % t)
                np.testing.assert_allclose(eng.get("prev_proj")[0], d["st_prev_proj"][t], rtol=0, atol=2e-6, err_msg="prev_proj t=%d" % t)
            np.testing.assert_allclose(eng.get("x")[0], d["st_x"][t], rtol=0, atol=2e-6)
            np.testing.assert_array_equal(eng.get("status")[0].astype(bool), d["st_status"][t])
            np.testing.assert_array_equal(eng.get("prev_phase")[0], d["st_prev_phase"][t])
            np.testing.assert_array_equal(eng.get("phase_reached")[0], d["st_phase_reached"][t])
        elif guided:
            inject()
        assert eng.get("rng_ctr")[0] == d["tape_pos"][t + 1], "draw count t=%d" % t
    eng.check_errors()
    eng.close()


def _compare_step(eo, oo, E, A, label):
    obs, ids, node, adj, rew, done, info, did = oo
    np.testing.assert_allclose(_np(eo.obs), obs, rtol=0, atol=TOL, err_msg=label + " obs")
    np.testing.assert_allclose(_np(eo.node_obs), node, rtol=0, atol=TOL, err_msg=label + " node")
    eadj = _np(eo.adj)
    np.testing.assert_allclose(eadj, np.broadcast_to(adj[:, None], eadj.shape), rtol=0, atol=TOL, err_msg=label + " adj")
    # rtol = half an fp32 ulp: the engine's outputs ARE float32, and a shared reward (the sum over all agents, environment.py:1056-1061) reaches
    # magnitudes (~300 with 21 agents) where float32 rounding alone exceeds the 1e-5 absolute bar
    np.testing.assert_allclose(_np(eo.reward), rew, rtol=6e-8, atol=TOL, err_msg=label + " rew")
    np.testing.assert_array_equal(_np(eo.done).astype(bool), done, err_msg=label + " done")
    np.testing.assert_allclose(_np(eo.info), info, rtol=2e-6, atol=2e-5, err_msg=label + " info")
    np.testing.assert_array_equal(_np(eo.agent_id), ids)
    # edge indices bit-exact under both thresholding rules (learner's `<` and update_graph's `<=`)
    o32 = adj.astype(np.float32)
    for dist, incl in ((1.0, False), (4.82802, True)):
        e_ref, _ = edges_numpy(o32, dist, incl)
        e_gpu, _ = edges_numpy(eadj[:, 0], dist, incl)
        np.testing.assert_array_equal(e_gpu, e_ref, err_msg=label + " edge indices")


STATE_F = ["x", "y", "s2", "s3", "p_dist", "time"]
STATE_I = ["status", "prev_phase", "phase_reached", "cooldown", "goal_tracker", "current_step", "rng_ctr",
           "times_required", "dists_to_goal", "dist_left", "goal_reached", "n_agent_coll", "n_obst_coll",
           "spacing_viol", "steps_in_corr", "conformance", "error_flags"]


def _compare_state(eng, orc, label):
    for f in STATE_F + ["tube", "landmarks", "obstacles", "goal_min_time", "delta_spacing", "prev_proj"]:
        np.testing.assert_allclose(eng.get(f), orc.get(f), rtol=0, atol=1e-9, err_msg=label + " " + f)
    for f in STATE_I:
        np.testing.assert_array_equal(eng.get(f), orc.get(f), err_msg=label + " " + f)


def _rollout_vs_oracle(cfg, steps, seed, shrink_world=False):
    import torch
    eng = _engine(cfg)
    orc = ol.Oracle(cfg)
    rng = np.random.RandomState(seed)
    eo = eng.reset(); oo = orc.reset()
    np.testing.assert_allclose(_np(eo.obs), oo[0], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(eo.node_obs), oo[2], rtol=0, atol=TOL)
    _compare_state(eng, orc, "reset")
    if shrink_world:
        # pull the agents close to the tube so phases 1/2, goal reaches and collisions happen
        st = {k: orc.get(k) for k in ("x", "y", "s2", "s3", "tube")}
        N, A = cfg.num_envs, cfg.num_agents
        e = st["tube"][:, 5:7]; ent = st["tube"][:, 1:3]
        k = np.arange(A)[None, :, None]
        pos = ent[:, None, :] - e[:, None, :] * (0.1 + 0.2 * k) + rng.uniform(-0.1, 0.1, (N, A, 2))
        th = np.arctan2(e[:, 1], e[:, 0])[:, None] + rng.uniform(-0.2, 0.2, (N, A))
        for tgt in (eng, orc):
            tgt.set("x", pos[..., 0]); tgt.set("y", pos[..., 1]); tgt.set("s2", th); tgt.set("s3", np.full((N, A), 0.08))
    n_resets = 0
    for t in range(steps):
        if shrink_world:
            # steer roughly along the tube: keep heading (w index 2) mostly, full accel
            act = np.where(rng.rand(cfg.num_envs, cfg.num_agents) < 0.7, 2 * 5 + 4, rng.randint(0, cfg.n_actions, (cfg.num_envs, cfg.num_agents)))
        else:
            act = rng.randint(0, cfg.n_actions, (cfg.num_envs, cfg.num_agents))
        act = act.astype(np.int32)
        eo = eng.step(torch.as_tensor(act))
        oo = orc.step(act)
        _compare_step(eo, oo, cfg.num_entities, cfg.num_agents, "t=%d" % t)
        _compare_state(eng, orc, "t=%d" % t)
        n_resets += int(oo[7].sum())
    eng.check_errors()
    eng.close()
    return n_resets


def test_july_random_rollout_vs_oracle_philox():
    cfg = gmpe.make_config(num_envs=96, num_agents=10, world_size=4.0, episode_length=12, seed=123)
    assert _rollout_vs_oracle(cfg, 30, seed=1) >= 96 * 2


def test_july_tube_transit_vs_oracle():
    cfg = gmpe.make_config(num_envs=64, num_agents=4, world_size=2.4, episode_length=40, seed=7)
    _rollout_vs_oracle(cfg, 45, seed=2, shrink_world=True)


@pytest.mark.parametrize("scen", ROTFAM)
def test_rotfam_random_rollout_vs_oracle_philox(scen):
    cfg = gmpe.make_config(scenario_name=scen, num_envs=96, num_agents=10, world_size=4.0, episode_length=12, seed=321)
    assert cfg.node_feats == 7 and cfg.obs_dim == (13 if scen == ROT else 15)
    assert _rollout_vs_oracle(cfg, 30, seed=5) >= 96 * 2


@pytest.mark.parametrize("scen", ROTFAM)
def test_rotfam_tube_transit_vs_oracle(scen):
    """Agents pushed through the corridor: entrance-gate bonus + cooldown, progress / heading terms, exit gate, finish."""
    cfg = gmpe.make_config(scenario_name=scen, num_envs=64, num_agents=4, world_size=2.4, episode_length=40, seed=9)
    _rollout_vs_oracle(cfg, 45, seed=6, shrink_world=True)


@pytest.mark.parametrize("scen", ROTFAM)
def test_rotfam_three_agents_and_64_agents(scen):
    _rollout_vs_oracle(gmpe.make_config(scenario_name=scen, num_envs=5, num_agents=3, world_size=4.0, episode_length=9, seed=2), 20, seed=7)
    _rollout_vs_oracle(gmpe.make_config(scenario_name=scen, num_envs=3, num_agents=64, world_size=30.0, episode_length=4, seed=3), 9, seed=8)


@pytest.mark.parametrize("scen", ["nav_metered_one_goal_graph_rotate_tube_july", "navigation_graph"] + ROTFAM)
def test_more_landmarks_than_agents(scen):
    """num_landmarks need not equal num_agents (…_july.py:274): extra landmarks are graph nodes only."""
    cfg = gmpe.make_config(scenario_name=scen, num_envs=20, num_agents=3, num_landmarks=5, world_size=4.0, episode_length=7, seed=17,
                           num_obstacles=1 if scen == "navigation_graph" else 0)
    assert cfg.num_entities == (9 if scen == "navigation_graph" else 8)
    _rollout_vs_oracle(cfg, 18, seed=9)


def test_july_small_config_c1():
    cfg = gmpe.make_config(num_envs=3, num_agents=3, world_size=4.0, episode_length=25, seed=5)
    _rollout_vs_oracle(cfg, 30, seed=3)


def test_navigation_graph_vs_oracle():
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=64, num_agents=6, num_obstacles=3,
                           num_walls=4, world_size=3.0, episode_length=20, seed=11)
    assert _rollout_vs_oracle(cfg, 45, seed=4) >= 64


def test_navigation_graph_c2_shape():
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=32, num_agents=10, world_size=4.0,
                           episode_length=25, seed=3)
    _rollout_vs_oracle(cfg, 30, seed=5)


def test_onehot_actions_match_index_actions():
    import torch
    cfg = gmpe.make_config(num_envs=16, num_agents=4, seed=9)
    e1, e2 = _engine(cfg), _engine(cfg)
    e1.reset(); e2.reset()
    rng = np.random.RandomState(0)
    for _ in range(5):
        act = rng.randint(0, 25, (16, 4)).astype(np.int32)
        onehot = np.eye(25, dtype=np.float32)[act]
        o1 = e1.step(torch.as_tensor(act)); o2 = e2.step_onehot(torch.as_tensor(onehot))
        for k in ("obs", "node_obs", "adj", "reward", "done"):
            assert torch.equal(getattr(o1, k), getattr(o2, k)), k


def test_compact_adj_is_the_broadcast():
    import torch
    cfg = gmpe.make_config(num_envs=8, num_agents=5, seed=2)
    e1, e2 = _engine(cfg), _engine(cfg, adj_compact=True)
    e1.reset(); e2.reset()
    act = torch.zeros((8, 5), dtype=torch.int32)
    o1, o2 = e1.step(act), e2.step(act)
    assert torch.equal(o1.adj, o2.adj[:, None].expand_as(o1.adj))


def test_sharded_equals_unsharded():
    """Env ranges on different handles (GPUs) reproduce the single-handle run bit-for-bit (§8e)."""
    import torch
    N, A = 48, 6
    full = _engine(gmpe.make_config(num_envs=N, num_agents=A, seed=77, episode_length=8))
    parts = [_engine(gmpe.make_config(num_envs=N // 3, num_agents=A, seed=77, episode_length=8, env_id_base=g * (N // 3)))
             for g in range(3)]
    rng = np.random.RandomState(1)
    full.reset(); [p.reset() for p in parts]
    for t in range(20):
        act = torch.as_tensor(rng.randint(0, 25, (N, A)).astype(np.int32))
        of = full.step(act)
        for g, p in enumerate(parts):
            sl = slice(g * (N // 3), (g + 1) * (N // 3))
            op = p.step(act[sl])
            for k in ("obs", "node_obs", "adj", "reward", "done"):
                assert torch.equal(getattr(of, k)[sl].cpu(), getattr(op, k).cpu()), (t, g, k)


def test_edges_from_adj_kernel_bit_exact():
    import torch
    cfg = gmpe.make_config(num_envs=64, num_agents=10, seed=4)
    eng = _engine(cfg)
    eng.reset()
    o = eng.step(torch.zeros((64, 10), dtype=torch.int32))
    adj = o.adj.reshape(-1, 20, 20)
    for dist, incl in ((1.0, False), (4.82802, True), (0.5, False)):
        ei, ea, m = eng.edges_from_adj(adj, dist, inclusive=incl)
        e_ref, w_ref = edges_numpy(_np(adj), np.float32(dist), incl)
        assert m == e_ref.shape[1]
        np.testing.assert_array_equal(_np(ei), e_ref)
        np.testing.assert_array_equal(_np(ea), w_ref)


def test_full_size_properties_c3():
    """N=4096 x 10 agents (BASELINE config 3): size-independent invariants of the outputs."""
    import torch
    cfg = gmpe.make_config(num_envs=4096, num_agents=10, seed=1234)
    eng = _engine(cfg)
    eng.reset()
    g = torch.Generator(device="cpu"); g.manual_seed(42)
    for t in range(30):
        act = torch.randint(0, 25, (4096, 10), generator=g, dtype=torch.int32)
        o = eng.step(act)
    adj = o.adj
    assert torch.equal(adj, adj.transpose(-1, -2))                       # symmetric
    assert (torch.diagonal(adj, dim1=-2, dim2=-1) == 0).all()             # zero diagonal
    assert torch.equal(adj, adj[:, :1].expand_as(adj))                    # one matrix per env
    r = o.reward
    assert (r >= -20.0 - 1e-6).all() and (r <= 25.0 + 1e-6).all()        # clip(-4*cr, 5*gr)
    node = o.node_obs
    assert (node[:, :, :10, 7] == 0).all() and (node[:, :, 10:, 7] == 1).all()
    # ego row of node_obs is zero relative position / velocity
    idx = torch.arange(10, device=node.device)
    assert (node[:, idx, idx, 0:4] == 0).all()
    # adjacency entries equal the distance implied by node_obs rel_pos of ego (row = ego)
    rel = node[:, idx, :, 2:4].double()
    d = torch.sqrt((rel ** 2).sum(-1)).float()
    row = adj[:, idx, idx, :]
    live = row != 0
    assert torch.allclose(row[live], d[live], atol=1e-5)
    eng.check_errors()


@pytest.mark.parametrize("scen", ["navigation_graph"] + ROTFAM)
def test_full_size_properties_other_scenarios(scen):
    """4096 x 10 (the bench shapes c2 / c3r / c3p2 / c3p3): size-independent invariants + a 512-env slice against the oracle
    (counter-based RNG keyed by global env id: envs [0, 512) of the big batch == a 512-env engine with the same seed)."""
    import torch
    cfg = gmpe.make_config(scenario_name=scen, num_envs=4096, num_agents=10, seed=1234, episode_length=25)
    small = gmpe.make_config(scenario_name=scen, num_envs=512, num_agents=10, seed=1234, episode_length=25)
    eng, orc = _engine(cfg), ol.Oracle(small)
    eng.reset(); orc.reset()
    g = torch.Generator(device="cpu"); g.manual_seed(43)
    for t in range(30):
        act = torch.randint(0, cfg.n_actions, (4096, 10), generator=g, dtype=torch.int32)
        o = eng.step(act)
        oo = orc.step(act[:512].numpy())
    F = cfg.node_feats
    adj = o.adj
    assert torch.equal(adj, adj.transpose(-1, -2)) and (torch.diagonal(adj, dim1=-2, dim2=-1) == 0).all()
    assert torch.equal(adj, adj[:, :1].expand_as(adj))
    assert torch.isfinite(o.obs).all() and torch.isfinite(o.node_obs).all() and torch.isfinite(o.reward).all()
    assert (o.node_obs[:, :, :10, F - 1] == 0).all() and (o.node_obs[:, :, 10:, F - 1] == 1).all()
    idx = torch.arange(10, device=adj.device)
    rel = o.node_obs[:, idx, :, 2:4].double()                    # rotation preserves the norm of rel_pos
    d = torch.sqrt((rel ** 2).sum(-1)).float()
    row = adj[:, idx, idx, :]
    live = row != 0
    assert torch.allclose(row[live], d[live], atol=2e-5)
    np.testing.assert_allclose(_np(o.obs)[:512], oo[0], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.node_obs)[:512], oo[2], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.reward)[:512], oo[4], rtol=0, atol=TOL)
    np.testing.assert_array_equal(_np(o.done)[:512].astype(bool), oo[5])
    np.testing.assert_array_equal(eng.get("rng_ctr")[:512], orc.get("rng_ctr"))
    eng.check_errors()


# ---------------------------------------------------------------- other shapes / variants
def test_c4_shape_obstacles_walls_vs_oracle():
    """BASELINE configs[3] shape: 32 agents + 8 obstacles + 4 walls (E = 72), small N."""
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=12, num_agents=32, num_obstacles=8,
                           num_walls=4, world_size=8.0, episode_length=6, seed=21)
    assert _rollout_vs_oracle(cfg, 14, seed=6) >= 12


def test_c5_shape_64_agents_vs_oracle():
    """BASELINE configs[4] shape: 64 agents / 64 landmarks (E = 128): one env per tile, >64 KiB LDS."""
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=5, num_agents=64, world_size=12.0,
                           episode_length=5, seed=22)
    assert _rollout_vs_oracle(cfg, 11, seed=7) >= 5


def test_july_64_agents_vs_oracle():
    cfg = gmpe.make_config(num_envs=3, num_agents=64, world_size=30.0, episode_length=4, seed=23)
    _rollout_vs_oracle(cfg, 9, seed=8)


def test_unicycle_constants_vs_oracle():
    cfg = gmpe.make_config(dynamics_type="unicycle_vehicle", num_envs=32, num_agents=4, world_size=4.0,
                           episode_length=15, seed=24)
    _rollout_vs_oracle(cfg, 32, seed=9)


def test_nine_action_double_integrator_and_collaborative_reward():
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=32, num_agents=5, num_obstacles=1,
                           total_actions=9, collaborative=True, world_size=3.0, episode_length=10, seed=25)
    _rollout_vs_oracle(cfg, 22, seed=10)


def test_odd_entity_count_scalar_store_path():
    """E*E not a multiple of 4 (E = 7): the adjacency falls back to scalar stores."""
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=20, num_agents=3, num_obstacles=1,
                           world_size=3.0, episode_length=7, seed=26)
    assert cfg.num_entities == 7
    _rollout_vs_oracle(cfg, 16, seed=11)


def test_tile_shapes_give_identical_results(monkeypatch):
    """G (envs per workgroup) and BLOCK are pure performance knobs."""
    import torch
    outs = []
    for G, B in ((1, 64), (3, 128), (6, 256)):
        monkeypatch.setenv("GMPE_G", str(G)); monkeypatch.setenv("GMPE_BLOCK", str(B))
        eng = _engine(gmpe.make_config(num_envs=50, num_agents=10, seed=31, episode_length=7))
        eng.reset()
        g = torch.Generator(); g.manual_seed(3)
        for t in range(16):
            o = eng.step(torch.randint(0, 25, (50, 10), generator=g, dtype=torch.int32))
        outs.append([x.clone() for x in (o.obs, o.node_obs, o.adj, o.reward, o.done, o.info)])
        eng.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_step_many_equals_host_loop():
    import torch
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=40, num_agents=10, seed=8, episode_length=9)
    e1, e2 = _engine(cfg), _engine(cfg)
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    acts = torch.randint(0, 5, (7, 40, 10), generator=g, device="cuda", dtype=torch.int32)
    for k in range(23):
        o1 = e1.step(acts[k % 7])
    o2 = e2.step_many(acts, 23)
    for k in ("obs", "node_obs", "adj", "reward", "done", "info"):
        assert torch.equal(getattr(o1, k), getattr(o2, k)), k
    for f in ("x", "y", "s2", "s3", "rng_ctr", "current_step"):
        np.testing.assert_array_equal(e1.get(f), e2.get(f))


def test_step_many_hipgraph_replay_equals_host_loop():
    """gmpe_step_many_prepare: the K launches recorded into a hipGraph, replayed twice (the action buffer's CONTENTS change
    between the replays: the graph bakes in pointers, not data) == the same steps issued one by one."""
    import torch
    cfg = gmpe.make_config(num_envs=70, num_agents=10, seed=18, episode_length=9)
    e1, e2 = _engine(cfg), _engine(cfg)
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    acts = torch.randint(0, 25, (5, 70, 10), generator=g, device="cuda", dtype=torch.int32)
    e2.step_many_prepare(acts, 12)
    for rep in range(2):
        for k in range(12):
            o1 = e1.step(acts[k % 5])
        o2 = e2.step_many(acts, 12)                      # prepared -> one hipGraphLaunch
        torch.cuda.synchronize()
        for k in ("obs", "node_obs", "adj", "reward", "done", "info"):
            assert torch.equal(getattr(o1, k), getattr(o2, k)), k
        _compare_state(e1, e2, "rep %d" % rep)
        acts.copy_(torch.randint(0, 25, (5, 70, 10), generator=g, device="cuda", dtype=torch.int32))
    # a new RNG tape changes the kernel parameters: prepared graphs are dropped, the same call falls back to plain launches
    tape = np.random.RandomState(3).rand(70, 4096)
    e1.set_tape(tape); e2.set_tape(tape)
    for k in range(12):
        o1 = e1.step(acts[k % 5])
    o2 = e2.step_many(acts, 12)
    assert torch.equal(o1.obs, o2.obs) and torch.equal(o1.node_obs, o2.node_obs)
    _compare_state(e1, e2, "after set_tape")
    o1 = [e1.step(acts[k % 5]) for k in range(7)][-1]
    o2 = e2.step_many(acts, 7)                            # not prepared -> plain launch loop
    assert torch.equal(o1.obs, o2.obs) and torch.equal(o1.adj, o2.adj)
    e2.check_errors()


@pytest.mark.parametrize("G,B", [(2, 64), (6, 128), (3, 256)])
@pytest.mark.parametrize("scen", ["nav_metered_one_goal_graph_rotate_tube_july", "navigation_graph"] + ROTFAM)
def test_packed_tiles_with_staggered_resets_vs_oracle(monkeypatch, G, B, scen):
    """Several envs per workgroup, and envs of one tile resetting at DIFFERENT steps (mixed tiles):
    the episode clocks are staggered through set_field so that resets are not simultaneous."""
    import torch
    monkeypatch.setenv("GMPE_G", str(G)); monkeypatch.setenv("GMPE_BLOCK", str(B))
    N, A = 50, 10
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, num_obstacles=2 if scen == "navigation_graph" else 0,
                           world_size=4.0, episode_length=6, seed=41)
    eng, orc = _engine(cfg), ol.Oracle(cfg)
    eng.reset(); orc.reset()
    stagger = (np.arange(N) % 5).astype(np.int32)
    eng.set("current_step", stagger); orc.set("current_step", stagger)
    rng = np.random.RandomState(12)
    seen_mixed = False
    for t in range(20):
        act = rng.randint(0, cfg.n_actions, (N, A)).astype(np.int32)
        eo = eng.step(torch.as_tensor(act)); oo = orc.step(act)
        _compare_step(eo, oo, cfg.num_entities, A, "G=%d t=%d" % (G, t))
        _compare_state(eng, orc, "G=%d t=%d" % (G, t))
        did = oo[7]
        tiles = did[: (N // G) * G].reshape(-1, G)
        seen_mixed |= bool(((tiles.sum(1) > 0) & (tiles.sum(1) < G)).any())
    assert seen_mixed
    eng.check_errors()


def test_edges_from_adj_many_graphs_and_caps():
    """More than one scan chunk (B > 1024 graphs), ragged sizes, and a cap smaller than the edge count."""
    import torch
    eng = _engine(gmpe.make_config(num_envs=4, num_agents=3, seed=4))
    rng = np.random.RandomState(3)
    for B, E in ((2500, 7), (1025, 20), (3, 33)):
        adj = (rng.rand(B, E, E) * 2.0).astype(np.float32)
        adj[rng.rand(B, E, E) < 0.3] = 0.0
        t = torch.as_tensor(adj, device="cuda")
        ei, ea, m = eng.edges_from_adj(t, 1.0)
        e_ref, w_ref = edges_numpy(adj, np.float32(1.0), False)
        assert m == e_ref.shape[1]
        np.testing.assert_array_equal(_np(ei), e_ref); np.testing.assert_array_equal(_np(ea), w_ref)
        cap = max(1, m // 3)
        ei2, ea2, m2 = eng.edges_from_adj(t, 1.0, cap=cap)
        assert m2 == m and ei2.shape[1] == cap
        np.testing.assert_array_equal(_np(ei2), e_ref[:, :cap]); np.testing.assert_array_equal(_np(ea2), w_ref[:cap])


@pytest.mark.parametrize("scen,A,O", [("nav_metered_one_goal_graph_rotate_tube_july", 10, 0), ("navigation_graph", 5, 3), ("two_phase_graph", 3, 0)])
def test_edges_from_compact_adjacency_equal_materialised(scen, A, O):
    """gmpe_edges_from_adj_compact (each env's matrix read once, A id-shifted copies emitted) == gmpe_edges_from_adj on the materialised
    [N*A,E,E] tensor == NumPy nonzero, under both threshold rules, int32 and int64 ids, with and without a cap; N not a multiple of 16."""
    import torch
    N = 203
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, num_obstacles=O, world_size=3.0, episode_length=6, seed=4)
    eng = _engine(cfg)
    eng.reset()
    g = torch.Generator(); g.manual_seed(2)
    for t in range(9):
        o = eng.step(torch.randint(0, cfg.n_actions, (N, A), generator=g, dtype=torch.int32))
    E = cfg.num_entities
    full = o.adj.reshape(-1, E, E)
    compact = o.adj[:, 0].contiguous()
    for dist, incl in ((1.0, False), (4.82802, True), (0.3, False)):
        e_ref, w_ref = edges_numpy(_np(full), np.float32(dist), incl)
        ei, ea, m = eng.edges_from_adj(full, dist, inclusive=incl)
        for i64 in (True, False):
            ci, ca, cm = eng.edges_from_adj_compact(compact, A, dist, inclusive=incl, index64=i64)
            assert cm == m == e_ref.shape[1] and ci.dtype == (torch.int64 if i64 else torch.int32)
            np.testing.assert_array_equal(_np(ci), e_ref)
            np.testing.assert_array_equal(_np(ca), w_ref)
            assert torch.equal(ci.to(torch.int32), ei) and torch.equal(ca, ea)
        if m == 0:
            continue
        # an empty edge set is a result, not an error: threshold 0 -> no edge; explicit cap = 0 -> count only
        zi, za, zm = eng.edges_from_adj_compact(compact, A, 0.0)
        assert zm == 0 and tuple(zi.shape) == (2, 0) and tuple(za.shape) == (0,)
        zi, za, zm = eng.edges_from_adj_compact(compact, A, dist, inclusive=incl, cap=0)
        assert zm == m and tuple(zi.shape) == (2, 0)
        cap = max(1, m // 3)
        ci, ca, cm = eng.edges_from_adj_compact(compact, A, dist, inclusive=incl, cap=cap)
        assert cm == m and ci.shape[1] == cap
        np.testing.assert_array_equal(_np(ci), e_ref[:, :cap]); np.testing.assert_array_equal(_np(ca), w_ref[:cap])
