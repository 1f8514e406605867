"""Round-2 widening of SURVEY §8's rows, each against the oracle (itself pinned by reference fixtures, tests/test_oracle_golden.py):
graph_feat_type='global' (…_july.py:1672-1691), the classic-MPE constant family of the force path (onpolicy/envs/mpe/core.py:125-130,
273-286; oracle routine pinned by force_classic.npz), the safety-filter hook slot (multiagent/core.py:692-736) and the shared-reward
shape quirk (multiagent/environment.py:1056-1061)."""
import argparse

import numpy as np
import pytest

import gmpe
import oracle_lib as ol
from test_gpu_parity import TOL, _compare_state, _compare_step, _engine, _np, _rollout_vs_oracle

pytestmark = pytest.mark.gpu
JULY = "nav_metered_one_goal_graph_rotate_tube_july"


@pytest.mark.parametrize("kw", [
    dict(scenario_name=JULY, num_envs=60, num_agents=10, world_size=4.0, episode_length=9, seed=101),
    dict(scenario_name=JULY, num_envs=17, num_agents=3, world_size=4.0, episode_length=7, seed=102),
    dict(scenario_name="navigation_graph", num_envs=33, num_agents=6, num_obstacles=3, num_walls=4, world_size=3.0, episode_length=8, seed=103),
    dict(scenario_name="navigation_graph", num_envs=40, num_agents=10, world_size=4.0, episode_length=8, seed=104),
    dict(scenario_name="nav_graph_metered_single_corridor_rot_inv", num_envs=40, num_agents=10, world_size=4.0, episode_length=8, seed=105),
    dict(scenario_name="two_phase_graph", num_envs=25, num_agents=4, world_size=3.0, episode_length=8, seed=106),
    dict(scenario_name="three_phase_graph", num_envs=25, num_agents=10, world_size=4.0, episode_length=8, seed=107),
], ids=["july-A10", "july-A3", "nav-walls", "nav-A10", "rotinv-A10", "twophase-A4", "threephase-A10"])
def test_global_graph_features_vs_oracle(kw):
    cfg = gmpe.make_config(graph_feat_type="global", **kw)
    assert cfg.node_feats == 7
    assert _rollout_vs_oracle(cfg, 26, seed=7) >= kw["num_envs"]


@pytest.mark.parametrize("form", ["line", "circle"])
@pytest.mark.parametrize("scen", [JULY, "nav_graph_metered_single_corridor_rot_inv", "two_phase_graph", "three_phase_graph"])
def test_formation_line_circle_vs_oracle(scen, form):
    """formation_type 'line' / 'circle' (…_july.py:492-495 -> custom_scenarios/utils.py:77-130, 231-267): distinct landmark positions through the
    in-kernel reset; exact-size (A = 10) and run-time-size (A = 4, more landmarks than agents) tiles, auto-resets included. The oracle's placement is pinned by
    the reference's own line / circle rollouts (tests/golden/*line*, *circle*), which test_golden_replay_on_gpu also replays on the GPU."""
    cfg = gmpe.make_config(scenario_name=scen, formation_type=form, num_envs=45, num_agents=10, world_size=4.0, episode_length=8, seed=131)
    assert cfg.formation_type == {"line": 1, "circle": 2}[form]
    assert _rollout_vs_oracle(cfg, 20, seed=11) >= 45
    cfg = gmpe.make_config(scenario_name=scen, formation_type=form, num_envs=19, num_agents=4, num_landmarks=6, world_size=2.4, episode_length=40, seed=132)
    _rollout_vs_oracle(cfg, 45, seed=12, shrink_world=True)
    lm = ol.Oracle(cfg); lm.reset()
    pts = lm.get("landmarks")[0]
    assert len({tuple(np.round(q, 9)) for q in pts}) == 6                  # distinct positions


def test_global_graph_features_are_ego_independent_except_redrawn_velocities():
    import torch
    cfg = gmpe.make_config(scenario_name=JULY, graph_feat_type="global", num_envs=64, num_agents=5, seed=9)
    eng = _engine(cfg)
    eng.reset()
    o = eng.step(torch.zeros((64, 5), dtype=torch.int32))
    n = o.node_obs
    assert n.shape == (64, 5, 10, 7)
    assert torch.equal(n[:, :, :, 2:], n[:, :1, :, 2:].expand_as(n[:, :, :, 2:]))          # pos / goal / type: world coordinates
    assert torch.equal(n[:, 0, :5, 2], torch.as_tensor(eng.get("x"), device="cuda").float())


@pytest.mark.parametrize("kw", [
    dict(num_envs=48, num_agents=6, num_obstacles=3, num_walls=4, world_size=2.5, episode_length=9, seed=111, agent_size=0.15, collider_size=0.2, agent_accel=3.0),
    dict(num_envs=30, num_agents=10, world_size=3.0, episode_length=8, seed=112, agent_size=0.15, agent_mass=2.0),
    dict(num_envs=20, num_agents=4, num_obstacles=2, world_size=2.0, episode_length=8, seed=113, total_actions=9, agent_size=0.05, collider_size=0.2, agent_accel=5.0),
], ids=["walls-accel3", "A10-mass2", "nine-actions"])
def test_classic_mpe_contact_family_vs_oracle(kw):
    """d_min = size_a + size_b, force on every collider side, F / mass, action force mass * accel, walls with the contact constants."""
    cfg = gmpe.make_config(scenario_name="navigation_graph", contact_family="classic", **kw)
    assert cfg.contact_family == 1 and cfg.contact_force == 100.0
    assert _rollout_vs_oracle(cfg, 24, seed=8) >= kw["num_envs"]


def test_classic_family_differs_from_multiagent_family():
    """Same seeds, the two constant families: the trajectories must differ once agents touch (guards against a dead flag)."""
    import torch
    kw = dict(scenario_name="navigation_graph", num_envs=64, num_agents=10, world_size=2.0, episode_length=30, seed=5)
    e0, e1 = _engine(gmpe.make_config(**kw)), _engine(gmpe.make_config(contact_family="classic", agent_size=0.25, **kw))
    e0.reset(); e1.reset()
    g = torch.Generator(); g.manual_seed(1)
    for t in range(12):
        a = torch.randint(0, 5, (64, 10), generator=g, dtype=torch.int32)
        o0, o1 = e0.step(a), e1.step(a)
    assert not torch.equal(o0.obs, o1.obs)


@pytest.mark.parametrize("scen", [JULY, "navigation_graph", "two_phase_graph"])
def test_control_override_hook_vs_oracle(scen):
    """Safety-filter slot: agents flagged in `use` integrate the given control instead of their decoded action."""
    import torch
    N, A = 40, 6
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, world_size=3.0, episode_length=9, seed=121)
    eng, orc = _engine(cfg), ol.Oracle(cfg)
    eng.reset(); orc.reset()
    rng = np.random.RandomState(3)
    ctrl_t = torch.zeros((N, A, 2), dtype=torch.float64, device="cuda")
    use_t = torch.zeros((N, A), dtype=torch.uint8, device="cuda")
    for t in range(22):
        act = rng.randint(0, cfg.n_actions, (N, A)).astype(np.int32)
        if t % 3 == 2:                                         # a step without the hook in between
            eng.set_control_override(None); orc.set_control_override(None)
        else:
            # |omega| bounded away from 0 (or exactly 0): the closed-form unicycle step divides by omega^2, so a tiny filtered omega
            # amplifies the 1-ulp differences between ocml and glibc sincos beyond the 1e-9 state bar (the discrete grid has |omega| >= 0.25)
            ctrl = rng.uniform(0.05, 0.3, (N, A, 2)) * rng.choice([-1.0, 1.0], (N, A, 2)); ctrl[rng.rand(N, A) < 0.2, 0] = 0.0
            use = (rng.rand(N, A) < 0.4).astype(np.uint8)
            ctrl_t.copy_(torch.as_tensor(ctrl)); use_t.copy_(torch.as_tensor(use))
            if t % 3 == 0:
                eng.set_control_override(ctrl_t, use_t); orc.set_control_override(ctrl, use)
            else:
                eng.set_control_override(ctrl_t); orc.set_control_override(ctrl)          # everywhere
        eo = eng.step(torch.as_tensor(act)); oo = orc.step(act)
        _compare_step(eo, oo, cfg.num_entities, A, "t=%d" % t)
        _compare_state(eng, orc, "t=%d" % t)
    # the rollout kernel honours the slot too
    eng.set_control_override(ctrl_t, use_t); orc.set_control_override(_np(ctrl_t), _np(use_t))
    acts = torch.as_tensor(rng.randint(0, cfg.n_actions, (4, N, A)).astype(np.int32), device="cuda")
    eo = eng.step_many(acts, 4)
    for k in range(4):
        oo = orc.step(_np(acts[k]))
    np.testing.assert_allclose(_np(eo.obs), oo[0], rtol=0, atol=TOL)
    _compare_state(eng, orc, "rollout")


def test_state_tensor_is_a_live_device_view():
    import torch
    cfg = gmpe.make_config(num_envs=16, num_agents=4, seed=2)
    eng = _engine(cfg)
    eng.reset()
    x = eng.state_tensor("x"); st = eng.state_tensor("status"); step = eng.state_tensor("current_step")
    assert x.shape == (16, 4) and x.dtype == torch.float64 and x.is_cuda and st.dtype == torch.uint8 and step.shape == (16,)
    np.testing.assert_array_equal(_np(x), eng.get("x"))
    eng.step(torch.zeros((16, 4), dtype=torch.int32)); torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(x), eng.get("x"))                        # same memory: sees the stepped state
    assert int(step[0]) == 1


def _args(**over):
    d = dict(env_name="GraphMPE", scenario_name=JULY, dynamics_type="air_taxi", world_size=4, num_agents=4, num_landmarks=4,
             num_scripted_agents=0, num_obstacles=0, num_walls=0, collaborative=False, max_speed=2, collision_rew=5, formation_rew=1, goal_rew=5,
             use_dones=False, episode_length=6, num_env_steps=10000, n_rollout_threads=12, render_episodes=None, fair_wt=1, fair_rew=1,
             formation_type="point", total_actions=5, zeroshift=5, graph_feat_type="relative", discrete_action=True, use_safety_filter=False, seed=11)
    d.update(over)
    return argparse.Namespace(**d)


def test_vec_env_safety_filter_slot_global_features_and_shared_reward_shape():
    import torch
    from gmpe.vec_env import BatchedGraphMPEVecEnv
    with pytest.raises(NotImplementedError):
        BatchedGraphMPEVecEnv(_args(use_safety_filter=True))               # no filter supplied: the HJ/CBF filter itself is not built
    calls = []

    def brake(engine, actions_dev):                                         # a toy filter: everybody decelerates, heading held
        calls.append(tuple(actions_dev.shape))
        ctrl = torch.zeros((engine.N, engine.A, 2), dtype=torch.float64, device=engine.device); ctrl[..., 1] = -0.005
        return ctrl, None
    env = BatchedGraphMPEVecEnv(_args(use_safety_filter=True, graph_feat_type="global", collaborative=True), safety_filter=brake)
    assert env.node_observation_space[0].shape == (8, 7)
    obs, ids, node, adj = env.reset()
    assert node.shape == (12, 4, 8, 7)
    v0 = env.engine.get("s3").copy()
    onehot = np.eye(25, dtype=np.float32)[np.full((12, 4), 24)]              # the policy asks for full acceleration
    o = env.step(onehot)
    assert calls == [(12, 4, 25)]
    assert (env.engine.get("s3") <= v0 + 1e-15).all()                       # ... and the filter's control is what was integrated
    assert o[4].shape == (12, 4, 1)                                          # shared reward: [[reward]] * n stacks to [N, A, 1]
    assert np.allclose(o[4], o[4][:, :1])                                    # ... and every agent carries the env's sum
    env.close()


@pytest.mark.parametrize("scen,kw", [("navigation_graph", dict(num_obstacles=2, num_walls=4)), (JULY, {}), ("two_phase_graph", {})])
def test_step_envs_ranges_equal_whole_batch_steps(scen, kw):
    """gmpe_step_envs / gmpe_step_many_envs: env ranges stepped separately (any order, ranges not aligned to the tile size, own streams) give
    bit-for-bit what whole-batch steps give — envs are independent (env_wrappers.py:968-975)."""
    import torch
    N, A = 53, 6
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, world_size=3.0, episode_length=7, seed=131, **kw)
    e1, e2, e3 = _engine(cfg), _engine(cfg), _engine(cfg)
    for e in (e1, e2, e3):
        e.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    acts = torch.randint(0, cfg.n_actions, (6, N, A), generator=g, device="cuda", dtype=torch.int32)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for k in range(17):
        o1 = e1.step(acts[k % 6]); e3.step(acts[k % 6])
        for (lo, hi, st) in ((30, 53, s1), (0, 7, s2), (7, 30, None)):        # out of order, unaligned, three streams
            o2 = e2.step_envs(acts[k % 6], lo, hi, st)
        torch.cuda.synchronize()
        for key in ("obs", "agent_id", "node_obs", "adj", "reward", "done", "info"):
            assert torch.equal(getattr(o1, key), getattr(o2, key)), (k, key)
    _compare_state(e1, e2, "ranges")
    for parts in (1, 2, 3, 4):
        o3 = e3.step_many_ranges(acts, 4, parts)
        for k in range(4):
            o1 = e1.step(acts[k % 6])
        torch.cuda.synchronize()
        for key in ("obs", "node_obs", "adj", "reward", "done", "info"):
            assert torch.equal(getattr(o1, key), getattr(o3, key)), (parts, key)
        _compare_state(e1, e3, "step_many_ranges parts=%d" % parts)
    with pytest.raises(gmpe._lib.GmpeError):
        e2.step_envs(acts[0], 10, 10)
    for e in (e1, e2, e3):
        e.check_errors(); e.close()
