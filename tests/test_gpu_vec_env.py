"""The drop-in boundary driven the way the rmappo runner drives GraphSubprocVecEnv
(onpolicy/runner/shared/graph_mpe_runner.py:40-207, 213-238, 343-382)."""
import argparse

import numpy as np
import pytest

import gmpe
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _args(**over):
    d = dict(env_name="GraphMPE", scenario_name="nav_metered_one_goal_graph_rotate_tube_july", dynamics_type="air_taxi",
             world_size=4, num_agents=4, num_landmarks=4, num_scripted_agents=0, num_obstacles=0, num_walls=0,
             collaborative=False, max_speed=2, collision_rew=5, formation_rew=1, goal_rew=5, use_dones=False,
             episode_length=6, num_env_steps=10000, n_rollout_threads=32, render_episodes=None, fair_wt=1, fair_rew=1,
             formation_type="point", total_actions=5, zeroshift=5, graph_feat_type="relative", discrete_action=True,
             use_safety_filter=False, seed=11)
    d.update(over)
    return argparse.Namespace(**d)


def test_runner_loop_contract_and_oracle_parity():
    from gmpe.vec_env import make_train_env
    from onpolicy_shapes import get_shape_from_obs_space, get_shape_from_act_space
    a = _args()
    envs = make_train_env(a)
    N, A, E = a.n_rollout_threads, a.num_agents, 2 * a.num_agents
    assert envs.num_envs == N
    # base_runner.py:81-119 reads these eight lists
    assert get_shape_from_obs_space(envs.observation_space[0]) == (19,)
    assert get_shape_from_obs_space(envs.share_observation_space[0]) == (A * 19,)
    assert get_shape_from_obs_space(envs.node_observation_space[0]) == (E, 8)
    assert get_shape_from_obs_space(envs.adj_observation_space[0]) == (E, E)
    assert get_shape_from_obs_space(envs.edge_observation_space[0]) == (1,)
    assert get_shape_from_obs_space(envs.agent_id_observation_space[0]) == (1,)
    assert get_shape_from_obs_space(envs.share_agent_id_observation_space[0]) == (A,)
    assert get_shape_from_act_space(envs.action_space[0]) == 1 and envs.action_space[0].n == 25

    orc = ol.Oracle(envs.cfg)
    obs, agent_id, node_obs, adj = envs.reset()
    oo = orc.reset()
    assert obs.shape == (N, A, 19) and agent_id.shape == (N, A, 1) and node_obs.shape == (N, A, E, 8) and adj.shape == (N, A, E, E)
    assert obs.dtype == np.float32 and adj.dtype == np.float32 and agent_id.dtype == np.int32
    np.testing.assert_allclose(obs, oo[0], atol=1e-5)
    rng = np.random.RandomState(0)
    resets = 0
    for step in range(15):
        actions = rng.randint(0, 25, (N, A))
        actions_env = np.squeeze(np.eye(25)[actions], 0) if N == 0 else np.eye(25)[actions]      # runner :375-377
        obs, agent_id, node_obs, adj, rewards, dones, infos = envs.step(actions_env, step)
        oo = orc.step(actions)
        assert rewards.shape == (N, A) and dones.shape == (N, A) and dones.dtype == bool
        rewards[:, :, np.newaxis]                                                               # runner :94
        masks = np.ones((N, A, 1), dtype=np.float32); masks[dones == True] = 0                  # runner :85-90  # noqa: E712
        np.testing.assert_allclose(obs, oo[0], atol=1e-5)
        np.testing.assert_allclose(node_obs, oo[2], atol=1e-5)
        np.testing.assert_allclose(adj, np.broadcast_to(oo[3][:, None], adj.shape), atol=1e-5)
        np.testing.assert_allclose(rewards, oo[4], atol=1e-5)
        np.testing.assert_array_equal(dones, oo[5])
        resets += int(oo[7].sum())
        if step % 5 == 4:                                                                       # log_interval path
            assert len(infos) == N and len(infos[0]) == A
            for k in ("Distance_mean", "Dist_to_goal", "individual_reward", "Num_agent_collisions", "Min_time_to_goal"):
                assert k in infos[3][1]
            np.testing.assert_allclose(infos.as_array(), oo[6], rtol=2e-6, atol=2e-5)
    assert resets >= 2 * N          # episode_length 6 -> auto-resets happened and were matched
    envs.close()


def test_index_actions_and_errors():
    from gmpe.vec_env import BatchedGraphMPEVecEnv
    envs = BatchedGraphMPEVecEnv(_args(n_rollout_threads=4), num_envs=4)
    envs.reset()
    out = envs.step(np.zeros((4, 4), dtype=np.int64))
    assert len(out) == 7
    with pytest.raises(ValueError):
        envs.step(np.zeros((3, 4), dtype=np.int64))
    with pytest.raises(RuntimeError):
        envs.step_wait()
    with pytest.raises(NotImplementedError):
        envs.render()
    envs.close(); envs.close()


def test_navigation_graph_vec_env():
    from gmpe.vec_env import make_train_env
    a = _args(scenario_name="navigation_graph", dynamics_type="double_integrator", num_agents=5, num_landmarks=5,
              num_obstacles=2, num_walls=4, world_size=3, n_rollout_threads=16)
    envs = make_train_env(a)
    assert envs.action_space[0].n == 5 and envs.observation_space[0].shape == (13,)
    obs, ids, node, adj = envs.reset()
    assert node.shape == (16, 5, 12, 8) and adj.shape == (16, 5, 12, 12)
    for t in range(8):
        o = envs.step(np.eye(5)[np.random.RandomState(t).randint(0, 5, (16, 5))])
        assert np.isfinite(o[0]).all() and o[4].shape == (16, 5)
    envs.close()


@pytest.mark.parametrize("scen,D", [("nav_graph_metered_single_corridor_rot_inv", 13), ("two_phase_graph", 15), ("three_phase_graph", 15)])
def test_shipped_weight_scenarios_vec_env(scen, D):
    """The scenarios of model_weights/tube/** (SURVEY 8f rank 2): F = 7 node features, 18 info keys incl. Phase_reached."""
    from gmpe.vec_env import make_train_env
    from onpolicy_shapes import get_shape_from_obs_space
    a = _args(scenario_name=scen, n_rollout_threads=24, episode_length=5)
    envs = make_train_env(a)
    N, A, E = 24, a.num_agents, 2 * a.num_agents
    assert get_shape_from_obs_space(envs.observation_space[0]) == (D,)
    assert get_shape_from_obs_space(envs.node_observation_space[0]) == (E, 7)
    orc = ol.Oracle(envs.cfg)
    obs, agent_id, node_obs, adj = envs.reset(); oo = orc.reset()
    assert obs.shape == (N, A, D) and node_obs.shape == (N, A, E, 7)
    np.testing.assert_allclose(obs, oo[0], atol=1e-5); np.testing.assert_allclose(node_obs, oo[2], atol=1e-5)
    rng = np.random.RandomState(1)
    for step in range(12):
        actions = rng.randint(0, 25, (N, A))
        obs, agent_id, node_obs, adj, rewards, dones, infos = envs.step(np.eye(25)[actions], step)
        oo = orc.step(actions)
        np.testing.assert_allclose(obs, oo[0], atol=1e-5); np.testing.assert_allclose(node_obs, oo[2], atol=1e-5)
        np.testing.assert_allclose(rewards, oo[4], atol=1e-5); np.testing.assert_array_equal(dones, oo[5])
    assert "Phase_reached" in infos[0][0] and len(infos[0][0]) == 18
    np.testing.assert_allclose(infos.as_array(), oo[6], rtol=2e-6, atol=2e-5)
    envs.close()


def test_returned_arrays_stay_valid_for_one_more_step():
    """Pinned staging is double-buffered: what step t returned is untouched by step t+1."""
    from gmpe.vec_env import BatchedGraphMPEVecEnv
    envs = BatchedGraphMPEVecEnv(_args(n_rollout_threads=8), num_envs=8)
    envs.reset()
    a = np.zeros((8, 4), dtype=np.int64)
    o1 = envs.step(a)
    keep = [x.copy() for x in o1[:6]]
    o2 = envs.step(a + 3)
    for x, k in zip(o1[:6], keep):
        np.testing.assert_array_equal(x, k)
    assert not np.array_equal(o2[0], keep[0])
    envs2 = BatchedGraphMPEVecEnv(_args(n_rollout_threads=8), num_envs=8, pinned_host=False, adj_broadcast_view=False)
    envs2.reset()
    p1 = envs2.step(a); p2 = envs2.step(a + 3)
    for x, y in zip(o2[:6], p2[:6]):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    envs.close(); envs2.close()


def test_too_small_world_raises_through_the_drop_in():
    """Sticky device error flags reach the drop-in user: the bounded rejection sampler gives up in a world too
    small for its agents (the reference's spins forever, …_july.py:462-486) and reset() / step() / close() raise GmpeError."""
    from gmpe._lib import GmpeError
    from gmpe.vec_env import BatchedGraphMPEVecEnv
    a = _args(world_size=0.5, num_agents=8, num_landmarks=8, n_rollout_threads=8)
    envs = BatchedGraphMPEVecEnv(a, num_envs=8)
    with pytest.raises(GmpeError, match="placement gave up"):
        envs.reset()
    # through step: the flags are sticky, and an auto-reset inside a step sets them the same way
    envs2 = BatchedGraphMPEVecEnv(a, num_envs=8)
    envs2.engine.reset()                                    # engine-level reset: no host check
    with pytest.raises(GmpeError, match="placement gave up"):
        envs2.step(np.zeros((8, 8), dtype=np.int64))
    envs2.close()                                           # already reported by step(): a close() in a clean-up path must not raise it again
    assert envs2.closed
    envs.close()                                            # reported by reset()
    assert envs.closed
    envs3 = BatchedGraphMPEVecEnv(a, num_envs=8)
    envs3.engine.reset()                                    # nobody has seen the flags yet: close() is the last hand-off and raises them once
    with pytest.raises(GmpeError, match="placement gave up"):
        envs3.close()
    assert envs3.closed


@pytest.mark.parametrize("pinned", [True, False])
def test_july_global_features_have_the_reference_17_info_keys(pinned):
    """graph_feat_type='global' makes F = 7 for every scenario, but 'Phase_reached' is a key of the rot_inv family only
    (rot_inv.py:835); the July info_callback has 17 keys (…_july.py:806-828)."""
    from gmpe.vec_env import BatchedGraphMPEVecEnv
    envs = BatchedGraphMPEVecEnv(_args(graph_feat_type="global", n_rollout_threads=6), num_envs=6, pinned_host=pinned)
    assert envs.node_observation_space[0].shape == (8, 7)
    envs.reset()
    out = envs.step(np.zeros((6, 4), dtype=np.int64))
    keys = set(out[6][0][0].keys())
    assert len(keys) == 17 and "Phase_reached" not in keys and "Min_time_to_goal" in keys
    envs.close()
    rot = BatchedGraphMPEVecEnv(_args(scenario_name="two_phase_graph", graph_feat_type="global", n_rollout_threads=6), num_envs=6)
    rot.reset()
    assert len(rot.step(np.zeros((6, 4), dtype=np.int64))[6][0][0]) == 18
    rot.close()


def test_eval_surface_eight_tuple_with_reset_count():
    """GraphDummyVecEnv.step_wait returns an 8-tuple whose last element is reset_count (env_wrappers.py:920-936), unpacked by
    GMPERunner.render (graph_mpe_runner.py:621-622); train_mpe.py:36 / eval_mpe.py:36 select it for one rollout thread."""
    from gmpe.vec_env import make_eval_env, make_train_env
    a = _args(n_rollout_threads=1, n_eval_rollout_threads=1, episode_length=4)
    envs = make_eval_env(a)
    orc = ol.Oracle(envs.cfg)
    envs.reset(); orc.reset()
    rng = np.random.RandomState(5)
    counts = []
    for step in range(9):
        actions = rng.randint(0, 25, (1, 4))
        out = envs.step(np.eye(25)[actions])                # the render loop passes no episode number (graph_mpe_runner.py:621)
        oo = orc.step(actions)
        assert len(out) == 8
        obs, agent_id, node_obs, adj, rewards, dones, infos, reset_count = out
        np.testing.assert_allclose(obs, oo[0], atol=1e-5); np.testing.assert_array_equal(dones, oo[5])
        assert reset_count == int(np.all(dones))
        counts.append(reset_count)
    assert counts == [0, 0, 0, 1, 0, 0, 0, 1, 0]
    envs.close()
    tr = make_train_env(a)                                  # one rollout thread: the collect loop unpacks 7 values (graph_mpe_runner.py:83) — the 8-tuple is opt-in
    tr.reset()
    assert len(tr.step(np.zeros((1, 4), dtype=np.int64))) == 7
    tr.close()
    tr = make_train_env(a, eval_surface=True)
    tr.reset()
    assert len(tr.step(np.zeros((1, 4), dtype=np.int64))) == 8
    tr.close()
    many = make_eval_env(_args(n_eval_rollout_threads=3))
    many.reset()
    assert len(many.step(np.zeros((3, 4), dtype=np.int64))) == 7
    many.close()


def test_infos_are_double_buffered_and_stale_reads_fail_loudly():
    """No per-step device clone of the info rows: two buffers alternate, a LazyInfos read within one more step is exact, a later one raises."""
    from gmpe.vec_env import BatchedGraphMPEVecEnv
    envs = BatchedGraphMPEVecEnv(_args(n_rollout_threads=8), num_envs=8)
    ref = BatchedGraphMPEVecEnv(_args(n_rollout_threads=8), num_envs=8)
    envs.reset(); ref.reset()
    a = np.zeros((8, 4), dtype=np.int64)
    i1 = envs.step(a)[6]; r1 = ref.step(a)[6].as_array().copy()
    i2 = envs.step(a + 3)[6]; r2 = ref.step(a + 3)[6].as_array().copy()
    np.testing.assert_array_equal(i1.as_array(), r1)       # read one step late: still this step's rows
    np.testing.assert_array_equal(i2.as_array(), r2)
    assert not np.array_equal(r1, r2)
    i3 = envs.step(a)[6]
    envs.step(a); envs.step(a)
    with pytest.raises(RuntimeError, match="overwrote"):
        i3[0]
    envs.close(); ref.close()


@pytest.mark.parametrize("scen,mode", [("nav_metered_one_goal_graph_rotate_tube_july", "gather"),
                                       ("nav_graph_metered_single_corridor_rot_inv", "gather"),
                                       ("three_phase_graph", "all_gather")])
def test_rollout_gather_real_engine(scen, mode):
    """RolloutGather over RCCL with the REAL engine (world_size 1 covers slab binding, the collective call and unpack): the
    unpacked arrays are the engine's outputs bit for bit, for an F = 8 and F = 7 scenario. Replaces env_wrappers.py:996-1004."""
    import os
    import torch
    import torch.distributed as dist
    from gmpe.engine import GmpeEngine
    from gmpe.sharding import RolloutGather
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    N, A = 48, 10
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, world_size=4.0, episode_length=5, seed=33)
    e_ref, e_g = GmpeEngine(cfg, adj_compact=True), GmpeEngine(cfg, adj_compact=True)
    e_ref.reset(); e_g.reset()
    rg = RolloutGather(e_g, 1, mode=mode)
    assert rg.dims[4] == cfg.node_feats
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    for t in range(11):
        act = torch.randint(0, 25, (N, A), generator=g, device="cuda", dtype=torch.int32)
        o = e_ref.step(act)
        rg.step_and_gather(act)
        u = rg.unpack()
        for k in ("obs", "node_obs", "adj", "reward"):
            assert torch.equal(u[k], getattr(o, k)), (t, k)
        assert torch.equal(u["done"], o.done.bool()), t
    with pytest.raises(ValueError):
        RolloutGather(GmpeEngine(cfg), 1)                  # needs the compact adjacency


@pytest.fixture(scope="module", autouse=True)
def _destroy_process_group_at_exit():
    yield
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def test_example_runner_loop_runs_both_paths():
    """examples/runner_loop.py: the reference runner's collect loop over the drop-in vec env and over the device-resident buffer."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("runner_loop", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "runner_loop.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    a, b, c = mod.main(["--envs", "48", "--agents", "4", "--episode-length", "8", "--episodes", "2"])
    assert c["shapes"]["obs"] == (9, 48, 4, 19) and c["env_steps_per_s"] > 0 and np.isfinite(c["mean_step_reward"])   # policy in the loop on device tensors
    assert a["shapes"]["adj"] == (48, 4, 8, 8) and a["info_keys"] >= 17 and a["env_steps_per_s"] > 0
    assert b["shapes"]["adj"] == (9, 48, 4, 8, 8) and b["edges_last_slot"] >= 0 and b["env_steps_per_s"] > 0
