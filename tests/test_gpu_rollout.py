"""DeviceRolloutBuffer vs a NumPy restatement of GraphReplayBuffer.insert / GMPERunner.insert
(onpolicy/utils/graph_buffer.py:168-251, onpolicy/runner/shared/graph_mpe_runner.py:384-428)."""
import numpy as np
import pytest

import gmpe
import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compact", [False, True])
def test_device_rollout_matches_reference_buffer_semantics(compact):
    import torch
    from gmpe.engine import GmpeEngine
    from gmpe.rollout import DeviceRolloutBuffer
    N, A, T = 24, 4, 7
    cfg = gmpe.make_config(num_envs=N, num_agents=A, episode_length=5, seed=17)
    E, D = cfg.num_entities, cfg.obs_dim
    eng = GmpeEngine(cfg, adj_compact=compact)
    buf = DeviceRolloutBuffer(eng, T, use_centralized_V=True)
    orc = ol.Oracle(cfg)
    # numpy "GraphReplayBuffer"
    obs = np.zeros((T + 1, N, A, D), np.float32); share = np.zeros((T + 1, N, A, A * D), np.float32)
    node = np.zeros((T + 1, N, A, E, 8), np.float32); adj = np.zeros((T + 1, N, A, E, E), np.float32)
    rew = np.zeros((T, N, A, 1), np.float32); masks = np.ones((T + 1, N, A, 1), np.float32); act_m = np.ones_like(masks)
    buf.warmup(); o = orc.reset()
    obs[0] = o[0]; share[0] = np.repeat(o[0].reshape(N, 1, -1), A, 1); node[0] = o[2]; adj[0] = o[3][:, None]
    rng = np.random.RandomState(1)
    for t in range(T):
        a = rng.randint(0, 25, (N, A)).astype(np.int32)
        buf.insert_step(torch.as_tensor(a))
        o = orc.step(a)
        dones = o[5]
        obs[t + 1] = o[0]; share[t + 1] = np.repeat(o[0].reshape(N, 1, -1), A, 1); node[t + 1] = o[2]; adj[t + 1] = o[3][:, None]
        rew[t] = o[4][..., None]
        m = np.ones((N, A, 1), np.float32); m[dones] = 0; masks[t + 1] = m
        am = np.ones((N, A, 1), np.float32); am[dones] = 0; am[np.all(dones, axis=1)] = 1; act_m[t + 1] = am
    g = lambda x: x.detach().cpu().numpy()
    np.testing.assert_allclose(g(buf.obs), obs, atol=1e-5)
    np.testing.assert_allclose(g(buf.share_obs), share, atol=1e-5)
    np.testing.assert_allclose(g(buf.node_obs), node, atol=1e-5)
    np.testing.assert_allclose(g(buf.adj), adj, atol=1e-5)
    np.testing.assert_allclose(g(buf.rewards), rew, atol=1e-5)
    np.testing.assert_array_equal(g(buf.masks), masks)
    np.testing.assert_array_equal(g(buf.active_masks), act_m)
    assert g(buf.agent_id).shape == (T + 1, N, A, 1) and (g(buf.share_agent_id)[3, 5, 2] == np.arange(A)).all()
    assert buf.step == 0
    buf.after_update()
    np.testing.assert_array_equal(g(buf.obs[0]), g(buf.obs[-1]))
    np.testing.assert_array_equal(g(buf.masks[0]), masks[-1])
