"""Golden vectors for the env-side buffer semantics, produced by RUNNING the reference in this container:

    python tests/golden/make_buffer_fixture.py        # writes tests/golden/graph_buffer_*.npz

What runs: the reference's own `GMPERunner.insert` (onpolicy/runner/shared/graph_mpe_runner.py:384-428 — masks / active_masks from the
dones, share_obs / share_agent_id) driving the reference's own `GraphReplayBuffer.insert` and `after_update`
(onpolicy/utils/graph_buffer.py:84-164, 168-283). The runner class is used UNBOUND on a plain namespace holding only the attributes
`insert` reads (constructing a real runner would build the policy); optional modules the import chain wants but this container lacks
(wandb, imageio, tensorboardX, ... — none of whose code runs here) get the same inert stubs as in ref_harness.py.
Inputs are seeded synthetic step outputs with the env's shapes; the vectors are data only (inputs + the reference's buffer contents).
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    H.install_stubs()
    _stub("wandb"); _stub("imageio"); _stub("setproctitle", setproctitle=lambda *a, **k: None)
    _stub("tensorboardX", SummaryWriter=object)
    flags = types.SimpleNamespace(FLAGS=lambda *a, **k: None)
    _stub("absl", flags=flags); _stub("absl.flags", FLAGS=lambda *a, **k: None)
    from onpolicy.utils.graph_buffer import GraphReplayBuffer
    from onpolicy.runner.shared.graph_mpe_runner import GMPERunner
    return GraphReplayBuffer, GMPERunner


def buffer_fixture(N, A, E, D, F, T, seed, centralized):
    GraphReplayBuffer, GMPERunner = load_reference()
    import gym
    Box, Discrete = gym.spaces.Box, gym.spaces.Discrete
    args = argparse.Namespace(episode_length=T, n_rollout_threads=N, hidden_size=8, recurrent_N=1, gamma=0.99, gae_lambda=0.95, use_gae=True,
                              use_popart=False, use_valuenorm=False, use_proper_time_limits=False, use_centralized_V=centralized)
    f32 = np.float32
    sp = lambda shape: Box(-np.inf, np.inf, shape, f32)
    buf = GraphReplayBuffer(args, A, sp((D,)), sp((A * D,) if centralized else (D,)), sp((E, F)), sp((1,)), sp((A,) if centralized else (1,)),
                            sp((E, E)), Discrete(25))
    runner = types.SimpleNamespace(n_rollout_threads=N, num_agents=A, recurrent_N=1, hidden_size=8, use_centralized_V=centralized, buffer=buf)
    rng = np.random.RandomState(seed)
    ids = np.tile(np.arange(A, dtype=np.int64)[None, :, None], (N, 1, 1))
    # warmup (graph_mpe_runner.py:213-238): slot 0
    obs0, node0, adj0 = rng.randn(N, A, D), rng.randn(N, A, E, F), np.abs(rng.randn(N, 1, E, E)).repeat(A, 1)
    share0 = obs0.reshape(N, -1)[:, None].repeat(A, 1) if centralized else obs0
    sid0 = ids.reshape(N, -1)[:, None].repeat(A, 1) if centralized else ids
    buf.share_obs[0] = share0.copy(); buf.obs[0] = obs0.copy(); buf.node_obs[0] = node0.copy(); buf.adj[0] = adj0.copy()
    buf.agent_id[0] = ids.copy(); buf.share_agent_id[0] = sid0.copy()
    rec = dict(N=N, A=A, E=E, D=D, F=F, T=T, centralized=centralized, obs0=obs0, node0=node0, adj0=adj0[:, 0])
    steps = {k: [] for k in ("obs", "node", "adj", "rew", "done")}
    for t in range(T):
        obs, node = rng.randn(N, A, D), rng.randn(N, A, E, F)
        adjc = np.abs(rng.randn(N, E, E)); adjc[rng.rand(N, E, E) < 0.3] = 0.0
        rew = rng.randn(N, A, 1)
        dones = rng.rand(N, A) < 0.35
        dones[t % N] = True                                   # one env with every agent done per step (active_masks stay 1 there)
        if t % 2:
            dones[(t + 1) % N] = False
        data = (obs, ids, node, adjc[:, None].repeat(A, 1), ids, rew, dones, [{}] * N,
                np.zeros((N, A, 1), f32), np.zeros((N, A, 1), f32), np.zeros((N, A, 1), f32),
                np.zeros((N, A, 1, 8), f32), np.zeros((N, A, 1, 8), f32), None)
        GMPERunner.insert(runner, data)
        for k, v in zip(("obs", "node", "adj", "rew", "done"), (obs, node, adjc, rew[..., 0], dones)):
            steps[k].append(v.copy())
    for k, v in steps.items():
        rec["in_" + k] = np.array(v)
    for k in ("obs", "share_obs", "node_obs", "adj", "agent_id", "share_agent_id", "rewards", "masks", "active_masks"):
        rec["buf_" + k] = np.array(getattr(buf, k))
    assert buf.step == 0
    buf.after_update()
    for k in ("obs", "share_obs", "node_obs", "adj", "agent_id", "share_agent_id", "masks", "active_masks"):
        rec["after0_" + k] = np.array(getattr(buf, k)[0])
    return rec


def main():
    for name, kw in (("graph_buffer_N6_A3_F8_central", dict(N=6, A=3, E=6, D=19, F=8, T=7, seed=3, centralized=True)),
                     ("graph_buffer_N5_A4_F7_decentral", dict(N=5, A=4, E=9, D=13, F=7, T=5, seed=4, centralized=False))):
        d = buffer_fixture(**kw)
        p = os.path.join(HERE, name + ".npz")
        np.savez_compressed(p, **d)
        print(p, os.path.getsize(p), "masks zeros", int((d["buf_masks"] == 0).sum()), "active zeros", int((d["buf_active_masks"] == 0).sum()),
              "buffer dtypes", d["buf_obs"].dtype, d["buf_agent_id"].dtype, d["buf_masks"].dtype)


if __name__ == "__main__":
    main()
