"""The reference runner's collect loop (onpolicy/runner/shared/graph_mpe_runner.py:57-103) over the batched engine, three ways:

  (1) drop-in: `BatchedGraphMPEVecEnv` in place of `GraphSubprocVecEnv` — NumPy in, NumPy out, the runner unchanged;
  (2) device-resident, open loop: `DeviceRolloutBuffer.collect` — ONE launch writes the whole `[T+1, N, A, ...]` rollout in HBM,
      and the GNN's edge list (`process_adj`, onpolicy/algorithms/utils/gnn_new.py:329-358) is built on the device;
  (3) device-resident, policy in the loop: the policy reads slot `step` of the buffer as device tensors and `insert_step` writes slot `step + 1`
      in place — the runner change of INTEGRATION.md §7 (no NumPy one-hot up, no seven arrays down per step).

The policy is a stand-in (uniform random actions): the learner is outside this package's scope.

    python examples/runner_loop.py --envs 4096 --agents 10 --episodes 3
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe  # noqa: E402


def reference_args(n_envs, n_agents, episode_length, scenario):
    """The fields `GraphMPEEnv(args)` reads (multiagent/MPE_env.py:56-84), as the reference's train script sets them."""
    return argparse.Namespace(
        env_name="GraphMPE", scenario_name=scenario, dynamics_type="air_taxi", world_size=4, num_agents=n_agents, num_landmarks=n_agents,
        num_scripted_agents=0, num_obstacles=0, num_walls=0, collaborative=False, max_speed=2, collision_rew=5, formation_rew=1, goal_rew=5,
        use_dones=False, episode_length=episode_length, num_env_steps=10 ** 6, n_rollout_threads=n_envs, render_episodes=None, fair_wt=1,
        fair_rew=1, formation_type="point", total_actions=5, zeroshift=5, graph_feat_type="relative", discrete_action=True,
        use_safety_filter=False, seed=1)


def drop_in_loop(args, episodes):
    from gmpe.vec_env import make_train_env
    envs = make_train_env(args)
    N, A, n_act = envs.num_envs, args.num_agents, envs.action_space[0].n
    rng = np.random.RandomState(0)
    obs, agent_id, node_obs, adj = envs.reset()                                  # GMPERunner.warmup, :213-238
    eye = np.eye(n_act)
    t0, ret = time.perf_counter(), 0.0
    for ep in range(episodes):
        for step in range(args.episode_length):
            actions_env = eye[rng.randint(0, n_act, (N, A))]                     # the runner's one-hot (:375-377)
            obs, agent_id, node_obs, adj, rewards, dones, infos = envs.step(actions_env)
            ret += float(rewards.mean())
    dt = time.perf_counter() - t0
    sample = infos[0][0]                                                         # built lazily, only when read (base_runner.py:194-290)
    envs.close()
    return dict(path="drop-in (NumPy boundary)", env_steps_per_s=N * episodes * args.episode_length / dt, mean_step_reward=ret / (episodes * args.episode_length),
                shapes=dict(obs=obs.shape, node_obs=node_obs.shape, adj=adj.shape), info_keys=len(sample))


def device_loop(args, episodes, max_edge_dist=1.0):
    import torch
    from gmpe.engine import GmpeEngine
    from gmpe.rollout import DeviceRolloutBuffer
    from gmpe.config import config_from_args
    cfg = config_from_args(args)
    eng = GmpeEngine(cfg, adj_compact=True)                                      # one [N,E,E] matrix per env; buf.adj is the broadcast view
    buf = DeviceRolloutBuffer(eng, args.episode_length)
    buf.warmup()
    g = torch.Generator(device=eng.device); g.manual_seed(0)
    T, N, A = args.episode_length, cfg.num_envs, cfg.num_agents
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_edges = 0
    for ep in range(episodes):
        actions = torch.randint(0, cfg.n_actions, (T, N, A), generator=g, device=eng.device, dtype=torch.int32)
        buf.collect(actions)                                                     # T steps, ONE launch, every slot written in place
        ei, ew, m = eng.edges_from_adj_compact(buf._adj[T], A, max_edge_dist)    # the learner's int64 edge set for the last slot's N*A graphs
        n_edges = int(m)
        buf.after_update()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dict(path="device-resident rollout buffer", env_steps_per_s=N * episodes * T / dt, mean_step_reward=float(buf.rewards.mean()),
                shapes=dict(obs=tuple(buf.obs.shape), node_obs=tuple(buf.node_obs.shape), adj=tuple(buf.adj.shape)), edges_last_slot=n_edges,
                masks_zero=int((buf.masks == 0).sum()))


def device_closed_loop(args, episodes):
    """Policy in the loop without leaving the GPU: GMPERunner.collect's inputs are slot `step` of the device buffer, its integer actions
    go straight into `insert_step`. The stand-in policy is one linear layer on `obs` (the learner is out of scope); what matters is the data flow."""
    import torch
    from gmpe.engine import GmpeEngine
    from gmpe.rollout import DeviceRolloutBuffer
    from gmpe.config import config_from_args
    cfg = config_from_args(args)
    eng = GmpeEngine(cfg, adj_compact=True)
    buf = DeviceRolloutBuffer(eng, args.episode_length)
    buf.warmup()                                                                 # GMPERunner.warmup, :213-238
    T, N, A = args.episode_length, cfg.num_envs, cfg.num_agents
    torch.manual_seed(0)
    head = torch.nn.Linear(cfg.obs_dim, cfg.n_actions).to(eng.device)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for ep in range(episodes):
            for step in range(T):
                logits = head(buf.obs[step])                                     # [N, A, n_act] from the slot the env wrote: device tensors, no host hop
                action = torch.distributions.Categorical(logits=logits).sample() # policy.get_actions' sampled action (:346-355)
                buf.insert_step(action.to(torch.int32))                          # envs.step + buffer.insert (:82-83, :384-428): slot step+1 written in place
            buf.after_update()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dict(path="device-resident, policy in the loop", env_steps_per_s=N * episodes * T / dt, mean_step_reward=float(buf.rewards.mean()),
                shapes=dict(obs=tuple(buf.obs.shape), node_obs=tuple(buf.node_obs.shape), adj=tuple(buf.adj.shape)))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=10)
    ap.add_argument("--episode-length", type=int, default=25)
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--scenario", default="nav_metered_one_goal_graph_rotate_tube_july")
    a = ap.parse_args(argv)
    args = reference_args(a.envs, a.agents, a.episode_length, a.scenario)
    out = [drop_in_loop(args, a.episodes), device_loop(args, a.episodes), device_closed_loop(args, a.episodes)]
    for r in out:
        print(r)
    return out


if __name__ == "__main__":
    main()
