"""Diagnostic: env-steps/s of the drop-in NumPy boundary (H2D one-hot actions + D2H observations every step)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe
from gmpe.vec_env import BatchedGraphMPEVecEnv
a = argparse.Namespace(env_name="GraphMPE", scenario_name="nav_metered_one_goal_graph_rotate_tube_july", dynamics_type="air_taxi",
                       world_size=4, num_agents=10, num_landmarks=10, num_scripted_agents=0, num_obstacles=0, num_walls=0,
                       collaborative=False, max_speed=2, collision_rew=5, formation_rew=1, goal_rew=5, use_dones=False,
                       episode_length=25, num_env_steps=10000, n_rollout_threads=4096, render_episodes=None, fair_wt=1, fair_rew=1,
                       formation_type="point", total_actions=5, zeroshift=5, graph_feat_type="relative", discrete_action=True,
                       use_safety_filter=False, seed=1)
for view in (True, False):
    envs = BatchedGraphMPEVecEnv(a, adj_broadcast_view=view)
    envs.reset()
    rng = np.random.RandomState(0)
    acts = np.eye(25, dtype=np.float32)[rng.randint(0, 25, (4096, 10))]
    for _ in range(5): envs.step(acts)
    t0 = time.perf_counter(); K = 50
    for _ in range(K):
        o = envs.step(acts)
    el = time.perf_counter() - t0
    nbytes = sum(x.nbytes for x in o[:6]) if not view else o[0].nbytes + o[1].nbytes + o[2].nbytes + o[3].nbytes // 10 + o[4].nbytes + o[5].nbytes
    print("adj_broadcast_view=%s: %.2f ms/step -> %.2f M env-steps/s (D2H %.1f MB/step, H2D %.1f MB/step)" % (view, el / K * 1e3, 4096 * K / el / 1e6, nbytes / 1e6, acts.nbytes / 1e6))
    envs.close()
