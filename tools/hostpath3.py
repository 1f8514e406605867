"""Diagnostic: per-phase medians and maxima of BatchedGraphMPEVecEnv.step for the action formats a runner may pass (upload, launch, kernel, D2H, done cast, info clone).
This is the script that found the ~90 ms CPU-quota throttling stalls caused by 128-thread torch CPU ops (profiles/r02_notes.md). NOGC=1 freezes the GC, SPIN=1 polls events
instead of blocking in stream synchronize (both ruled out as causes)."""
import sys, time, argparse, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import gmpe, bench
from gmpe.vec_env import BatchedGraphMPEVecEnv, LazyInfos
wl = bench.WORKLOADS["c3"]
a = argparse.Namespace(env_name="GraphMPE", scenario_name=wl["scenario_name"], dynamics_type=None, world_size=4.0, num_agents=10, num_landmarks=10, num_scripted_agents=0,
                       num_obstacles=0, num_walls=0, collaborative=False, max_speed=2, collision_rew=5, formation_rew=1, goal_rew=5, episode_length=25,
                       n_rollout_threads=4096, total_actions=5, graph_feat_type="relative", discrete_action=True, use_safety_filter=False, seed=1)
import gc
if os.environ.get("NOGC"): gc.collect(); gc.freeze(); print("gc frozen")
env = BatchedGraphMPEVecEnv(a, num_envs=4096)
env.reset()
rng = np.random.RandomState(0)
idx = rng.randint(0, 25, (4, 4096, 10))
P = time.perf_counter
for name, oh in (("f32 onehot", np.eye(25, dtype=np.float32)[idx]), ("f64 onehot", np.eye(25)[idx]), ("i32 index", idx.astype(np.int32))):
    for k in range(3): env.step(oh[k % 4])
    rows = []
    for k in range(30):
        x = oh[k % 4]
        t0 = P(); t = env._upload(np.asarray(x), "onehot" if x.ndim == 3 else "index"); t1 = P()
        o = env.engine.step_onehot(t) if x.ndim == 3 else env.engine.step(t); t2 = P()
        if os.environ.get("SPIN"):
            ev0 = torch.cuda.Event(); ev0.record()
            while not ev0.query(): pass
        else:
            torch.cuda.current_stream().synchronize()
        t3 = P()
        h = env._host[0]
        for key in ("obs", "agent_id", "node_obs", "adj", "reward", "done"):
            h[key].copy_(getattr(o, key), non_blocking=True)
        t4 = P()
        if os.environ.get("SPIN"):
            ev = torch.cuda.Event(); ev.record()
            while not ev.query(): pass
        else:
            torch.cuda.current_stream().synchronize()
        t5 = P()
        done = h["done"].numpy().astype(bool); t6 = P()
        inf = LazyInfos(o.info.clone(), 4096, 10); t7 = P()
        rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6])
    r = np.array(rows) * 1e3
    print(name, "MEDIAN upload %.3f | launch %.3f | sync %.3f | D2H issue %.3f | D2H sync %.3f | done.astype %.3f | info clone %.3f ms" % tuple(np.median(r, axis=0)))
    print(name, "   MAX upload %.3f | launch %.3f | sync %.3f | D2H issue %.3f | D2H sync %.3f | done.astype %.3f | info clone %.3f ms" % tuple(r.max(axis=0)), "argmax iter", r.argmax(axis=0))
