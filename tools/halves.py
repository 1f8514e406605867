"""Diagnostic: closed-loop stepping of the two halves of the batch on two streams (gmpe_step_envs) vs whole-batch steps. Each half's step k+1 depends only
on that half's step k, so one half's latency chain runs under the other half's store drain."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe, bench
from gmpe.engine import GmpeEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W = bench.WORKLOADS[wl]; N = W["envs"]
cfg = gmpe.make_config(scenario_name=W["scenario_name"], num_envs=N, num_agents=W["num_agents"], num_obstacles=W["num_obstacles"], num_walls=W["num_walls"],
                       world_size=W["world_size"], episode_length=W["episode_length"], seed=1234)
eng = GmpeEngine(cfg); eng.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.randint(0, cfg.n_actions, (64, N, cfg.num_agents), generator=g, device="cuda", dtype=torch.int32)
K = 400
def whole(k0):
    for k in range(K): eng.step(acts[(k0 + k) % 64])
streams = [torch.cuda.Stream() for _ in range(parts)]
bounds = [N * q // parts for q in range(parts + 1)]
def split(k0):
    for k in range(K):
        a = acts[(k0 + k) % 64]
        for q in range(parts): eng.step_envs(a, bounds[q], bounds[q + 1], streams[q])
def split_c(k0):
    eng.step_many_ranges(acts, K, parts)
def whole_c(k0):
    eng.step_many_loop(acts, K)
for name, fn in (("whole batch, C loop", whole_c), ("%d ranges, C loop" % parts, split_c), ("whole batch, C loop", whole_c), ("%d ranges, C loop" % parts, split_c), ("whole batch, one stream", whole), ("%d ranges on %d streams" % (parts, parts), split), ("whole batch, one stream", whole), ("%d ranges on %d streams" % (parts, parts), split)):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(7); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-28s %s  %.2f us per full step  %.3e env-steps/s" % (name, wl, dt / K * 1e6, N * K / dt), flush=True)
eng.check_errors()
