"""Diagnostic: event-timed k_env duration vs number of envs (fixed cost vs per-env cost)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe
from gmpe.engine import GmpeEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
scen = "navigation_graph" if wl == "c2" else "nav_metered_one_goal_graph_rotate_tube_july"
for n in (64, 256, 1024, 2048, 4096, 8192, 16384, 32768):
    cfg = gmpe.make_config(scenario_name=scen, num_envs=n, num_agents=10, seed=1234)
    eng = GmpeEngine(cfg)
    eng.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.randint(0, cfg.n_actions, (16, n, 10), generator=g, device="cuda", dtype=torch.int32)
    for k in range(20): eng.step(acts[k % 16])
    torch.cuda.synchronize()
    eng.timing(True); eng.timing_read()
    for k in range(200): eng.step(acts[k % 16])
    ms, nl = eng.timing_read()
    us = ms / nl * 1e3
    print("N=%6d  kernel %.1f us  -> %.1f M env-steps/s  frac %.3f" % (n, us, n / us, eng.bytes_per_env_step * n / (us * 1e-6) / 8e12))
    eng.close()
