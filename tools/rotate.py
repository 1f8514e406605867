#!/usr/bin/env python3
"""Does a step run faster when consecutive steps write DIFFERENT output buffers? One gmpe_step per step (closed-loop shape), the engine re-bound to output set
k % S before step k (S = 1: every step overwrites the same buffers).   python tools/rotate.py c4 | c5 [envs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, gmpe, bench
from gmpe.engine import GmpeEngine, StepOutputs
key = sys.argv[1] if len(sys.argv) > 1 else "c4"
wl = bench.WORKLOADS[key]; n = int(sys.argv[2]) if len(sys.argv) > 2 else wl["envs"]
cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"], num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"],
                       world_size=wl["world_size"], episode_length=25, seed=1234)
dev = torch.device("cuda", 0); eng = GmpeEngine(cfg)
print(key, n, "tuning", eng.tuning(), flush=True)
g = torch.Generator(device=dev); g.manual_seed(42)
actions = torch.randint(0, cfg.n_actions, (32, n, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
o = eng.out
keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
step_bytes = sum(getattr(o, k).numel() * getattr(o, k).element_size() for k in keys)
B = eng.bytes_per_env_step
for S in [int(x) for x in os.environ.get("ROTATE_S", "1,2,3,4,8,1").split(",")]:
    if step_bytes * S > (200 << 30): continue
    sets = [o] + [StepOutputs(**{k: torch.empty_like(getattr(o, k)) for k in keys}) for _ in range(S - 1)]
    eng.rebind(sets[0]); eng.reset()
    for k in range(6):
        eng.rebind(sets[k % S]); eng.step(actions[k])
    ms = []
    for rep in range(3):
        torch.cuda.synchronize(); eng.region_mark(0)
        for k in range(30):
            eng.rebind(sets[k % S]); eng.step(actions[k % 32])
        eng.region_mark(1); torch.cuda.synchronize(); ms.append(eng.region_ms() / 30 * 1e3)
    ms.sort()
    print("%s: %d output set(s) in rotation: %.1f us per step (frac %.3f)  [%s]" % (key, S, ms[1], B * n / (ms[1] * 1e-6) / 8e12, ", ".join("%.1f" % x for x in ms)), flush=True)
    eng.rebind(o); del sets
    torch.cuda.empty_cache()
