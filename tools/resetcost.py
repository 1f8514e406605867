#!/usr/bin/env python3
"""What does the all-env auto-reset step cost inside a rollout? (episode_length 25: step 25 of every episode resets every env of the batch)
Times rollouts of 24 steps (no reset inside) and of 1 step (the reset step) alternately, from a fresh reset."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, gmpe, bench
from gmpe.engine import GmpeEngine, StepOutputs
key = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.WORKLOADS[key]; n = wl["envs"]
cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"], num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"],
                       world_size=wl["world_size"], episode_length=25, seed=1234)
dev = torch.device("cuda", 0); eng = GmpeEngine(cfg)
g = torch.Generator(device=dev); g.manual_seed(42)
actions = torch.randint(0, cfg.n_actions, (64, n, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
T = 26; o = eng.out
keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
st = {k: torch.empty((T,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device=dev) for k in keys}
slot0 = StepOutputs(**{k: v[0] for k, v in st.items()}); strides = {k: v[0].numel() for k, v in st.items()}
eng.reset()
def timed(K):
    torch.cuda.synchronize(); eng.region_mark(0)
    eng.rollout(actions, K, slot0=slot0, num_slots=T, strides=strides)
    eng.region_mark(1); torch.cuda.synchronize(); return eng.region_ms() * 1e3
a, b, c = [], [], []
for ep in range(8):
    a.append(timed(24)); b.append(timed(1))           # steps 1..24 | step 25 = the reset step
for ep in range(8):
    c.append(timed(25))
a.sort(); b.sort(); c.sort()
print("%s: 24 steps without a reset %.1f us (%.2f per step) | the reset step alone (own launch) %.1f us | 25 steps incl. the reset %.1f us (%.2f per step) -> reset step ~ %.1f us inside a rollout"
      % (key, a[4], a[4] / 24, b[4], c[4], c[4] / 25, c[4] - a[4]))
# and a single non-reset step as its own launch, for the launch overhead
eng.reset(); d = sorted(timed(1) for _ in range(9))
print("%s: a non-reset step as its own rollout launch: %.1f us" % (key, d[4]))
