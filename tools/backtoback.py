#!/usr/bin/env python3
"""Is the short-rollout overhead a per-launch transient? Time 1, 2, 4, 8 back-to-back K=20 rollout launches (no sync in between) in one region."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, gmpe, bench
from gmpe.engine import GmpeEngine, StepOutputs
wl = bench.WORKLOADS["c2"]; n = wl["envs"]
cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"], world_size=wl["world_size"], episode_length=25, seed=1234)
dev = torch.device("cuda", 0); eng = GmpeEngine(cfg)
g = torch.Generator(device=dev); g.manual_seed(42)
actions = torch.randint(0, cfg.n_actions, (64, n, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
eng.reset()
T = 26; o = eng.out
keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
st = {k: torch.empty((T,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device=dev) for k in keys}
slot0 = StepOutputs(**{k: v[0] for k, v in st.items()}); strides = {k: v[0].numel() for k, v in st.items()}
K = 20
for L in (1, 2, 4, 8, 1):
    ms = []
    for rep in range(7):
        torch.cuda.synchronize()
        eng.region_mark(0)
        for _ in range(L):
            eng.rollout(actions, K, slot0=slot0, num_slots=T, strides=strides)
        eng.region_mark(1); torch.cuda.synchronize(); ms.append(eng.region_ms())
    ms.sort()
    print("launches back to back %d: %.2f us per step (median of 7; min %.2f)" % (L, ms[3] / (L * K) * 1e3, ms[0] / (L * K) * 1e3), flush=True)
# same with a busy GPU right before the region (a 300-step rollout enqueued first, not timed)
for L in (1, 2):
    ms = []
    for rep in range(7):
        torch.cuda.synchronize()
        eng.rollout(actions, 300, slot0=slot0, num_slots=T, strides=strides)
        eng.region_mark(0)
        for _ in range(L):
            eng.rollout(actions, K, slot0=slot0, num_slots=T, strides=strides)
        eng.region_mark(1); torch.cuda.synchronize(); ms.append(eng.region_ms())
    ms.sort()
    print("after a 300-step launch, %d launch(es): %.2f us per step (min %.2f)" % (L, ms[3] / (L * K) * 1e3, ms[0] / (L * K) * 1e3), flush=True)
