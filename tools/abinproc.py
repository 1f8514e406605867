#!/usr/bin/env python3
"""In-process A/B of an engine knob on ONE set of slot buffers (slot rollouts spread +-5 % between processes, profiles/r03_notes.md):
    python tools/abinproc.py <workload> <KNOB> <value A> <value B> [K]
Two engines created with KNOB=A / KNOB=B (gmpe_create reads the knobs), the same [26, ...] storage and the same one-slot buffers, alternating launches; three fresh allocations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, gmpe, bench
from gmpe.engine import GmpeEngine, StepOutputs
key, knob, va, vb = sys.argv[1:5]
K = int(sys.argv[5]) if len(sys.argv) > 5 else 300
wl = bench.WORKLOADS[key]; n = wl["envs"]
cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"], num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"],
                       world_size=wl["world_size"], episode_length=25, seed=1234)
dev = torch.device("cuda", 0)
engs = {}
for v in (va, vb):
    os.environ[knob] = v
    engs["%s=%s" % (knob, v)] = GmpeEngine(cfg); engs["%s=%s" % (knob, v)].reset()
os.environ.pop(knob, None)
g = torch.Generator(device=dev); g.manual_seed(42)
actions = torch.randint(0, cfg.n_actions, (64, n, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
T = 26; o = next(iter(engs.values())).out
keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
for e in engs.values():
    e.rebind(o)                                           # both engines write the same one-slot buffers too
for alloc in range(3):
    st = {k: torch.empty((T,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device=dev) for k in keys}
    slot0 = StepOutputs(**{k: v[0] for k, v in st.items()}); strides = {k: v[0].numel() for k, v in st.items()}
    res = {k: {"slots26": [], "one_slot": []} for k in engs}
    for rep in range(4):
        for name, e in engs.items():
            for shape in ("slots26", "one_slot"):
                torch.cuda.synchronize(); e.region_mark(0)
                if shape == "slots26": e.rollout(actions, K, slot0=slot0, num_slots=T, strides=strides)
                else: e.rollout(actions, K)
                e.region_mark(1); torch.cuda.synchronize(); res[name][shape].append(e.region_ms() / K * 1e3)
    for name in engs:
        print(key, "allocation", alloc, name, {s: " ".join("%.2f" % x for x in v) for s, v in res[name].items()}, flush=True)
    del st, slot0
    torch.cuda.empty_cache()
