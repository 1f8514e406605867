#!/usr/bin/env python3
"""Adjacency store order of nontemporal launches at small E, decided inside ONE process on ONE set of slot buffers (process-to-process spread of slot rollouts is +-5 %):
two engines of an experiment build (-DGMPE_NTORDER_KNOB: GMPE_ABLATE=100 output order, 200 lane-keeps-a-float4), the same [26, ...] storage, alternating launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GMPE_LIB"] = os.path.join(ROOT, "contracts-marl-aam-corridors_amd", "libgmpe_knob.so")
import torch, gmpe, bench
from gmpe.engine import GmpeEngine, StepOutputs
key = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.WORKLOADS[key]; n = wl["envs"]
cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"], world_size=wl["world_size"], episode_length=25, seed=1234)
dev = torch.device("cuda", 0)
engs = {}
for name, v in (("output-order", "100"), ("lane-keeps", "200")):
    os.environ["GMPE_ABLATE"] = v
    engs[name] = GmpeEngine(cfg); engs[name].reset()
g = torch.Generator(device=dev); g.manual_seed(42)
actions = torch.randint(0, cfg.n_actions, (64, n, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
T = 26; o = engs["lane-keeps"].out
keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
for alloc in range(3):                                    # three fresh allocations of the slot storage: the placement effect
    st = {k: torch.empty((T,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device=dev) for k in keys}
    slot0 = StepOutputs(**{k: v[0] for k, v in st.items()}); strides = {k: v[0].numel() for k, v in st.items()}
    res = {k: [] for k in engs}
    for rep in range(4):
        for name, e in engs.items():
            torch.cuda.synchronize(); e.region_mark(0)
            e.rollout(actions, 300, slot0=slot0, num_slots=T, strides=strides)
            e.region_mark(1); torch.cuda.synchronize(); res[name].append(e.region_ms() / 300 * 1e3)
    print(key, "allocation", alloc, {k: ["%.2f" % x for x in v] for k, v in res.items()}, flush=True)
    del st, slot0
    torch.cuda.empty_cache()
