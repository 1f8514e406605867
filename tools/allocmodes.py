#!/usr/bin/env python3
"""Diagnostic: why do slot rollouts spread +-5 % between allocations (profiles/r03_notes.md "Spread", r04_ab_*_inproc.log: c3 16.7 vs 18.4 us per step by allocation)?
Same engine, same launch, different placement of the [26, ...] slot storage:
  separate   one torch.empty per output (what bench.py / DeviceRolloutBuffer do)
  slab       ONE allocation, arrays at 2 MiB-aligned offsets, natural slot strides
  slab_pad   ONE allocation, arrays at 2 MiB-aligned offsets, slot strides padded to multiples of 2 MiB
each with several fresh allocations (empty_cache in between); prints us per step (K = 300, best of 3 launches) and the base addresses mod 2 MiB.
    python tools/allocmodes.py c2 [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, gmpe, bench
from gmpe.engine import GmpeEngine, StepOutputs
key = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wl = bench.WORKLOADS[key]; n = wl["envs"]
cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"], num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"],
                       world_size=wl["world_size"], episode_length=25, seed=1234)
dev = torch.device("cuda", 0)
eng = GmpeEngine(cfg); eng.reset()
g = torch.Generator(device=dev); g.manual_seed(42)
actions = torch.randint(0, cfg.n_actions, (64, n, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
T = 26; o = eng.out
keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
MB2 = 2 << 20
up = lambda x, a: (x + a - 1) // a * a

def make(mode, junk):
    if mode == "separate":
        st = {k: torch.empty((T,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device=dev) for k in keys}
        return st, {k: v[0].numel() for k, v in st.items()}, [st]
    pad = mode == "slab_pad"
    sizes = {k: getattr(o, k).numel() * getattr(o, k).element_size() for k in keys}
    stride_b = {k: (up(sizes[k], MB2) if pad else sizes[k]) for k in keys}
    total = sum(up(stride_b[k] * T, MB2) for k in keys) + MB2
    slab = torch.empty(total, dtype=torch.uint8, device=dev)
    off = (-slab.data_ptr()) % MB2
    st, strides = {}, {}
    for k in keys:
        t = getattr(o, k)
        es = t.element_size()
        flat = slab[off:off + stride_b[k] * T].view(t.dtype).view(T, stride_b[k] // es)
        st[k] = flat[:, :t.numel()]                              # [T, numel] rows at stride_b apart
        strides[k] = stride_b[k] // es
        off += up(stride_b[k] * T, MB2)
    return st, strides, [slab]

for rnd in range(3):
    for mode in ("separate", "slab", "slab_pad"):
        junk = torch.empty((17 + 31 * rnd) << 20, dtype=torch.uint8, device=dev)          # perturb the allocator between rounds
        st, strides, keep = make(mode, junk)
        slot0 = StepOutputs(**{k: (v[0] if mode == "separate" else v[0].view(getattr(o, k).shape)) for k, v in st.items()})
        launch = eng.prepare_rollout(actions, K, slot0=slot0, num_slots=T, strides=strides)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); eng.region_mark(0); launch(); eng.region_mark(1); torch.cuda.synchronize()
            ts.append(eng.region_ms() / K * 1e3)
        bases = {k: hex(getattr(slot0, k).data_ptr() % MB2) for k in ("adj", "node_obs")}
        print(key, "round", rnd, "%-9s" % mode, " ".join("%.2f" % x for x in ts), "us per step | adj / node base mod 2MiB", bases, flush=True)
        del st, slot0, launch, keep, junk
        torch.cuda.empty_cache()
eng.check_errors()
