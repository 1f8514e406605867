"""Diagnostic: does WHERE the c4 adjacency buffer sits decide the launch time (the per-process bimodality of c4)? One process, one engine, the adj output re-bound to
different offsets inside one big allocation and to fresh allocations; 30 launch-per-step steps each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe, bench
from gmpe.engine import GmpeEngine, StepOutputs
W = bench.WORKLOADS["c4"]
N = W["envs"]
cfg = gmpe.make_config(scenario_name=W["scenario_name"], num_envs=N, num_agents=W["num_agents"], num_obstacles=W["num_obstacles"], num_walls=W["num_walls"],
                       world_size=W["world_size"], episode_length=W["episode_length"], seed=1234)
eng = GmpeEngine(cfg)
print("tuning", eng.tuning(), flush=True)
eng.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.randint(0, cfg.n_actions, (16, N, cfg.num_agents), generator=g, device="cuda", dtype=torch.int32)
shape = tuple(eng.out.adj.shape); numel = eng.out.adj.numel()
def timed(label):
    eng.step_many_loop(acts, 5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.step_many_loop(acts, 30); e1.record(); torch.cuda.synchronize()
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r0.record(); eng.rollout(acts, 30); r1.record(); torch.cuda.synchronize()
    print("%-44s ptr %#x  launch-per-step %.1f us   rollout %.1f us" % (label, eng.out.adj.data_ptr(), e0.elapsed_time(e1) / 30 * 1e3, r0.elapsed_time(r1) / 30 * 1e3), flush=True)
base = eng.out
timed("engine's own buffer")
pad = 64 << 20
big = torch.empty(numel + pad // 4, dtype=torch.float32, device="cuda")
for off in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 16 << 20, 33 << 20):
    view = big[off // 4: off // 4 + numel].view(shape)
    eng.rebind(StepOutputs(**{k: (view if k == "adj" else getattr(base, k)) for k in StepOutputs.__slots__}))
    timed("big + %d KB" % (off >> 10))
keep = []
for q in range(4):
    keep.append(torch.empty((17 << 20) * (q + 1), dtype=torch.uint8, device="cuda"))     # perturb the allocator
    fresh = torch.empty(shape, dtype=torch.float32, device="cuda")
    eng.rebind(StepOutputs(**{k: (fresh if k == "adj" else getattr(base, k)) for k in StepOutputs.__slots__}))
    timed("fresh allocation %d" % q)
    keep.append(fresh)
