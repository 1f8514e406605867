"""Measurement for the SURVEY §8(f) rank-1 pieces: learner-side edge compaction (process_adj) and the in-place
device rollout buffer. Prints one JSON line per piece (not the headline metric; see bench.py for that)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe
from gmpe.engine import GmpeEngine
from gmpe.rollout import DeviceRolloutBuffer

N, A = 4096, 10
cfg = gmpe.make_config(num_envs=N, num_agents=A, seed=1234)     # C3
eng = GmpeEngine(cfg)
eng.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.randint(0, 25, (64, N, A), generator=g, device="cuda", dtype=torch.int32)
for k in range(30): eng.step(acts[k % 64])
adj = eng.out.adj.clone()                      # [N, A, E, E]: what GR_Actor.forward hands to process_adj
E = cfg.num_entities
for dist, incl, name in ((1.0, False, "process_adj (adj < 1.0, gnn_new.py:329-358)"), (cfg.coord_range, True, "update_graph (adj <= 4.83, _july.py:1651-1670)")):
    B = N * A
    cap = B * E * E
    ei = torch.empty((2, cap), dtype=torch.int32, device="cuda"); ea = torch.empty((cap,), dtype=torch.float32, device="cuda")
    ne = torch.zeros((1,), dtype=torch.int32, device="cuda")
    import ctypes as C
    from gmpe import _lib
    def run():
        _lib.check(eng.lib.gmpe_edges_from_adj(eng.h, adj.data_ptr(), B, E, float(dist), int(incl), ei.data_ptr(), ea.data_ptr(), cap, ne.data_ptr(), eng._stream()), "edges")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # same stream as the launches
    K = 100
    e0.record()
    for _ in range(K): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    m = int(ne.item())
    nbytes = 2 * 4 * B * E * E + 12 * m + 8 * B        # adj read twice (count + write passes), 12 B per edge, counts/offsets
    print(json.dumps({"piece": name, "graphs": B, "nodes": E, "edges": m, "ms_per_call": ms, "edges_per_s": m / (ms * 1e-3),
                      "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (ms * 1e-3) / 8e12}))
# the same edge sets from the COMPACT adjacency (one matrix per env, A id-shifted copies emitted): gmpe_edges_from_adj_compact
adjc = adj[:, 0].contiguous()                   # [N, E, E]
for dist, incl, name in ((1.0, False, "process_adj from the compact adjacency"), (cfg.coord_range, True, "update_graph from the compact adjacency")):
    for i64 in (False, True):
        cap = N * A * E * E
        ei = torch.empty((2, cap), dtype=torch.int64 if i64 else torch.int32, device="cuda"); ea = torch.empty((cap,), dtype=torch.float32, device="cuda")
        ne = torch.zeros((1,), dtype=torch.int32, device="cuda")
        def run():
            _lib.check(eng.lib.gmpe_edges_from_adj_compact(eng.h, adjc.data_ptr(), N, A, E, float(dist), int(incl), int(i64), ei.data_ptr(), ea.data_ptr(), cap,
                                                           ne.data_ptr(), eng._stream()), "edges_compact")
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 100
        e0.record()
        for _ in range(K): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        m = int(ne.item())
        nbytes = 2 * 4 * N * E * E + (20 if i64 else 12) * m + 8 * N     # compact adj read twice, 12 / 20 B per edge, counts
        print(json.dumps({"piece": name + (" (int64 ids)" if i64 else " (int32 ids)"), "graphs": N * A, "nodes": E, "edges": m, "ms_per_call": ms,
                          "edges_per_s": m / (ms * 1e-3), "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
                          "frac_of_8TBps": nbytes / (ms * 1e-3) / 8e12}))
# in-place rollout buffer: T steps written straight into [T+1, N, A, ...] slots
T = 25
buf = DeviceRolloutBuffer(eng, T)
buf.warmup()
torch.cuda.synchronize()
t0 = time.perf_counter(); R = 8
for r in range(R):
    buf.collect(acts[:T])
    buf.after_update()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(json.dumps({"piece": "DeviceRolloutBuffer.collect (ONE launch of the rollout kernel writes the T slots in place, masks included) + after_update",
                  "env_steps_per_s": N * T * R / el, "ms_per_step": el / (T * R) * 1e3}))
# in-place rollout buffer: T steps written straight into [T+1, N, A, ...] slots
T = 25
buf = DeviceRolloutBuffer(eng, T)
buf.warmup()
for k in range(T): buf.insert_step(acts[k % 64])
torch.cuda.synchronize()
t0 = time.perf_counter(); R = 8
for r in range(R):
    for k in range(T): buf.insert_step(acts[k % 64])
    buf.after_update()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(json.dumps({"piece": "DeviceRolloutBuffer.insert_step (engine writes slot t+1 in place) + after_update", "env_steps_per_s": N * T * R / el,
                  "ms_per_step": el / (T * R) * 1e3, "buffer_GB": sum(x.numel() * x.element_size() for x in (buf.obs, buf.node_obs, buf._adj, buf.agent_id, buf.rewards, buf.masks, buf.active_masks)) / 1e9}))
