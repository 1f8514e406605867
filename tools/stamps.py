"""Diagnostic: per-phase cycle shares of gmpe::k_env from the -DGMPE_STAMPS build (not a benchmark)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe
from gmpe import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgmpe_stamps.so")
from gmpe.engine import GmpeEngine
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
W = bench.WORKLOADS[wl]
N = int(sys.argv[2]) if len(sys.argv) > 2 else min(W["envs"], 4096)
cfg = gmpe.make_config(scenario_name=W["scenario_name"], num_envs=N, num_agents=W["num_agents"], num_obstacles=W["num_obstacles"],
                       num_walls=W["num_walls"], world_size=W["world_size"], episode_length=W["episode_length"], seed=1234)
eng = GmpeEngine(cfg)
print("tuning", eng.tuning())
eng.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.randint(0, cfg.n_actions, (40, N, cfg.num_agents), generator=g, device="cuda", dtype=torch.int32)
ROLLOUT = len(sys.argv) > 3 and sys.argv[3] == "roll"      # stamps of the LAST step of a K-step rollout launch (k_env<*, 0, SC, 2>)
KROLL = int(sys.argv[4]) if len(sys.argv) > 4 else 37     # 25 (= episode_length): the stamped last step is the all-env auto-reset step
if ROLLOUT:
    eng.rollout(acts, KROLL)
else:
    for k in range(20):
        eng.step(acts[k])
torch.cuda.synchronize()
lib = _lib.load()
lib.gmpe_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
buf = np.zeros((8192, 32), dtype=np.uint64)
nb = lib.gmpe_debug_stamps(eng.h, buf.ctypes.data_as(C.c_void_p), 8192)
s = buf[:nb].astype(np.int64)
s = s[s[:, 0] > 0]                                      # split path: every chunk launch stamps blocks 0..tiles-1 of ITS grid; the last chunk's rows survive
names = ["S0 load", "S1 F-pass", "S2 dynamics", "S3 dist+static", "S4 phase/draw", "S5", "S6", "S7", "S8", "S9", "S10", "S11", "-"]
d = np.diff(s[:, :13], axis=1)
# v3 order in non-reset tiles: S0..S5 (load, F, dyn, dist, phase), then S9 (mask) S10 (adj) -> node -> S13/14/15 sec3 -> S6 -> sec4 -> S7 -> S11 obs -> S12
def seg(a, b): return np.median(s[:, b] - s[:, a])
print("timeline (median cycles): load %d | F %d | dyn %d | dist %d | phase %d | sec3 %d | sec4 %d | tail %d | total %d" % (
    seg(0,1), seg(1,2), seg(2,3), seg(3,4), seg(4,5), seg(5,6), seg(6,7), seg(7,12), seg(0,12)))
# slots 9 / 10: wave 1 (thread 64) right before / after its share of the graph stores (the issue of the stores, not their completion)
print("streaming wave 1: starts %d cycles after the phase barrier, issues its stores for %d cycles (adjacency %d | node rows %d); wave 3: %d (adjacency %d | node rows %d); "
      "wave 0 finishes sections 3+4 %d cycles after the phase barrier" % (seg(5,9), seg(9,10), seg(9,16), seg(16,10), seg(17,18), seg(17,19), seg(19,18), seg(5,7)))
print("reset-path segments (median cycles; non-reset steps: ~0): sec4 end -> reset done (S7->S8) %d | reset done -> graph stores issued (S8->S11) %d | -> end (S11->S12) %d" % (seg(7,8), seg(8,11), seg(11,12)))
print("reset section split (median cycles): sec4 end -> placement start %d | placement (reset_world_coop) %d | barrier + per-agent re-init + state write-through %d | barrier + distance pass + barrier %d | re-observation %d" % (
    seg(7,20), seg(20,21), seg(21,22), seg(22,23), seg(23,8)))
print("blocks", nb, "G/BLOCK env:", os.environ.get("GMPE_G"), os.environ.get("GMPE_BLOCK"))
tot = (s[:, 12] - s[:, 0])
print("total cycles/block: median %d  p90 %d" % (np.median(tot), np.percentile(tot, 90)))
for k, nme in enumerate(names[:12]):
    print("%-14s median %7d  mean %8.0f" % (nme, np.median(d[:, k]), d[:, k].mean()))
ok = s[:, 0] > 0
t0 = s[ok, 0].min()
sec3 = s[:, [5, 13, 14, 15, 6]]
print("section 3 split (obs | collisions | obstacle+phase+goal+clip | own-info): ", np.median(np.diff(sec3, axis=1), axis=0))
print("span first-start..last-end: %d cycles; starts p50 %d p99 %d max %d; ends p1 %d p50 %d" % (
    s[ok, 12].max() - t0, np.percentile(s[ok, 0] - t0, 50), np.percentile(s[ok, 0] - t0, 99), (s[ok, 0] - t0).max(),
    np.percentile(s[ok, 12] - t0, 1), np.percentile(s[ok, 12] - t0, 50)))
eng.timing(True); eng.timing_read()
for k in range(20): eng.step(acts[k])
ms, nl = eng.timing_read()
print("event-timed kernel: %.1f us" % (ms / nl * 1e3))
