"""Diagnostic: per-phase cycle shares of gmpe::k_env from the -DGMPE_STAMPS build (not a benchmark)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gmpe
from gmpe import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgmpe_stamps.so")
from gmpe.engine import GmpeEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
scen = "navigation_graph" if wl == "c2" else "nav_metered_one_goal_graph_rotate_tube_july"
cfg = gmpe.make_config(scenario_name=scen, num_envs=4096, num_agents=10, seed=1234)
eng = GmpeEngine(cfg)
eng.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.randint(0, cfg.n_actions, (40, 4096, 10), generator=g, device="cuda", dtype=torch.int32)
for k in range(30):
    eng.step(acts[k])
torch.cuda.synchronize()
lib = _lib.load()
lib.gmpe_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
buf = np.zeros((8192, 16), dtype=np.uint64)
nb = lib.gmpe_debug_stamps(eng.h, buf.ctypes.data_as(C.c_void_p), 8192)
s = buf[:nb].astype(np.int64)
names = ["load", "pre-dist", "dynamics", "sync+write", "post-dist", "phase/draw", "obs/reward", "info/persist", "reset", "M", "adj", "node", "obs+id"]
d = np.diff(s[:, :13], axis=1)
print("blocks", nb, "G/BLOCK env:", os.environ.get("GMPE_G"), os.environ.get("GMPE_BLOCK"))
tot = (s[:, 12] - s[:, 0])
print("total cycles/block: median %d  p90 %d" % (np.median(tot), np.percentile(tot, 90)))
for k, nme in enumerate(names[:12]):
    print("%-14s median %7d  mean %8.0f" % (nme, np.median(d[:, k]), d[:, k].mean()))
print("span first-start..last-end: %d cycles" % (s[:, 12].max() - s[:, 0].min()))
