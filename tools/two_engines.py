"""Diagnostic: closed-loop stepping of a big-E batch (split path: k_env -> compact scratch -> k_adj_expand, fork / join around every gmpe_step) as ONE
engine vs TWO engines of half the envs each on two streams. Envs are independent (env_wrappers.py:968-975) and the RNG streams are keyed by global env id, so two handles with
env_id_base 0 / N/2 reproduce the single handle env by env; a runner that runs the policy on one half's observations while the other half steps gets the pipelines of the two
halves overlapped — one half's fill / drain under the other half's steady part — without any new entry point.
    python tools/two_engines.py c5 [envs]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe, bench
from gmpe.config import algorithmic_bytes_per_env_step
from gmpe.engine import GmpeEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
W = bench.WORKLOADS[wl]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
mk = lambda n, base: gmpe.make_config(scenario_name=W["scenario_name"], num_envs=n, num_agents=W["num_agents"], num_obstacles=W["num_obstacles"], num_walls=W["num_walls"],
                                      world_size=W["world_size"], episode_length=W["episode_length"], seed=1234, env_id_base=base)
one = GmpeEngine(mk(N, 0)); one.reset()
halves = [GmpeEngine(mk(N // 2, q * (N // 2))) for q in range(2)]
for e in halves: e.reset()
A = W["num_agents"]
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.randint(0, one.cfg.n_actions, (16, N, A), generator=g, device="cuda", dtype=torch.int32)
acts_h = [acts[:, q * (N // 2):(q + 1) * (N // 2)].contiguous() for q in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
K = 40
B = algorithmic_bytes_per_env_step(one.cfg)
def whole(k0):
    for k in range(K): one.step(acts[(k0 + k) % 16])
def two(k0):
    for k in range(K):
        for q in range(2):
            with torch.cuda.stream(streams[q]):
                halves[q].step(acts_h[q][(k0 + k) % 16])
def chained(k0):
    one.step_many(acts, K)
print("tuning", one.tuning(), flush=True)
for name, fn in (("one engine, gmpe_step per step", whole), ("two engines of N/2 on two streams", two), ("one engine, gmpe_step per step", whole), ("two engines of N/2 on two streams", two),
                 ("one engine, gmpe_step_many (open loop, chained)", chained)):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(3); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    us = dt / K * 1e6
    print("%-48s %s N=%d  %.1f us per full step  frac %.3f" % (name, wl, N, us, B * N / (us * 1e-6) / 1e9 / bench.HBM_PEAK_GBS), flush=True)
# same results: env n of the halves == env n of the whole batch after the same number of steps from the same reset
one2 = GmpeEngine(mk(N, 0)); one2.reset()
h2 = [GmpeEngine(mk(N // 2, q * (N // 2))) for q in range(2)]
for e in h2: e.reset()
for k in range(3):
    o = one2.step(acts[k]); oh = [h2[q].step(acts_h[q][k]) for q in range(2)]
torch.cuda.synchronize()
ok = all(torch.equal(getattr(o, key)[q * (N // 2):(q + 1) * (N // 2)], getattr(oh[q], key)) for q in range(2) for key in ("obs", "node_obs", "adj", "reward", "done"))
print("two half engines == one engine, bit for bit:", ok, flush=True)
for e in [one, one2] + halves + h2: e.check_errors()
