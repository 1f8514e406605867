"""Diagnostic: closed-loop stepping (ONE launch per step, every step waits for the previous one of the same envs) of a small-E batch as ONE handle vs M handles of
N/M envs each on M streams. A step launch is a dependent chain (state load, integration, distance pass) in front of a 98 MB store drain, and inside one launch the two
cannot overlap; handles of disjoint env ranges (env_id_base: same results env by env) are independent, so one handle's drain runs under another's chain — what a runner
that evaluates its policy per half-batch gets. Each handle replays its own prepared hipGraph of K step launches (gmpe_step_many_launches), so no Python sits between
the launches; the M graphs are launched back to back on M streams and the region is one HIP event pair on a common stream that forks / joins them.
    python tools/split_closed_loop.py c2 [K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmpe, bench
from gmpe.config import algorithmic_bytes_per_env_step
from gmpe.engine import GmpeEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
W = bench.WORKLOADS[wl]
N = W["envs"]
mk = lambda n, base: gmpe.make_config(scenario_name=W["scenario_name"], num_envs=n, num_agents=W["num_agents"], num_obstacles=W["num_obstacles"], num_walls=W["num_walls"],
                                      world_size=W["world_size"], episode_length=W["episode_length"], seed=1234, env_id_base=base)
A = W["num_agents"]
g = torch.Generator(device="cuda"); g.manual_seed(1)
S = 16
acts = torch.randint(0, mk(N, 0).n_actions, (S, N, A), generator=g, device="cuda", dtype=torch.int32)
B = None


def build(M):
    n = N // M
    engs = [GmpeEngine(mk(n, q * n)) for q in range(M)]
    a = [acts[:, q * n:(q + 1) * n].contiguous() for q in range(M)]
    for e, aq in zip(engs, a):
        e.reset()
        try:
            e.step_many_prepare(aq, K)
        except Exception as ex:                                              # plain launch loop instead of the graph: same kernels
            print("no hipGraph (%s)" % ex, flush=True)
    return engs, a, [torch.cuda.Stream() for _ in range(M)]


def run(engs, a, streams):
    main = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(main)
    for e, aq, s in zip(engs, a, streams):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            e.step_many_loop(aq, K)
    for s in streams:
        main.wait_stream(s)
    ev1.record(main)
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / K


sets = {M: build(M) for M in (1, 2, 4)}
B = algorithmic_bytes_per_env_step(sets[1][0][0].cfg)
print("tuning M=1", sets[1][0][0].tuning(), flush=True)
print("tuning M=2", sets[2][0][0].tuning(), flush=True)
for rep in range(3):
    for M in (1, 2, 4):
        run(*sets[M])
        us = sorted(run(*sets[M]) for _ in range(5))
        print("%s N=%d K=%d  %d handle(s) x %d envs: %.2f us per full step (min %.2f max %.2f)  frac %.3f" % (wl, N, K, M, N // M, us[2], us[0], us[-1], B * N / (us[2] * 1e-6) / 1e9 / bench.HBM_PEAK_GBS), flush=True)
# same results: env n of the parts == env n of the whole batch after the same steps from the same reset
one = GmpeEngine(mk(N, 0)); one.reset()
parts = [GmpeEngine(mk(N // 2, q * (N // 2))) for q in range(2)]
for e in parts: e.reset()
for k in range(3):
    o = one.step(acts[k]); oh = [parts[q].step(acts[k, q * (N // 2):(q + 1) * (N // 2)].contiguous()) for q in range(2)]
torch.cuda.synchronize()
ok = all(torch.equal(getattr(o, key)[q * (N // 2):(q + 1) * (N // 2)], getattr(oh[q], key)) for q in range(2) for key in ("obs", "node_obs", "adj", "reward", "done"))
print("two half handles == one handle, bit for bit:", ok, flush=True)
for M in sets:
    for e in sets[M][0]: e.check_errors()
