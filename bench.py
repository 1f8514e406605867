#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched GraphMPE step engine on MI355X.

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (decode action -> integrate -> distances -> reward/done/info ->
auto-reset -> graph observation) over one batch of 4096 environments per GPU, synthetic uniform
random actions already resident in HBM, outputs written to HBM. One process per GPU; env ranges are
sharded across ranks with NO communication inside step, so scaling is weak (4096 envs
per GPU).

The timed region is EXACTLY K steps, repeated `--reps` times (default 5); `value` is the
MEDIAN repetition (max over ranks of each). By default the K steps of a repetition are one launch of the
persistent rollout kernel (gmpe_step_many -> gmpe_rollout_steps: open-loop rollout, state carried in LDS);
`--host-loop` / `--launch-loop` time the closed-loop shape (one launch per step) instead, and the default
line carries that number too (`closed_loop`).

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1] — the configuration the headline metric is quoted on
    "c2": dict(name="navigation_graph 10 agents / 10 landmarks, 4096 vec envs, random actions",
               scenario_name="navigation_graph", num_agents=10, num_obstacles=0, num_walls=0,
               world_size=4.0, episode_length=25, envs=4096),
    # BASELINE.json configs[2]
    "c3": dict(name="nav_metered_one_goal_graph_rotate_tube_july (air_taxi) 10 agents, 4096 vec envs",
               scenario_name="nav_metered_one_goal_graph_rotate_tube_july", num_agents=10, num_obstacles=0,
               num_walls=0, world_size=4.0, episode_length=25, envs=4096),
    # the shipped-weight corridor scenario (rotated-frame float32 features, F = 7, D = 13)
    "c3r": dict(name="nav_graph_metered_single_corridor_rot_inv (air_taxi) 10 agents, 4096 vec envs",
                scenario_name="nav_graph_metered_single_corridor_rot_inv", num_agents=10, num_obstacles=0,
                num_walls=0, world_size=4.0, episode_length=25, envs=4096),
    "c3p2": dict(name="two_phase_graph (air_taxi) 10 agents, 4096 vec envs",
                 scenario_name="two_phase_graph", num_agents=10, num_obstacles=0, num_walls=0, world_size=4.0, episode_length=25, envs=4096),
    "c3p3": dict(name="three_phase_graph (air_taxi) 10 agents, 4096 vec envs",
                 scenario_name="three_phase_graph", num_agents=10, num_obstacles=0, num_walls=0, world_size=4.0, episode_length=25, envs=4096),
    # BASELINE.json configs[3] per-GPU shard and configs[4] per-GPU shard
    "c4": dict(name="navigation_graph 32 agents + 8 obstacles + 4 walls, 8192 envs sharded",
               scenario_name="navigation_graph", num_agents=32, num_obstacles=8, num_walls=4,
               world_size=8.0, episode_length=25, envs=8192),
    "c5": dict(name="navigation_graph 64 agents, 16384 envs sharded",
               scenario_name="navigation_graph", num_agents=64, num_obstacles=0, num_walls=0,
               world_size=12.0, episode_length=25, envs=16384),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
HEADLINE_METRIC = "env-steps/sec (whole node) at 4096 envs x 10 agents, navigation_graph"     # BASELINE.json `metric` (configs[1] = c2)
# Knobs that change which kernel instantiation / library runs. Performance knobs are recorded; result-changing ones are refused.
PERF_KNOBS = ("GMPE_G", "GMPE_BLOCK", "GMPE_NT", "GMPE_SPEC", "GMPE_SPLIT", "GMPE_ROLL", "GMPE_GROLL", "GMPE_CHUNKS", "GMPE_RAMP", "GMPE_AP", "GMPE_ROLLNT", "GMPE_AHEAD", "GMPE_XSTEP", "GMPE_FUSE")
DIAG_KNOBS = ("GMPE_ABLATE", "GMPE_LIB")


def host_cpu_budget():
    """(logical CPUs visible, CPUs this process may use): affinity mask capped by the cgroup CPU quota (cpu.max), which is what the
    GPU boxes enforce (16 of 128 hardware threads: profiles/r02_notes.md)."""
    seen = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = seen
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    return seen, usable, quota


def _oracle_worker(ol, gmpe, wl, n, base, acts, budget_s, barrier, result, slot):
    """One host thread: its own oracle handle over its own env shard (global env ids base .. base+n), buffers allocated once; the C call
    releases the GIL, so W threads are W cores of sequential C — the shape of the reference's W SubprocVecEnv workers (env_wrappers.py:959-1037)."""
    import ctypes as C
    import numpy as np
    cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"],
                           num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"], world_size=wl["world_size"],
                           episode_length=wl["episode_length"], seed=1234, env_id_base=base)
    orc = ol.Oracle(cfg)
    orc.reset()
    obs, ids, node, adj = orc._bufs()
    A = cfg.num_agents
    rew = np.zeros((n, A)); done = np.zeros((n, A), np.uint8); info = np.zeros((n, A, 18)); did = np.zeros(n, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    args = [p(x) for x in (obs, ids, node, adj, rew, done, info, did)]
    sets = [np.ascontiguousarray(acts[k, base:base + n]) for k in range(acts.shape[0])]
    barrier.wait()
    t0 = time.perf_counter(); steps = 0
    step, h, ns = orc.lib.gmpo_step, orc.h, len(sets)
    aps = [p(x) for x in sets]
    while True:
        for _ in range(4):                              # few interpreter round trips per C call: W threads share one GIL between calls
            step(h, aps[steps % ns], *args, 1); steps += 1
        if time.perf_counter() - t0 >= budget_s or steps >= 200000:
            break
    result[slot] = (n * steps, time.perf_counter() - t0, steps)
    orc.close()


def cpu_baseline(wl, budget_s=10.0, single_s=4.0, envs_per_worker=256):
    """The CPU oracle (C port of the reference path) timed on a bounded sample of the same workload at HOST scale: W = usable host cores
    worker threads, each stepping its own env shard (`W = host cores`, aggregate and per-core), after a single-thread
    leg. Test infrastructure used as the reported baseline only — never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading
    import numpy as np
    import gmpe
    import oracle_lib as ol
    ol.load()
    seen, W, quota = host_cpu_budget()
    n = envs_per_worker
    rng = np.random.RandomState(42)
    n_act = 25 if wl["scenario_name"] != "navigation_graph" else 5
    acts = rng.randint(0, n_act, (16, n * W, wl["num_agents"])).astype(np.int32)

    def leg(workers, secs):
        res = [None] * workers
        bar = threading.Barrier(workers)
        th = [threading.Thread(target=_oracle_worker, args=(ol, gmpe, wl, n, q * n, acts, secs, bar, res, q)) for q in range(workers)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        return sum(r[0] for r in res) / max(r[1] for r in res), res, wall
    one, r1, _ = leg(1, single_s)
    agg, rw, wall = leg(W, budget_s) if W > 1 else (one, r1, 0.0)
    return {"value": agg, "unit": "env-steps/s", "cores": W, "kind": "port",
            "per_core": agg / W, "single_thread": one, "host_cores": seen, "cpu_quota": quota,
            "sample": "%d worker threads x %d envs of the same workload, %d-%d steps each in %.1f s (aggregate over the threads' common window), after a "
                      "single-thread leg of %d steps (%.1f s); C oracle (oracle/gmpe_oracle.c, fp64 outputs, one handle per thread, the C call releases the GIL)"
                      % (W, n, min(r[2] for r in rw), max(r[2] for r in rw), budget_s, r1[0][2], single_s)}


def numpy_boundary(wl, n_envs, device, steps=30):
    """PCIe-inclusive rate of the drop-in boundary the unchanged runner uses (never `value`): BatchedGraphMPEVecEnv.step with the
    runner's float one-hot actions [N,A,n_act] in NumPy and NumPy observations out (graph_mpe_runner.py:343-382)."""
    import argparse as ap
    import numpy as np
    from gmpe.vec_env import BatchedGraphMPEVecEnv
    a = ap.Namespace(env_name="GraphMPE", scenario_name=wl["scenario_name"], dynamics_type=None, world_size=wl["world_size"],
                     num_agents=wl["num_agents"], num_landmarks=wl["num_agents"], num_scripted_agents=0, num_obstacles=wl["num_obstacles"],
                     num_walls=wl["num_walls"], collaborative=False, max_speed=2, collision_rew=5, formation_rew=1, goal_rew=5,
                     episode_length=wl["episode_length"], n_rollout_threads=n_envs, total_actions=5, graph_feat_type="relative",
                     discrete_action=True, use_safety_filter=False, seed=1234)
    env = BatchedGraphMPEVecEnv(a, num_envs=n_envs, device=device)
    env.reset()
    rng = np.random.RandomState(0)
    n_act = env.action_space[0].n
    onehot = np.eye(n_act)[rng.randint(0, n_act, (4, n_envs, wl["num_agents"]))]      # float64, exactly what the runner builds (graph_mpe_runner.py:375-377)
    for k in range(3):
        env.step(onehot[k % 4])
    per = []
    for k in range(steps):
        t0 = time.perf_counter()
        env.step(onehot[k % 4])
        per.append(time.perf_counter() - t0)
    env.close()
    per.sort()
    med = per[len(per) // 2]
    # median AND mean are reported: they agree since the action conversion is single-threaded NumPy (a 128-thread torch CPU op per step got the
    # process CPU-throttled for ~90 ms about once in 30 steps under the boxes' 16-CPU quota: profiles/r02_notes.md)
    return {"value": n_envs / med, "unit": "env-steps/s", "ms_per_step": med * 1e3, "mean_ms_per_step": sum(per) / len(per) * 1e3, "max_ms": per[-1] * 1e3,
            "steps": steps,
            "what": "median BatchedGraphMPEVecEnv.step: float64 one-hot NumPy actions in (converted into pinned staging, H2D), NumPy obs / node_obs / "
                    "adj (zero-copy broadcast of the compact matrix) / reward / done out (pinned D2H); PCIe-inclusive, never `value`"}


def verify_timed_region(eng, cfg, snap, action_of_step, K, read_step, surviving_steps, n_verify):
    """Checker for the launch the line is timed on: the CPU oracle (test infrastructure, loaded here as the checker only, after
    the timed region) restarts the first `n_verify` envs from `snap` — the engine's persistent state right before the LAST timed repetition —, replays that
    repetition's K action sets and must reproduce (a) every output of every step whose slot survives in the storage the timed launch wrote and (b) the
    engine's final state: floats within 1e-5 (north_star), integer state / dones / agent ids / RNG counters bit-exact. Returns the `verified` object;
    raises AssertionError on a mismatch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import gmpe
    import oracle_lib as ol
    from gmpe.config import FIELDS
    n = n_verify
    c = type(cfg)()
    import ctypes
    ctypes.memmove(ctypes.byref(c), ctypes.byref(cfg), ctypes.sizeof(cfg))
    c.num_envs = n                                                          # same seed / env_id_base: env ids [base, base + n) are the engine's first n envs
    orc = ol.Oracle(c)
    for k in FIELDS:
        orc.set(k, snap[k][:n])
    keep = set(surviving_steps)
    worst = {"obs": 0.0, "node_obs": 0.0, "adj": 0.0, "reward": 0.0, "info": 0.0}
    ints_exact = True
    n_checked = 0
    resets = 0
    for k in range(K):
        obs, ids, node, adj, rew, done, info, did = orc.step(action_of_step(k)[:n])
        resets += int(did.sum())
        if k not in keep:
            continue
        got = read_step(k)                                                   # dict of numpy arrays, first n envs
        n_checked += 1
        A, E = cfg.num_agents, cfg.num_entities
        adj_ref = adj if got["adj"].ndim == 3 else np.broadcast_to(adj[:, None], (n, A, E, E))
        for name, ref in (("obs", obs), ("node_obs", node), ("adj", adj_ref), ("reward", rew)):
            worst[name] = max(worst[name], float(np.abs(got[name].astype(np.float64) - ref).max()))
        assert np.array_equal(got["adj"] == 0, adj_ref == 0), "step %d: masked adjacency entries differ" % k
        if got.get("info") is not None:
            err = np.abs(got["info"].astype(np.float64) - info) / (1.0 + np.abs(info))
            worst["info"] = max(worst["info"], float(err.max()))
        ok = np.array_equal(got["done"].astype(bool), done) and np.array_equal(got["agent_id"].reshape(n, A), ids.reshape(n, A))
        ints_exact = ints_exact and ok
        assert ok, "step %d: done / agent_id differ from the oracle" % k
    fin = eng.get_state()
    state_err = 0.0
    for k, (_, dt, _) in FIELDS.items():
        a, b = fin[k][:n], orc.get(k)
        if np.issubdtype(dt, np.floating):
            state_err = max(state_err, float(np.abs(a - b).max()) if a.size else 0.0)
        else:
            same = np.array_equal(a, b)
            ints_exact = ints_exact and same
            assert same, "final state field %r differs from the oracle" % k
    max_out = max(worst["obs"], worst["node_obs"], worst["adj"], worst["reward"])
    assert max_out <= 1e-5, "outputs differ from the oracle: %r" % worst
    assert worst["info"] <= 2e-5, "info rows differ from the oracle: %r" % worst
    assert state_err <= 1e-5, "final float state differs from the oracle by %g" % state_err
    orc.close()
    return {"envs": n, "steps": K, "steps_compared": n_checked, "max_abs_err": max_out, "max_abs_err_by_output": worst, "max_abs_err_final_state": state_err,
            "ints_exact": bool(ints_exact), "auto_resets_inside": resets, "tolerance": 1e-5,
            "what": "CPU oracle restarted from the engine's state before the LAST timed repetition, fed that repetition's actions: every output of the "
                    "steps whose slots survive in the storage the timed launch wrote + the final persistent state (ints / dones / RNG counters exact)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step timed region; the median is reported (SURVEY 8d)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: workload's)")
    ap.add_argument("--adj-compact", action="store_true", help="write one ExE matrix per env instead of A copies")
    ap.add_argument("--no-info", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the NumPy-boundary (PCIe-inclusive) side measurement")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the one-launch-per-step side measurement")
    ap.add_argument("--host-loop", action="store_true", help="timed region = gmpe_step called from Python once per step (closed-loop shape)")
    ap.add_argument("--launch-loop", action="store_true", help="timed region = one launch per step enqueued by one C call (hipGraph of K kernel nodes)")
    ap.add_argument("--slots", type=int, default=None, help="rollout output slots (default: 26 = slot-per-step storage [26, ...] as DeviceRolloutBuffer keeps it, "
                    "so every step's bytes are certainly paid in HBM writes; fewer where 26 steps of outputs exceed min(75 %% of the free HBM, 200 GB) — "
                    "c4 allocates 158 GB —, one slot below 4). 1 = every step overwrites the same buffers")
    ap.add_argument("--gather", action="store_true", help="also time step + RCCL gather of the compact rollout slab to rank 0")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-region oracle check of the timed launch's outputs (`verified`)")
    ap.add_argument("--diag", action="store_true", help="allow GMPE_LIB / GMPE_ABLATE (diagnostic A/B runs; recorded in the line, never a result)")
    args = ap.parse_args()

    env_knobs = {k: os.environ[k] for k in PERF_KNOBS + DIAG_KNOBS if k in os.environ}
    bad = [k for k in DIAG_KNOBS if k in env_knobs]
    if bad and not args.diag:
        sys.exit("bench.py: %s set — these select a diagnostic library / wrong-by-construction ablations; unset them or pass --diag" % ", ".join(bad))

    import numpy as np
    import torch
    import gmpe
    from gmpe.config import algorithmic_bytes_per_env_step
    from gmpe.engine import GmpeEngine, StepOutputs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    dist = None
    # GMPE_BENCH_NCCL_SOLO=1 (tests): take the multi-rank branch with a ONE-rank RCCL group, so that the backend-"nccl" code path the driver's N > 1 runs use —
    # process group bound to the device, barriers, the on-device MAX reduction, the gathers — executes on a one-GPU box too (RCCL refuses two ranks on one device)
    solo = world == 1 and os.environ.get("GMPE_BENCH_NCCL_SOLO") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or solo:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # backend "nccl" IS RCCL on ROCm. GMPE_BENCH_REHEARSAL=1: gloo + every rank on device 0, to rehearse the
        # multi-process path on a one-GPU box (never used for reported numbers).
        rehearsal = os.environ.get("GMPE_BENCH_REHEARSAL") == "1"
        if rehearsal:
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    wl = WORKLOADS[args.workload]
    weak = args.workload in ("c2", "c3", "c3r", "c3p2", "c3p3")             # 4096 envs PER GPU; c4/c5 split a fixed total
    n_envs = args.envs or (wl["envs"] if weak else wl["envs"] // world)
    # configs[1]/[2] are quoted per GPU (4096 envs on 1 MI355X): weak scaling keeps 4096 per GPU
    mk = lambda: gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n_envs, num_agents=wl["num_agents"],
                                  num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"],
                                  world_size=wl["world_size"], episode_length=wl["episode_length"],
                                  seed=1234, env_id_base=rank * n_envs)
    cfg = mk()
    eng = GmpeEngine(cfg, device=local_rank, adj_compact=args.adj_compact, with_info=not args.no_info)
    tuning = eng.tuning()
    if tuning["diag_build"] and not args.diag:
        sys.exit("bench.py: libgmpe.so is a -DGMPE_DIAG build (ablations compiled in); rebuild with `make` or pass --diag")
    K, W, R = args.steps, args.warmup, max(1, args.reps)
    g = torch.Generator(device=dev); g.manual_seed(42 + rank)
    # synthetic uniform random actions for every step, generated on device and resident in HBM
    n_act_sets = min(K + W, 256)
    actions = torch.randint(0, cfg.n_actions, (n_act_sets, n_envs, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
    eng.reset()
    for k in range(W):
        eng.step(actions[k % n_act_sets])

    # ---- what one repetition of the timed region launches
    rollout_ok = bool(tuning["roll"]) and not tuning["split"]
    mode = "host-loop" if args.host_loop else ("launch-loop" if (args.launch_loop or not rollout_ok) else "rollout")
    o = eng.out
    out_keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
    step_bytes = sum(getattr(o, k).numel() * getattr(o, k).element_size() for k in out_keys)
    # Rollouts write slot-per-step storage [T, ...] by default (what DeviceRolloutBuffer.collect does, onpolicy/utils/graph_buffer.py:168-251: T = episode
    # length + 1 = 26). With ONE slot every step overwrites the same buffers: at c2 / c3 sizes (98 MB) the 256 MiB Infinity Cache absorbs that — the fill
    # pattern alone then "reaches" 8.3 TB/s (profiles/r02_tilebw.log), so a fraction of the 8 TB/s HBM peak would be priced against a ceiling that does
    # not bind; 26 slots are 2.5 GB per pass. At c4 (6 GB per step: DRAM traffic either way) the persistent tiles of a one-slot rollout rewrite the same
    # 0.7 MB block every step and run 29 % slower than into 26 slots (1122 vs 870 us per step, profiles/r03_notes.md) — both are reported.
    free_b = torch.cuda.mem_get_info(dev)[0]
    budget = min(int(free_b * 0.75), 200 << 30)
    n_slots = args.slots if args.slots is not None else (26 if step_bytes * 26 <= budget else int(max(1, budget // max(1, step_bytes))))
    if n_slots < 4:
        n_slots = 1
    slots = None
    if mode == "rollout" and n_slots > 1:
        slots = {k: torch.empty((n_slots,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device=dev) for k in out_keys}
    dram_certain = step_bytes * (n_slots if mode == "rollout" else 1) > (256 << 20)
    graph_ok = False
    if mode == "launch-loop":
        try:
            eng.step_many_prepare(actions, K)            # capture + instantiate the K-launch hipGraph: setup, outside the timed region
            graph_ok = True
        except Exception as e:                           # same kernels either way: without a graph gmpe_step_many loops over plain launches
            print("bench.py: hipGraph capture unavailable (%s); using the launch loop" % e, file=sys.stderr)

    # the rollout launch of the timed region, prepared once (slot views + argument structs): a repetition is then ONE C call — what a collect loop that launches
    # the same rollout every episode does (building the views in Python costs ~30 us per launch: 7 % of a 20-step region, none of it the step's)
    launch_K = None
    if mode == "rollout":
        launch_K = (eng.prepare_rollout(actions, K, slot0=StepOutputs(**{k: v[0] for k, v in slots.items()}), num_slots=n_slots, strides={k: v[0].numel() for k, v in slots.items()})
                    if slots is not None else eng.prepare_rollout(actions, K))

    def run_k_steps(kk):
        if mode == "host-loop":
            for k in range(kk):
                eng.step(actions[(W + k) % n_act_sets])
        elif mode == "rollout" and kk == K:
            launch_K()                                   # ONE launch: the persistent rollout kernel (into the slot storage, or overwriting one set of buffers)
        elif mode == "rollout" and slots is not None:
            eng.rollout(actions, kk, slot0=StepOutputs(**{k: v[0] for k, v in slots.items()}), num_slots=n_slots,
                        strides={k: v[0].numel() for k, v in slots.items()})
        elif mode == "rollout":
            eng.rollout(actions, kk)
        else:
            eng.step_many_loop(actions, kk)              # one C call, one launch per step (prepared hipGraph when available)
    # rollout: ONE launch covers the K steps; otherwise one k_env launch per step (+ k_adj_expand on the split path: priced together)
    launches_per_rep = 1 if mode == "rollout" else K

    def barrier():
        if dist is not None:
            dist.barrier()

    red_dev = dev if (dist is not None and dist.get_backend() == "nccl") else torch.device("cpu")

    def timed(fn, kk):
        """barrier + sync | HIP event on the launch stream | fn | event | sync + barrier  ->  (wall s max over ranks, event ms)"""
        barrier(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        eng.region_mark(0)
        fn(kk)
        eng.region_mark(1)
        torch.cuda.synchronize(dev); barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, eng.region_ms()

    run_k_steps(K); torch.cuda.synchronize(dev)                     # untimed: first use of this launch shape (same K: every rollout launch of a run is alike)
    do_verify = rank == 0 and not args.no_verify and not args.diag
    reps, snap = [], None
    for r in range(R):
        if r == R - 1 and do_verify:
            snap = eng.get_state()                                   # between two timed regions (both bracketed by synchronize): the state the last repetition starts from
        reps.append(timed(run_k_steps, K))
    eng.check_errors()
    verified = None
    if do_verify:
        # ---- outside every timed region: what did the last timed repetition write? (the number sits on a launch nothing else compares at this size)
        n_ver = min(n_envs, 64 if cfg.num_agents <= 16 else (16 if cfg.num_agents <= 32 else 8))
        acts_host = actions.cpu().numpy()
        act_of = (lambda k: acts_host[(W + k) % n_act_sets]) if mode == "host-loop" else (lambda k: acts_host[k % n_act_sets])
        if slots is not None:
            surviving = range(max(0, K - n_slots), K)                # step k -> slot k % n_slots (first_slot 0): the last n_slots steps survive
            read = lambda k: {key: v[k % n_slots][:n_ver].cpu().numpy() for key, v in slots.items()}
        else:
            surviving = [K - 1]                                       # every step overwrites the same buffers
            read = lambda k: {key: getattr(eng.out, key)[:n_ver].cpu().numpy() for key in out_keys}
        try:
            verified = verify_timed_region(eng, cfg, snap, act_of, K, read, surviving, n_ver)
        except AssertionError as e:
            sys.exit("bench.py: the timed launch's outputs do NOT match the CPU oracle: %s" % e)
        verified["launch"] = mode + (" into %d slots" % n_slots if slots is not None else "")
    order = sorted(range(R), key=lambda q: reps[q][0])
    med = order[(R - 1) // 2]                                        # the median repetition (lower median for even R)
    el, region_ms = reps[med]

    # ---- side measurement: the closed-loop shape (one launch per step, what a policy-in-the-loop runner gets per step)
    closed = None
    if mode == "rollout" and not args.no_closed_loop:
        kc = min(K, 300)
        try:
            eng.step_many_prepare(actions, kc)
        except Exception:
            pass
        eng.step_many_loop(actions, kc); torch.cuda.synchronize(dev)
        cl = [timed(lambda kk: eng.step_many_loop(actions, kk), kc) for _ in range(3)]
        cl_el, cl_ms = sorted(cl)[1]
        closed = {"ms_per_step": cl_ms / kc, "steps": kc, "launches": kc,
                  "what": "one k_env launch per step (hipGraph of K kernel nodes), same buffers: launch + latency chain + drain per step"}
        # the same steps with the batch cut into two env ranges, each on its own stream (a runner that double-buffers halves of the batch):
        # one half's latency chain runs under the other half's store drain
        eng.step_many_ranges(actions, kc, 2); torch.cuda.synchronize(dev)
        tr = sorted(timed(lambda kk: eng.step_many_ranges(actions, kk, 2), kc) for _ in range(3))[1]
        closed["two_ranges_ms_per_step"] = tr[1] / kc
    if mode == "launch-loop" and tuning["split"] and tuning.get("xstep") and not args.no_closed_loop:
        kc = min(K, 40)
        host = lambda kk: [eng.step(actions[(W + k) % n_act_sets]) for k in range(kk)]
        host(3); torch.cuda.synchronize(dev)
        cl = [timed(host, kc) for _ in range(3)]
        cl_el, cl_ms = sorted(cl)[1]
        closed = {"ms_per_step": cl_ms / kc, "steps": kc, "launches": kc,
                  "what": "one gmpe_step per step: the split pipeline forks from and joins into the caller's stream around EVERY step (what a policy-in-the-loop runner gets)"}
    # ---- side measurement: the same rollout overwriting ONE set of output buffers per step (gmpe_step_many's shape). At c2 / c3 sizes the 98 MB stay in
    # the Infinity Cache, so this is NOT an HBM figure (reported as env-steps/s with its algorithmic GB/s, no fraction of the HBM peak).
    one_slot = None
    if mode == "rollout" and slots is not None and not args.no_closed_loop and world == 1:
        eng.rollout(actions, K); torch.cuda.synchronize(dev)
        sl = sorted(timed(lambda kk: eng.rollout(actions, kk), K) for _ in range(3))[1]
        cache_resident = step_bytes <= (256 << 20)
        one_slot = {"ms_per_step": sl[1] / K, "steps": K, "launches": 1, "bytes_overwritten_per_step": step_bytes, "cache_resident": cache_resident,
                    "what": ("ONE launch of the rollout kernel, every step overwrites the same output buffers (ordinary stores); at this size the "
                             "writes are absorbed by the 256 MiB Infinity Cache: not an HBM-roofline figure") if cache_resident else
                            ("ONE launch of the rollout kernel, every step overwrites the same output buffers (nontemporal stores; far past the "
                             "Infinity Cache, so this is HBM traffic too): each persistent tile rewrites its own block every step")}
    # separate pass: per-launch events (isolated kernel duration incl. event overhead)
    iso = None
    if mode != "rollout" or not args.no_closed_loop:
        eng.timing(True); eng.timing_read(reset=True)
        eng.step_many_loop(actions, min(K, 200))
        torch.cuda.synchronize(dev)
        iso_ms, iso_n = eng.timing_read(reset=True)
        eng.timing(False)
        iso = iso_ms / max(1, iso_n)
    eng.check_errors()

    gather = None
    if args.gather and dist is not None:
        from gmpe.sharding import RolloutGather
        eng_c = eng if args.adj_compact else GmpeEngine(mk(), device=local_rank, adj_compact=True, with_info=not args.no_info)
        eng_c.reset()
        rg = RolloutGather(eng_c, world, dst=0, mode="gather")
        for k in range(5):
            rg.step_and_gather(actions[k % n_act_sets])
        kg = min(K, 200)
        barrier(); torch.cuda.synchronize(dev)
        tg0 = time.perf_counter()
        last = None
        for k in range(kg):                                         # depth-2 pipeline: gather of step k overlaps step k+1
            last = rg.step_and_gather_async(actions[(W + k) % n_act_sets])
        rg.wait(last); rg.wait(last ^ 1)
        torch.cuda.synchronize(dev); barrier()
        tg = torch.tensor([time.perf_counter() - tg0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather = {"value": world * n_envs * kg / float(tg.item()), "unit": "env-steps/s", "steps": kg, "slab_bytes_per_rank": rg.slab_bytes,
                  "what": "PER-STEP shape (round 3): closed-loop step (one launch) + RCCL gather of the rows-form slab (obs, node_obs rows, ExE adj, reward, done) "
                          "to rank 0, gather of step k overlapped with step k+1"}
        del rg, eng_c
        # ---- rollout granularity (round 4): ONE launch of the rollout kernel per T steps into the rank's compact slab (obs + fp64 entity table + one ExE adjacency +
        # rewards / dones / masks) and ONE gather of that slab per rollout, double-buffered; rank 0 rebuilds the node rows (gmpe_expand_node_obs) in its global arrays
        from gmpe.sharding import ShardedRolloutCollector, rollout_bytes_per_env_step
        if not (tuning["split"] or not tuning["roll"]):
            T = cfg.episode_length
            eng_t = GmpeEngine(mk(), device=local_rank, adj_compact=True, with_info=not args.no_info, node_form="table", adj_form="none")
            col = ShardedRolloutCollector(eng_t, T, world, dst=0)
            col.warmup()
            acts_T = actions[:T] if n_act_sets >= T else actions.repeat((T + n_act_sets - 1) // n_act_sets, 1, 1)[:T].contiguous()
            outbuf = {}
            for _ in range(2):
                b = col.collect_and_gather_async(acts_T)
                col.unpack(b, out=outbuf)
            n_roll = max(2, min(8, K // T + 1))
            barrier(); torch.cuda.synchronize(dev)
            tr0 = time.perf_counter()
            prev = None
            for r in range(n_roll):                                  # the gather of rollout r runs while rollout r+1 is collected; rank 0 unpacks r-1 meanwhile
                b = col.collect_and_gather_async(acts_T)
                if prev is not None:
                    col.unpack(prev, out=outbuf)
                prev = b
            col.unpack(prev, out=outbuf)
            torch.cuda.synchronize(dev); barrier()
            tr = torch.tensor([time.perf_counter() - tr0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            gather["rollout"] = {"value": world * n_envs * T * n_roll / float(tr.item()), "unit": "env-steps/s", "rollouts": n_roll, "steps_per_rollout": T,
                                 "slab_bytes_per_rank": col.slab_bytes, "bytes_per_env_step": rollout_bytes_per_env_step(cfg, T, "table"),
                                 "bytes_per_env_step_with_adjacency": rollout_bytes_per_env_step(cfg, T, "compact"),
                                 "bytes_per_env_step_rows_form": rollout_bytes_per_env_step(cfg, T, "rows"),
                                 "what": "ONE rollout-kernel launch per T steps into the rank's slab (obs + fp64 entity table + rewards / dones / masks; neither node rows nor adjacency are "
                                         "written or shipped) + ONE gather of the slab per rollout to rank 0 (two slabs alternate: the gather of rollout r overlaps the collection of rollout "
                                         "r+1), rank 0 rebuilding node_obs and the ExE adjacency of every rank's tables (gmpe_expand_node_obs / gmpe_expand_adj, bit-identical) in its global arrays"}
            del col, eng_t

    if rank == 0:
        B = algorithmic_bytes_per_env_step(cfg, adj_compact=args.adj_compact)      # SURVEY.md §8(d), for the outputs this run really writes
        env_steps_per_launch = n_envs * K // launches_per_rep
        avg_ms = region_ms / launches_per_rep
        achieved = B * env_steps_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = None, None
        for tag in ("r03", "r02", "r01"):
            pj = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, args.workload))
            if os.path.exists(pj) and traffic is None:
                try:
                    d = json.load(open(pj))
                    traffic = d.get("hbm_bytes_per_env_step", d.get("hbm_bytes_per_launch", 0) / max(1, d.get("env_steps_per_launch", wl["envs"]))) * env_steps_per_launch
                    traffic_src = "profiles/%s_pmc_%s.json: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate runs of this command (NOT measured in this run), scaled to this launch's env-steps" % (tag, args.workload)
                except Exception:
                    traffic = None
        # what this launch's store geometry reaches with no compute at all (tools/tilebw.hip via tools/fillbw_r04.sh: persistent tiles, per-tile contiguous blocks,
        # slot-per-step storage, best of plain / nontemporal stores) at the size class this launch writes per pass over its output storage; newest record first
        fill, fill_src = None, None
        pass_bytes = step_bytes * (n_slots if mode == "rollout" else 1)
        size_key = "98MB" if pass_bytes <= (256 << 20) else ("2.5GB" if pass_bytes <= (4 << 30) else ("6GB" if pass_bytes <= (16 << 30) else "158GB"))
        for tag in ("r04", "r01"):
            fj = os.path.join(ROOT, "profiles", "%s_fillbw.json" % tag)
            if fill is None and os.path.exists(fj):
                try:
                    tab = json.load(open(fj))["fill_GBps"]
                    key = size_key if size_key in tab else ("6GB" if size_key in ("2.5GB", "158GB") and "6GB" in tab else "98MB")
                    fill, fill_src = tab[key], "profiles/%s_fillbw.json[%s]" % (tag, key)
                except Exception:
                    fill = None
        sc = {"navigation_graph": 1 if cfg.num_walls > 0 else 0, "nav_metered_one_goal_graph_rotate_tube_july": 2,
              "nav_graph_metered_single_corridor_rot_inv": 3, "two_phase_graph": 4, "three_phase_graph": 5}[wl["scenario_name"]]
        ap_roll = tuning["ap"] if (tuning["block_roll"] == 256 and tuning["ap"] == 10) else 0       # gmpe_sc.hip launch_env: what fl == 2 dispatches
        gc_roll = tuning["G_roll"] if (ap_roll == 10 and tuning["G_roll"] in (4, 6)) else 0       # compile-time envs per tile (round 4)
        fl_step = 1 if (tuning["block"] == 256 and tuning["ap"] == 10 and not tuning["nt"] and tuning["spec"]) else 0
        k_step = "gmpe::k_env<%d, %d, %d, %d, %d>" % (tuning["block"], tuning["ap"], sc, fl_step, 4 if (fl_step and tuning["G"] == 4) else 0)
        kernel = {"rollout": "gmpe::k_env<%d, %d, %d, 2, %d> (persistent rollout: K steps per launch, G = %d envs per tile)" % (tuning["block_roll"], ap_roll, sc, gc_roll, tuning["G_roll"]),
                  "launch-loop": k_step + " (one launch per step)" + (" + gmpe::k_adj_expand" if tuning["split"] and not args.adj_compact else ""),
                  "host-loop": k_step + " (one launch per step, Python loop)"}[mode]
        restated = wl["scenario_name"] == "navigation_graph"
        out = {
            "metric": HEADLINE_METRIC if args.workload == "c2" and n_envs == 4096 else "env-steps/sec (whole node), " + wl["name"],
            "value": world * n_envs * K / el, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "repetitions": {"n": R, "reported": "median", "env_steps_per_s": [world * n_envs * K / r[0] for r in reps],
                            "region_ms": [r[1] for r in reps]},
            "config": {"workload": wl["name"] + (" [scenario restated from the reference's blocks: end-to-end parity unpinned, DESIGN.md §6]" if restated else ""),
                       "key": args.workload, "envs_per_gpu": n_envs,
                       "agents": cfg.num_agents, "entities": cfg.num_entities, "obs_dim": cfg.obs_dim, "node_feats": cfg.node_feats,
                       "episode_length": cfg.episode_length, "adj": "compact [N,E,E]" if args.adj_compact else "materialised [N,A,E,E]",
                       "info": not args.no_info, "state_dtype": "f64", "outputs_dtype": "f32 / i32 / u8",
                       "launch": {"rollout": "ONE launch of the persistent rollout kernel for the K steps (gmpe_rollout_steps), outputs: "
                                             + ("slot-per-step storage, step k -> slot k %% %d of [%d, ...] (%.2f GB per pass over the slots; nontemporal graph stores)" % (n_slots, n_slots, step_bytes * n_slots / 1e9)
                                                if slots is not None else "one slot (every step overwrites the same %.0f MB)" % (step_bytes / 1e6)),
                                  "launch-loop": ("ONE chunk pipeline over the K open-loop steps from one C call: k_env / k_adj_expand chunk launches of consecutive steps chained "
                                                  "by per-chunk events, fork before the first step and join after the last (gmpe_step_many on the split path)"
                                                  if tuning["split"] and tuning.get("xstep") and not args.adj_compact else
                                                  "one launch per step from one C call" + (" (hipGraph of K kernel nodes)" if graph_ok else "")),
                                  "host-loop": "one launch per step, Python loop over gmpe_step"}[mode],
                       "tuning": tuning, "env": env_knobs, "diag": bool(args.diag),
                       "sharding": "env ranges, %d per GPU, no collective in step" % n_envs},
            "roofline": {"bound": "hbm", "dram_certain": bool(dram_certain),
                         "dram_note": ("%.2f GB written per pass over the output storage: past the 256 MiB Infinity Cache, so the stores are paid in HBM writes" % (step_bytes * (n_slots if mode == "rollout" else 1) / 1e9))
                                      if dram_certain else "the launch overwrites %.0f MB that fit the 256 MiB Infinity Cache: `frac` is algorithmic bytes over the HBM peak, NOT a measured HBM fraction" % (step_bytes / 1e6),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "measured_fill_peak": fill, "measured_fill_source": fill_src, "frac_of_measured_fill": (achieved / fill) if fill else None,
                         "kernel": kernel, "avg_launch_ms": avg_ms, "launches": launches_per_rep,
                         "timing": "one HIP event pair on the launch stream around the launch(es) of the median repetition",
                         "isolated_launch_ms": iso,
                         "algorithmic_bytes_per_env_step": B, "env_steps_per_launch": env_steps_per_launch},
        }
        # launch shape of the timed region, so that lines of different rounds compare like for like (round 1: one launch per step; round 2: one-slot rollout;
        # round 3 on: rollout into slot-per-step storage) — the other shapes' rates ride along as top-level fields
        out["launch_shape"] = {"rollout": "rollout_into_%d_slots" % n_slots if slots is not None else "rollout_one_slot", "launch-loop": "one_launch_per_step",
                               "host-loop": "one_launch_per_step_python_loop"}[mode]
        if one_slot is not None:
            out["value_one_slot"] = n_envs / (one_slot["ms_per_step"] * 1e-3)
        if closed is not None:
            out["value_closed_loop"] = n_envs / (closed["ms_per_step"] * 1e-3)
        if one_slot is not None:
            one_slot["algorithmic_GBps"] = B * n_envs / (one_slot["ms_per_step"] * 1e-3) / 1e9
            if not one_slot["cache_resident"]:
                one_slot["frac"] = one_slot["algorithmic_GBps"] / HBM_PEAK_GBS
            one_slot["env_steps_per_s"] = n_envs / (one_slot["ms_per_step"] * 1e-3)
            out["one_slot"] = one_slot
        if closed is not None:
            closed["frac"] = B * n_envs / (closed["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            closed["env_steps_per_s"] = n_envs / (closed["ms_per_step"] * 1e-3)
            if "two_ranges_ms_per_step" in closed:
                closed["two_ranges_frac"] = B * n_envs / (closed["two_ranges_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["closed_loop"] = closed
        if verified is not None:
            out["verified"] = verified
        if gather is not None:
            out["with_gather"] = gather
        if world == 1 and not args.no_boundary:
            try:
                out["numpy_boundary"] = numpy_boundary(wl, n_envs, local_rank)
            except Exception as e:                      # a side measurement never takes the line down
                out["numpy_boundary"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(wl)
            out["host"] = {"logical_cpus": cb["host_cores"], "cpu_quota": cb["cpu_quota"], "cores_used_by_cpu_baseline": cb["cores"]}
            cj = os.path.join(ROOT, "profiles", "r02_cpu_calibration.json")
            if os.path.exists(cj):
                try:
                    cb["calibration"] = json.load(open(cj))
                except Exception:
                    pass
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
