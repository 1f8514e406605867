#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched GraphMPE step engine on MI355X.

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (decode action -> integrate -> distances -> reward/done/info ->
auto-reset -> graph observation) over one batch of 4096 environments per GPU, synthetic uniform
random actions already resident in HBM. One process per GPU; env ranges are sharded across ranks
with NO communication inside step, so scaling is weak (4096 envs per GPU).

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


def cpu_baseline(wl, budget_s=12.0):
    """Time the CPU oracle (single-thread C port of the reference path) on a bounded sample of the
    same workload. Test infrastructure used as the reported baseline only — never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import gmpe
    import oracle_lib as ol
    n = 512
    cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"],
                           num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"],
                           world_size=wl["world_size"], episode_length=wl["episode_length"], seed=1234)
    orc = ol.Oracle(cfg)
    orc.reset()
    rng = np.random.RandomState(42)
    acts = rng.randint(0, cfg.n_actions, (64, n, cfg.num_agents)).astype(np.int32)
    t0 = time.perf_counter(); steps = 0
    while True:
        orc.step(acts[steps % 64]); steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 20000:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d steps of the same workload (%.1f s), single-thread C oracle "
                      "(oracle/gmpe_oracle.c, fp64 outputs)" % (n, steps, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: workload's)")
    ap.add_argument("--adj-compact", action="store_true", help="write one ExE matrix per env instead of A copies")
    ap.add_argument("--no-info", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-loop", action="store_true", help="call gmpe_step from Python once per step instead of gmpe_step_many")
    ap.add_argument("--no-graph", action="store_true", help="plain launch loop inside gmpe_step_many instead of the prepared hipGraph")
    ap.add_argument("--gather", action="store_true", help="also time step + RCCL all_gather of the compact rollout slab")
    args = ap.parse_args()

    import numpy as np
    import torch
    import gmpe
    from gmpe.engine import GmpeEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    dist = None
    if world > 1:
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
    cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n_envs, num_agents=wl["num_agents"],
                           num_obstacles=wl["num_obstacles"], num_walls=wl["num_walls"],
                           world_size=wl["world_size"], episode_length=wl["episode_length"],
                           seed=1234, env_id_base=rank * n_envs)
    eng = GmpeEngine(cfg, device=local_rank, adj_compact=args.adj_compact, with_info=not args.no_info)
    K, W = args.steps, args.warmup
    g = torch.Generator(device=dev); g.manual_seed(42 + rank)
    # synthetic uniform random actions for every step, generated on device and resident in HBM
    n_act_sets = min(K + W, 256)
    actions = torch.randint(0, cfg.n_actions, (n_act_sets, n_envs, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
    eng.reset()
    for k in range(W):
        eng.step(actions[k % n_act_sets])
    graph_ok = False
    if not args.host_loop and not args.no_graph:
        try:
            eng.step_many_prepare(actions, K)    # capture + instantiate the K-launch hipGraph: setup, outside the timed region
            graph_ok = True
        except Exception as e:                   # same kernels either way: without a graph gmpe_step_many loops over plain launches
            print("bench.py: hipGraph capture unavailable (%s); using the launch loop" % e, file=sys.stderr)
    torch.cuda.synchronize(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- timed region: EXACTLY K steps. One HIP event pair on the LAUNCH stream brackets the K launches
    # (recorded by libgmpe.so itself): average launch duration = elapsed / K. Per-launch event pairs are not
    # used here because they perturb back-to-back launches (~+5 us per step, measured); they are taken in a
    # separate short pass below for comparison with the rocprofv3 per-kernel average.
    barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng.region_mark(0)
    if args.host_loop:
        for k in range(K):
            eng.step(actions[(W + k) % n_act_sets])
    else:
        # one C call enqueues the K launches (gmpe_step_many): no Python / ctypes round trip between steps
        eng.step_many(actions, K)
    eng.region_mark(1)
    torch.cuda.synchronize(dev); barrier()
    t1 = time.perf_counter()
    region_ms = eng.region_ms()
    kern_ms, launches = region_ms, K
    # separate pass: per-launch events (isolated kernel duration incl. event overhead)
    eng.timing(True); eng.timing_read(reset=True)
    eng.step_many(actions, min(K, 200))
    torch.cuda.synchronize(dev)
    iso_ms, iso_n = eng.timing_read(reset=True)
    eng.timing(False)
    eng.check_errors()
    el = t1 - t0
    red_dev = dev if (dist is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    gather = None
    if args.gather and dist is not None:
        from gmpe.sharding import RolloutGather
        rg = RolloutGather(eng, world)
        for k in range(5):
            rg.step_and_gather(actions[k % n_act_sets])
        barrier(); torch.cuda.synchronize(dev)
        tg0 = time.perf_counter()
        for k in range(K):
            rg.step_and_gather(actions[(W + k) % n_act_sets])
        torch.cuda.synchronize(dev); barrier()
        tg = torch.tensor([time.perf_counter() - tg0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather = {"value": world * n_envs * K / float(tg.item()), "unit": "env-steps/s",
                  "what": "step + RCCL all_gather of the compact rollout slab (obs, node_obs, ExE adj, reward, done)"}

    if rank == 0:
        B = eng.bytes_per_env_step                      # SURVEY.md §8(d) algorithmic bytes per env-step
        avg_ms = kern_ms / max(1, launches)
        achieved = B * n_envs / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        pj = os.path.join(ROOT, "profiles", "r01_pmc_%s.json" % args.workload)
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        fill = None                                    # what a pure streaming-store kernel reaches on this part (tools_fillbw.hip)
        fj = os.path.join(ROOT, "profiles", "r01_fillbw.json")
        if os.path.exists(fj):
            try:
                fill = json.load(open(fj))["fill_GBps"]["98MB" if B * n_envs < (512 << 20) else "6GB"]
            except Exception:
                fill = None
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs x 10 agents, navigation_graph",
            "value": world * n_envs * K / el, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["name"], "key": args.workload, "envs_per_gpu": n_envs,
                       "agents": cfg.num_agents, "entities": cfg.num_entities, "obs_dim": cfg.obs_dim,
                       "episode_length": cfg.episode_length, "adj": "compact [N,E,E]" if args.adj_compact else "materialised [N,A,E,E]",
                       "info": not args.no_info,
                       "launch": "host loop" if args.host_loop else ("hipGraph of K kernel nodes (gmpe_step_many_prepare)" if graph_ok else "launch loop in gmpe_step_many"),
                       "sharding": "env ranges, %d per GPU, no collective in step" % n_envs},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_fill_peak": fill, "frac_of_measured_fill": (achieved / fill) if fill else None,
                         "kernel": "gmpe::k_env", "avg_launch_ms": avg_ms, "launches": launches,
                         "timing": "one HIP event pair on the launch stream around the K launches of the timed region",
                         "isolated_launch_ms": iso_ms / max(1, iso_n),
                         "algorithmic_bytes_per_env_step": B, "env_steps_per_launch": n_envs},
        }
        if gather is not None:
            out["with_gather"] = gather
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
