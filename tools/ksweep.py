#!/usr/bin/env python3
"""value / HBM fraction vs rollout length K, per launch shape:
    python tools/ksweep.py c2 [out.json]          (on the GPU box)
Shapes: `slots` = one launch of the rollout kernel into slot-per-step storage [26, ...] (DRAM-certain writes), `one_slot` = the same launch
overwriting one set of buffers (Infinity-Cache-resident), `closed_loop` = one k_env launch per step (hipGraph of K nodes).
Each point: median of 7 repetitions of the K-step region, one HIP event pair around the region on the launch stream."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gmpe  # noqa: E402
from gmpe.config import algorithmic_bytes_per_env_step  # noqa: E402
from gmpe.engine import GmpeEngine, StepOutputs  # noqa: E402
import bench  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "c2"
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    Ks = [int(x) for x in os.environ.get("KSWEEP_K", "5,20,100,1000").split(",")]
    wl = bench.WORKLOADS[key]
    n = int(os.environ.get("KSWEEP_ENVS", wl["envs"]))
    cfg = gmpe.make_config(scenario_name=wl["scenario_name"], num_envs=n, num_agents=wl["num_agents"], num_obstacles=wl["num_obstacles"],
                           num_walls=wl["num_walls"], world_size=wl["world_size"], episode_length=wl["episode_length"], seed=1234)
    dev = torch.device("cuda", 0)
    eng = GmpeEngine(cfg, device=0)
    B = algorithmic_bytes_per_env_step(cfg)
    g = torch.Generator(device=dev); g.manual_seed(42)
    actions = torch.randint(0, cfg.n_actions, (256, n, cfg.num_agents), generator=g, device=dev, dtype=torch.int32)
    eng.reset()
    for k in range(25):
        eng.step(actions[k])
    T = int(os.environ.get("KSWEEP_SLOTS", "26"))
    o = eng.out
    keys = [k for k in StepOutputs.__slots__ if getattr(o, k) is not None]
    st = {k: torch.empty((T,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device=dev) for k in keys}
    slot0 = StepOutputs(**{k: v[0] for k, v in st.items()})
    strides = {k: v[0].numel() for k, v in st.items()}
    if os.environ.get("KSWEEP_STRIDE0") == "1":                # diagnostic: T slots that all alias slot 0 (same kernel path, no address diversity)
        strides = {k: 0 for k in strides}
    closed = os.environ.get("KSWEEP_CLOSED", "1") == "1"

    def region(fn, K, reps=int(os.environ.get("KSWEEP_REPS", "7"))):
        fn(K); torch.cuda.synchronize(dev)
        ms = []
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            eng.region_mark(0); fn(K); eng.region_mark(1)
            torch.cuda.synchronize(dev)
            ms.append(eng.region_ms())
        ms.sort()
        return ms[len(ms) // 2], ms

    res = {"workload": key, "envs": n, "algorithmic_bytes_per_env_step": B, "peak_GBps": bench.HBM_PEAK_GBS, "tuning": eng.tuning(), "points": []}
    for K in Ks:
        shapes = {
            "slots%d" % T: lambda kk: eng.rollout(actions, kk, slot0=slot0, num_slots=T, strides=strides),
            "one_slot": lambda kk: eng.rollout(actions, kk),
        }
        if closed:
            try:
                eng.step_many_prepare(actions, K)
            except Exception:
                pass
            shapes["closed_loop"] = lambda kk: eng.step_many_loop(actions, kk)
        if eng.tuning()["split"]:
            shapes = {"split_pipeline_chained": lambda kk: eng.step_many(actions, kk)}
        for name, fn in shapes.items():
            med, all_ms = region(fn, K)
            us = med / K * 1e3
            frac = B * n / (us * 1e-6) / 1e9 / bench.HBM_PEAK_GBS
            res["points"].append({"K": K, "shape": name, "us_per_step": us, "env_steps_per_s": n / (us * 1e-6), "frac": frac, "region_ms": all_ms})
            print("%s K=%-5d %-12s %7.2f us/step  frac %.3f  (min %.2f max %.2f)" % (key, K, name, us, frac, all_ms[0] / K * 1e3, all_ms[-1] / K * 1e3), flush=True)
    eng.check_errors()
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
