"""Summarises gpurun_out/{prof,pmcW,pmcF}_<wl>/ (tools/profile.sh) into profiles/: the kernel-stats CSV rows of our kernels and
the per-launch HBM traffic JSON that bench.py copies into roofline.traffic. usage: python tools/profile.py c2 [round-tag [steps [dirtag]]]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csv.field_size_limit(1 << 30)


def one(pattern):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True))
    if not f:
        sys.exit("missing " + pattern)
    return f[-1]


def main(wl, tag="r02", steps="300", dirtag=None):
    """Summaries of one workload's three rocprofv3 passes. The dominant kernel = the one with the largest total duration among
    gmpe::k_env instantiations: the persistent rollout kernel k_env<256, 0, SC, 2> (ONE launch = `steps` steps of every env) for
    c2 / c3-shaped runs, the per-step kernel (+ gmpe::k_adj_expand on the split path) for c4 / c5."""
    import gmpe
    import bench
    K = int(steps)
    dt = dirtag or wl                                  # gpurun_out/prof_<dt>/ ... and profiles/<tag>_*_<dt>.* (e.g. c5 at two batch sizes)
    stats = one("prof_%s/**/*kernel_stats.csv" % dt)
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "gmpe::" in r["Name"]]
    out = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, dt))
    with open(out, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in keep:
            w.writerow(r)
        other = sum(float(r["Percentage"]) for r in rows if "gmpe::" not in r["Name"])
        fh.write('"(all other kernels: torch RNG / fills / copies)",,,,%.4f,,,\n' % other)
    short = lambda n: n.replace("void ", "").split("(")[0]
    fam = [r for r in keep if "k_env" in r["Name"] or "k_adj_expand" in r["Name"]]        # the step's kernels
    dom = max(fam, key=lambda r: float(r["TotalDurationNs"]))
    kenv = max([r for r in fam if "k_env" in r["Name"]], key=lambda r: float(r["TotalDurationNs"]))
    roll = re.search(r", 2(, \d+)?>", kenv["Name"]) is not None          # k_env<BLOCK, AP, SC, 2[, GC]>: the rollout instantiations
    split = any("k_adj_expand" in r["Name"] for r in fam)
    names = set(short(r["Name"]) for r in fam if short(r["Name"]) == short(kenv["Name"]) or "k_adj_expand" in r["Name"])
    vals = {}
    for c, pat in (("WRITE_SIZE", "pmcW_%s/**/*counter_collection.csv"), ("FETCH_SIZE", "pmcF_%s/**/*counter_collection.csv")):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(one(pat % dt))):
            if r["Counter_Name"] == c and short(r["Kernel_Name"]) in names:
                tot += float(r["Counter_Value"])
                n += 1 if short(r["Kernel_Name"]) == short(kenv["Name"]) else 0
        vals[c] = (tot, n)                             # summed over every launch of the step's kernels; n = launches of the k_env instantiation
    wlc = bench.WORKLOADS[wl]
    n_envs = int(os.environ.get("GMPE_PROFILE_ENVS", wlc["envs"]))
    cfg = gmpe.make_config(scenario_name=wlc["scenario_name"], num_envs=n_envs, num_agents=wlc["num_agents"],
                           num_obstacles=wlc["num_obstacles"], num_walls=wlc["num_walls"], world_size=wlc["world_size"],
                           episode_length=wlc["episode_length"])
    from gmpe.config import algorithmic_bytes_per_env_step
    # env-steps behind the summed counters: a rollout launch runs K steps of every env; a per-step launch one step of every env; a
    # launch of the split pipeline one step of one chunk (N / chunks envs). Every k_env launch of the profiled command is of that kind
    # (tools/profile.sh) except the single reset launch, which is counted like a step.
    # per-step launches: the profiled command (tools/profile.sh) runs 1 reset + 20 warm-up steps + (1 untimed + 3 timed) x K steps + the isolated-launch
    # pass of min(K, 200) steps; on the split path a step is several chunk launches (and their number differs between gmpe_step and the chained
    # gmpe_step_many), so the step count, not the launch count, prices the counters
    steps_total = 21 + 4 * K + min(K, 200)
    if not roll and not split:
        assert vals["WRITE_SIZE"][1] == steps_total, (vals["WRITE_SIZE"][1], steps_total)
    env_steps = vals["WRITE_SIZE"][1] * n_envs * K if roll else steps_total * n_envs
    hbm_total = (vals["WRITE_SIZE"][0] + 2.0 * vals["FETCH_SIZE"][0]) * 1024.0
    per_step_us = (float(kenv["AverageNs"]) / 1e3 / K) if roll else None
    d = {"workload": wl, "envs": n_envs, "dominant_kernel": short(dom["Name"]), "rollout_kernel": roll, "split_path": split,
         "kernels": [{"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                      "max_us": float(r["MaxNs"]) / 1e3, "pct_of_gpu_time": float(r["Percentage"])} for r in fam],
         "avg_launch_us_rocprof": float(dom["AverageNs"]) / 1e3, "steps_per_launch": K if roll else 1, "us_per_step_rocprof": per_step_us,
         "pmc_k_env_launches": vals["WRITE_SIZE"][1], "steps_behind_the_counters": None if roll else steps_total,
         "WRITE_SIZE_KB_total": vals["WRITE_SIZE"][0], "FETCH_SIZE_KB_total_raw": vals["FETCH_SIZE"][0],
         "correction": "WRITE_SIZE exact for 16-B/lane streaming stores; FETCH_SIZE doubled (gfx950 reports half of a coalesced "
                       "read stream) per MI355X_MICROARCH.md HBM section; separate --pmc passes",
         "env_steps_behind_the_counters": env_steps,
         "hbm_bytes_per_env_step": hbm_total / env_steps,
         "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step(cfg)}
    d["traffic_over_algorithmic"] = d["hbm_bytes_per_env_step"] / d["algorithmic_bytes_per_env_step"]
    with open(os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, dt)), "w") as fh:
        json.dump(d, fh, indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main(*sys.argv[1:])
