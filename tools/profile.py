"""Summarises gpurun_out/{prof,pmcW,pmcF}_<wl>/ (tools/profile.sh) into profiles/: the kernel-stats CSV rows of our kernels and
the per-launch HBM traffic JSON that bench.py copies into roofline.traffic. usage: python tools/profile.py c2 [round-tag]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csv.field_size_limit(1 << 30)


def one(pattern):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True))
    if not f:
        sys.exit("missing " + pattern)
    return f[-1]


def main(wl, tag="r01"):
    import gmpe
    import bench
    stats = one("prof_%s/**/*kernel_stats.csv" % wl)
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "gmpe::" in r["Name"]]
    out = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, wl))
    with open(out, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in keep:
            w.writerow(r)
        other = sum(float(r["Percentage"]) for r in rows if "gmpe::" not in r["Name"])
        fh.write('"(all other kernels: torch RNG / fills)",,,,%.4f,,,\n' % other)
    kenv = max(keep, key=lambda r: float(r["TotalDurationNs"]))
    vals = {}
    for c, pat in (("WRITE_SIZE", "pmcW_%s/**/*counter_collection.csv"), ("FETCH_SIZE", "pmcF_%s/**/*counter_collection.csv")):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(one(pat % wl))):
            if "k_env" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); n += 1
        vals[c] = (tot / n, n)
    wlc = bench.WORKLOADS[wl]
    cfg = gmpe.make_config(scenario_name=wlc["scenario_name"], num_envs=wlc["envs"], num_agents=wlc["num_agents"],
                           num_obstacles=wlc["num_obstacles"], num_walls=wlc["num_walls"], world_size=wlc["world_size"],
                           episode_length=wlc["episode_length"])
    from gmpe.config import algorithmic_bytes_per_env_step
    d = {"workload": wl, "kernel": kenv["Name"].replace("void ", "").replace("(gmpe::KParams)", ""),
         "avg_launch_us_rocprof": float(kenv["AverageNs"]) / 1e3, "calls": int(kenv["Calls"]),
         "launches": vals["WRITE_SIZE"][1],
         "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"][0], "FETCH_SIZE_KB_per_launch_raw": vals["FETCH_SIZE"][0],
         "hbm_bytes_per_launch": (vals["WRITE_SIZE"][0] + 2.0 * vals["FETCH_SIZE"][0]) * 1024.0,
         "correction": "WRITE_SIZE exact for 16-B/lane streaming stores; FETCH_SIZE doubled (gfx950 reports half of a coalesced "
                       "read stream) per MI355X_MICROARCH.md HBM section; separate --pmc passes",
         "algorithmic_bytes_per_launch": algorithmic_bytes_per_env_step(cfg) * wlc["envs"]}
    with open(os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, wl)), "w") as fh:
        json.dump(d, fh, indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main(*sys.argv[1:])
