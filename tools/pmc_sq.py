"""Summarise gpurun_out/sq{A,B}_<wl>/ (tools/pmc_sq.sh): SQ counters of the dominant k_env launches -> per tile-step figures."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(1 << 30)
sys.path.insert(0, ROOT)
import bench
wl = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = bench.WORKLOADS[wl]["envs"]
out = {}
for p in ("sqA", "sqB"):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s/**/*counter_collection.csv" % (p, wl)), recursive=True))
    if not f: continue
    rows = [r for r in csv.DictReader(open(f[-1])) if "k_env" in r["Kernel_Name"] and re.search(r", 2(, \d+)?>", r["Kernel_Name"])]
    # the K-step rollout launches of the timed region are the longest ones: group by dispatch, keep those with the max WAVE_CYCLES class
    by = {}
    for r in rows:
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if not by: continue
    key = "SQ_WAVE_CYCLES" if p == "sqA" else "SQ_INSTS_LDS"
    big = max(v.get(key, 0) for v in by.values())
    sel = [v for v in by.values() if v.get(key, 0) > 0.8 * big]
    for c in sel[0]:
        out[c] = sum(v[c] for v in sel) / len(sel)
    out["launches_" + p] = len(sel)
out["steps_per_launch"] = K
print(json.dumps(out, indent=1))
if "SQ_WAVE_CYCLES" in out:
    wc = out["SQ_WAVE_CYCLES"]
    print("fractions of wave-cycles: parked (WAIT_ANY) %.3f | issue-stalled (WAIT_INST_ANY) %.3f | active (ACTIVE_INST_ANY) %.3f | of which VALU %.3f" % (
        out["SQ_WAIT_ANY"] / wc, out["SQ_WAIT_INST_ANY"] / wc, out["SQ_ACTIVE_INST_ANY"] / wc, out["SQ_ACTIVE_INST_VALU"] / wc))
    print("VALU wave-instructions per env-step: %.0f (%d envs x %d steps per launch); quad-cycles per VALU instruction %.2f" % (
        out["SQ_INSTS_VALU"] / N / K, N, K, out["SQ_ACTIVE_INST_VALU"] / out["SQ_INSTS_VALU"]))
    # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): cycles of the launch = SQ_BUSY_CYCLES / 32
    cyc = out["SQ_BUSY_CYCLES"] / 32
    print("launch: %.0f shader cycles (%.0f per step); VALU pipe busy = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x launch cycles) = %.3f" % (
        cyc, cyc / K, out["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc))
