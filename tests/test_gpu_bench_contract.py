"""bench.py's one-line JSON contract and __graft_entry__.smoke(), exercised on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", *flags],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_default_line_has_every_contract_field():
    d = _bench("--no-cpu-baseline", "--reps", "3")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "repetitions", "closed_loop", "one_slot", "numpy_boundary"):   # cpu_baseline: next test
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "env-steps/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["metric"] == "env-steps/sec (whole node) at 4096 envs x 10 agents, navigation_graph"
    c = d["config"]
    assert c["key"] == "c2" and c["envs_per_gpu"] == 4096 and "model" not in c and "parity unpinned" in c["workload"]
    assert c["tuning"]["roll"] == 1 and c["tuning"]["diag_build"] == 0 and c["env"] == {} and c["diag"] is False
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # the headline launch writes slot-per-step storage (2.5 GB per pass: past the Infinity Cache), and names the instantiation that ran
    assert r["dram_certain"] is True and "26" in c["launch"] and r["kernel"].startswith("gmpe::k_env<256, 10, 0, 2, 4>")
    assert r["launches"] == 1 and r["env_steps_per_launch"] == 4096 * 20      # the K steps are ONE launch of the rollout kernel
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / average launch duration (HIP events on the launch stream)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_env_step"] * r["env_steps_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_env_step"] == 24010
    assert abs(d["value"] - 4096 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]
    rp = d["repetitions"]
    assert rp["n"] == 3 and len(rp["env_steps_per_s"]) == 3 and sorted(rp["env_steps_per_s"])[1] == d["value"]   # the median is reported
    assert d["closed_loop"]["launches"] == 20 and d["closed_loop"]["ms_per_step"] > 0
    assert d["one_slot"]["launches"] == 1 and d["one_slot"]["ms_per_step"] > 0 and "frac" not in d["one_slot"]     # cache-resident overwrite: no HBM fraction
    assert d["launch_shape"] == "rollout_into_26_slots" and d["value_one_slot"] > 0 and d["value_closed_loop"] > 0
    assert d["numpy_boundary"]["value"] > 1e5 and d["numpy_boundary"]["value"] < d["value"]
    assert d["value"] > 1e6                                  # BASELINE.json target on one MI355X


def test_launch_loop_and_compact_lines_price_their_own_bytes():
    d = _bench("--no-cpu-baseline", "--no-boundary", "--launch-loop", "--reps", "1")
    assert d["roofline"]["launches"] == 20 and d["roofline"]["env_steps_per_launch"] == 4096 and "closed_loop" not in d
    c = _bench("--no-cpu-baseline", "--no-boundary", "--adj-compact", "--reps", "1")
    A, E, F, D = 10, 20, 8, 13
    assert c["roofline"]["algorithmic_bytes_per_env_step"] == 4 * (E * E + A * (E * F + D + 2)) + A + 4 * A + 2 * A * 48


def test_diagnostic_knobs_are_refused_and_perf_knobs_recorded():
    env = dict(os.environ, GMPE_ABLATE="3")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "GMPE_ABLATE" in out.stderr
    env = dict(os.environ, GMPE_G="6", GMPE_ROLL="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--reps", "1", "--no-cpu-baseline", "--no-boundary"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["env"] == {"GMPE_G": "6", "GMPE_ROLL": "0"} and d["config"]["tuning"]["G"] == 6 and d["config"]["tuning"]["roll"] == 0
    assert d["roofline"]["launches"] == 5                       # no rollout kernel -> one launch per step


def test_cpu_baseline_leg_and_other_workload():
    d = _bench("--workload", "c3r", "--envs", "512", "--reps", "2", "--no-boundary")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "env-steps/s" and cb["value"] > 0 and "sample" in cb
    assert cb["cores"] <= cb["host_cores"] and cb["single_thread"] > 0 and abs(cb["per_core"] * cb["cores"] - cb["value"]) < 1e-6 * cb["value"]
    assert d["host"]["logical_cpus"] == cb["host_cores"] and d["host"]["cores_used_by_cpu_baseline"] == cb["cores"]
    if cb["cores"] > 1:
        assert cb["value"] > cb["single_thread"]
    assert d["config"]["obs_dim"] == 13 and d["config"]["envs_per_gpu"] == 512 and d["config"]["node_feats"] == 7
    assert d["metric"].startswith("env-steps/sec (whole node), nav_graph_metered_single_corridor_rot_inv")


def test_smoke_entry():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]


def test_two_rank_rehearsal_of_the_multi_gpu_bench_path():
    """The driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` on an 8-GPU node; this box has one GPU, so the same command runs
    as a REHEARSAL — two ranks on the one GPU, gloo instead of RCCL (GMPE_BENCH_REHEARSAL=1) — to pin the multi-rank branch end to end: process-group init before any GPU
    work, env-range sharding by rank (env_id_base), barrier + max-over-ranks timing, rank 0 printing ONE line, the rollout gather. No scaling claim follows from it."""
    env = dict(os.environ, GMPE_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--reps", "2", "--workload", "c3", "--envs", "512",
           "--gather", "--no-cpu-baseline", "--no-boundary"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                        # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 512
    assert d["value"] > 0 and abs(d["value"] - 2 * 512 * 10 / (d["ms_per_step"] * 1e-3 * 10)) < 1e-6 * d["value"]     # whole-job aggregate over both ranks
    assert d["with_gather"]["value"] > 0 and d["with_gather"]["slab_bytes_per_rank"] > 0
    ro = d["with_gather"]["rollout"]                          # round 4: one gather per 25-step rollout of the compact slab (entity table instead of node rows)
    assert ro["value"] > 0 and ro["steps_per_rollout"] == 25 and ro["bytes_per_env_step"] < 0.2 * ro["bytes_per_env_step_rows_form"]
    assert abs(ro["slab_bytes_per_rank"] / (25 * 512) - ro["bytes_per_env_step"]) < 64
    assert "cpu_baseline" not in d and "one_slot" not in d     # N = 1 only


def test_one_rank_rccl_run_of_the_multi_gpu_bench_branch():
    """The rehearsal above swaps RCCL for gloo (two ranks cannot share a device under RCCL). This one keeps backend "nccl" = RCCL and runs the same multi-rank branch
    with ONE rank under torch.distributed.run (GMPE_BENCH_NCCL_SOLO=1): process group bound to the device before any GPU work, barriers, the on-device MAX reduction
    of the timing, both gathers over RCCL and the learner-side expansion — what the driver's N > 1 commands execute, minus the peers."""
    env = dict(os.environ, GMPE_BENCH_NCCL_SOLO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--reps", "2", "--gather", "--no-cpu-baseline", "--no-boundary"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["envs_per_gpu"] == 4096
    assert d["verified"]["ints_exact"] is True and d["verified"]["max_abs_err"] < 1e-5
    assert d["with_gather"]["value"] > 0 and d["with_gather"]["rollout"]["value"] > 0            # both gathers ran over RCCL
    assert d["roofline"]["frac"] > 0.3
