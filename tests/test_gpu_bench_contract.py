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
    d = _bench("--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):             # cpu_baseline: next test (skipped here for speed)
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "env-steps/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["key"] == "c2" and d["config"]["envs_per_gpu"] == 4096 and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["launches"] == 20
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / average launch duration (HIP events on the launch stream)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_env_step"] * r["env_steps_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(d["value"] - 4096 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]
    assert d["value"] > 1e6                                  # BASELINE.json target on one MI355X


def test_cpu_baseline_leg_and_other_workload():
    d = _bench("--workload", "c3r", "--envs", "512")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "env-steps/s" and cb["value"] > 0 and "sample" in cb
    assert d["config"]["obs_dim"] == 13 and d["config"]["envs_per_gpu"] == 512


def test_smoke_entry():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
