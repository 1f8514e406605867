"""A consumer of the C ABI that is neither Python nor torch (examples/c_abi_consumer.c, plain C + the HIP runtime): the boundary of SURVEY.md §8(b) is
`include/gmpe.h` and nothing else. The program is compiled with gcc against the header and libgmpe.so, fed a gmpe_config as raw bytes, and its checksums of
reset / closed-loop / rollout-launch outputs must equal the Python engine's on the same config and actions."""
import os
import re
import subprocess

import numpy as np
import pytest

import gmpe
from c_consumer_build import ROOT, build_c_consumer

pytestmark = pytest.mark.gpu
JULY = "nav_metered_one_goal_graph_rotate_tube_july"


@pytest.mark.parametrize("scen,A", [("navigation_graph", 10), (JULY, 10), ("nav_graph_metered_single_corridor_rot_inv", 4)])
def test_plain_c_consumer_matches_the_python_engine(tmp_path, scen, A):
    import torch
    from gmpe.engine import GmpeEngine
    exe = build_c_consumer(str(tmp_path))
    N, K = 37, 6
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, world_size=4.0, episode_length=4, seed=5)
    path = str(tmp_path / "cfg.bin")
    with open(path, "wb") as fh:
        fh.write(bytes(cfg))                                               # the config is a POD: this is all a foreign host language has to produce
    out = subprocess.run([exe, path, str(K)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = {m.group(1): [float(x) for x in re.findall(r"[-+]?\d\.\d+e[-+]\d+", m.group(2))] for m in re.finditer(r"^(reset|loop|rollout) (.*)$", out.stdout, re.M)}
    assert set(got) == {"reset", "loop", "rollout"}, out.stdout

    eng = GmpeEngine(cfg, adj_compact=True)
    n, a, k = np.meshgrid(np.arange(N), np.arange(A), np.arange(K), indexing="ij")
    acts = torch.as_tensor(((7 * n + 3 * a + k) % cfg.n_actions).transpose(2, 0, 1).astype(np.int32).copy(), device="cuda")
    sums = lambda o: [float(getattr(o, key).double().sum()) for key in ("obs", "node_obs", "adj")]
    o = eng.reset()
    np.testing.assert_allclose(got["reset"], sums(o), rtol=1e-9, atol=1e-9)
    rsum = 0.0
    for q in range(K):
        o = eng.step(acts[q]); rsum += float(o.reward.double().sum())
    np.testing.assert_allclose(got["loop"], sums(o) + [rsum], rtol=1e-9, atol=1e-9)
    eng.reset()                                                            # second episode: the env streams continue from their counters, as in the C program
    o = eng.rollout(acts, K)
    np.testing.assert_allclose(got["rollout"], sums(o), rtol=1e-9, atol=1e-9)
    eng.check_errors()
