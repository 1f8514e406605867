"""Seeded random sweep over the configuration space — scenario, agents / landmarks / obstacles / walls, batch size, feature type,
contact family, shared reward — crossed with the performance knobs that select a kernel instantiation (tile shape, exact-size vs
run-time sizes, nontemporal stores, split big-E path, wave specialisation): every draw must give the oracle's results, step by step,
through single launches and then through one rollout launch. The fixed parametrised cases (test_gpu_parity.py,
test_gpu_instantiations.py) cover the shapes the bench runs; this covers the combinations nobody thought of."""
import os

import numpy as np
import pytest

import gmpe
import oracle_lib as ol
from test_gpu_parity import TOL, _compare_state, _compare_step, _engine, _np

pytestmark = pytest.mark.gpu
JULY = "nav_metered_one_goal_graph_rotate_tube_july"
SCENS = [JULY, "navigation_graph", "nav_graph_metered_single_corridor_rot_inv", "two_phase_graph", "three_phase_graph"]


def _draw(k):
    r = np.random.RandomState(7000 + k)
    scen = SCENS[k % len(SCENS)]
    A = int(r.choice([1, 2, 3, 3, 4, 5, 6, 7, 8, 10, 10, 12, 16, 21]))
    ws = r.choice([3.0, 4.0, 6.0] if A <= 6 else ([4.0, 6.0, 8.0] if A <= 12 else [8.0, 12.0]))   # room for the placement sampler (…_july.py:895-904)
    kw = dict(scenario_name=scen, num_envs=int(r.randint(1, 70)), num_agents=A, world_size=float(ws),
              episode_length=int(r.randint(4, 11)), seed=int(r.randint(1, 10 ** 6)), collaborative=bool(r.rand() < 0.25))
    if scen == "navigation_graph":
        kw["num_landmarks"] = A + int(r.choice([0, 0, 1, 3]))
        kw["num_obstacles"] = int(r.choice([0, 0, 1, 2, 4]))
        kw["num_walls"] = int(r.choice([0, 0, 4]))
        kw["total_actions"] = int(r.choice([5, 5, 9]))
        if r.rand() < 0.35:
            kw.update(contact_family="classic", agent_size=float(r.choice([0.05, 0.15])), collider_size=float(r.choice([0.1, 0.2])),
                      agent_mass=float(r.choice([1.0, 2.0])), agent_accel=(None if r.rand() < 0.5 else float(r.choice([3.0, 5.0]))))
    if r.rand() < 0.3:
        kw["graph_feat_type"] = "global"
    knobs = {}
    if r.rand() < 0.6:
        knobs["G"] = int(r.randint(1, max(2, 64 // A + 1)))
        knobs["BLOCK"] = int(r.choice([64, 128, 256]))
    if r.rand() < 0.3:
        knobs["AP"] = 0
    if r.rand() < 0.4:
        knobs["NT"] = int(r.rand() < 0.7)
    if r.rand() < 0.3:
        knobs["SPLIT"] = 1
        knobs["CHUNKS"] = int(r.choice([1, 2, 3, 5]))
        knobs["AHEAD"] = int(r.choice([0, 1, 2]))
        knobs["XSTEP"] = int(r.choice([0, 1, 1]))
    if r.rand() < 0.2:
        knobs["SPEC"] = 0
    if r.rand() < 0.2:
        knobs["ROLLNT"] = 1
    if scen != "navigation_graph" and r.rand() < 0.4:                     # drawn last: the earlier draws of every k stay what they were in round 3
        kw["formation_type"] = str(r.choice(["line", "circle"]))            # distinct landmarks (…_july.py:492-495)
    return kw, knobs


@pytest.mark.parametrize("k", range(int(os.environ.get("GMPE_SWEEP_DRAWS", "64"))))     # GMPE_SWEEP_DRAWS=1000: the occasional long soak
def test_random_config_and_knobs_vs_oracle(monkeypatch, k):
    import torch
    kw, knobs = _draw(k)
    for name, val in knobs.items():
        monkeypatch.setenv("GMPE_" + name, str(val))
    cfg = gmpe.make_config(**kw)
    N, A = cfg.num_envs, cfg.num_agents
    eng, orc = _engine(cfg), ol.Oracle(cfg)
    rng = np.random.RandomState(k)
    eo, oo = eng.reset(), orc.reset()
    np.testing.assert_allclose(_np(eo.obs), oo[0], rtol=0, atol=TOL, err_msg=str((kw, knobs)))
    _compare_state(eng, orc, "reset %r %r" % (kw, knobs))
    T = 2 * cfg.episode_length + 3
    for t in range(T):                                                       # one launch per step (auto-resets included)
        act = rng.randint(0, cfg.n_actions, (N, A)).astype(np.int32)
        eo, oo = eng.step(torch.as_tensor(act)), orc.step(act)
        _compare_step(eo, oo, cfg.num_entities, A, "t=%d %r %r" % (t, kw, knobs))
    _compare_state(eng, orc, "after the step loop %r %r" % (kw, knobs))
    K = cfg.episode_length + 2                                               # then K steps through gmpe_step_many (rollout kernel where eligible)
    acts = rng.randint(0, cfg.n_actions, (K, N, A)).astype(np.int32)
    eo = eng.step_many(torch.as_tensor(acts, device="cuda"), K)
    for j in range(K):
        oo = orc.step(acts[j])
    _compare_step(eo, oo, cfg.num_entities, A, "step_many %r %r" % (kw, knobs))
    _compare_state(eng, orc, "after step_many %r %r" % (kw, knobs))
    eng.check_errors()
    eng.close()
