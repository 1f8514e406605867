"""GPU parity of every kernel instantiation the bench runs.

`gmpe_create` picks the kernel instantiation from the launch size: big launches (c4 / c5) take nontemporal stores
(`FL = 0` + `global_store_dwordx4 ... nt`) and, from round 2 on, the split path (fused kernel writes the compact matrix,
`k_adj_expand` materialises the A ego copies). Small-N tests never reach those by themselves, so they are forced here
(GMPE_NT / GMPE_SPLIT / GMPE_SPEC / GMPE_G / GMPE_BLOCK are performance knobs: results must not change), and the
full-size c3 / c4 / c5-shard launches are compared with the oracle on an env slice (counter-based RNG keyed by the
global env id makes envs [0, n) of a big batch equal to an n-env engine with the same seed).

Reference semantics under test: multiagent/environment.py:1021-1063 (step), onpolicy/envs/env_wrappers.py:865-873 (auto-reset).
"""
import numpy as np
import pytest

import gmpe
import oracle_lib as ol
from test_gpu_parity import ROT, TOL, _compare_state, _compare_step, _engine, _np, _rollout_vs_oracle

pytestmark = pytest.mark.gpu

JULY = "nav_metered_one_goal_graph_rotate_tube_july"


def _knobs(monkeypatch, **kw):
    for k, v in kw.items():
        monkeypatch.setenv("GMPE_" + k.upper(), str(v))


NT_CASES = [
    # (scenario kwargs, steps, G, BLOCK)
    (dict(scenario_name="navigation_graph", num_envs=40, num_agents=6, num_obstacles=3, num_walls=4, world_size=3.0,
          episode_length=9, seed=51), 24, None, None),
    (dict(scenario_name="navigation_graph", num_envs=33, num_agents=10, world_size=4.0, episode_length=8, seed=52), 20, 4, 256),
    (dict(scenario_name=JULY, num_envs=70, num_agents=10, world_size=4.0, episode_length=10, seed=53), 24, 4, 256),
    (dict(scenario_name=JULY, num_envs=21, num_agents=10, world_size=4.0, episode_length=10, seed=54), 14, 2, 64),
    (dict(scenario_name=ROT, num_envs=48, num_agents=10, world_size=4.0, episode_length=10, seed=55), 24, 6, 256),
    (dict(scenario_name="three_phase_graph", num_envs=30, num_agents=4, world_size=4.0, episode_length=10, seed=56), 22, 3, 128),
    (dict(scenario_name="navigation_graph", num_envs=6, num_agents=32, num_obstacles=8, num_walls=4, world_size=8.0,
          episode_length=5, seed=57), 12, None, None),
]


@pytest.mark.parametrize("kw,steps,G,B", NT_CASES, ids=["%s-A%d-G%s-B%s" % (c[0]["scenario_name"][:12], c[0]["num_agents"], c[2], c[3]) for c in NT_CASES])
def test_nontemporal_store_instantiation_vs_oracle(monkeypatch, kw, steps, G, B):
    """FL = 0 + inline-asm nt stores (what c4 / c5 run) give the oracle's results."""
    _knobs(monkeypatch, nt=1, split=0)
    if G:
        _knobs(monkeypatch, g=G, block=B)
    cfg = gmpe.make_config(**kw)
    eng = _engine(cfg)
    assert eng.tuning()["nt"] == 1 and eng.tuning()["split"] == 0
    eng.close()
    assert _rollout_vs_oracle(cfg, steps, seed=61) >= 1


@pytest.mark.parametrize("kw,steps,G,B", NT_CASES[:1] + NT_CASES[2:3] + NT_CASES[4:5] + NT_CASES[6:], ids=["nav-walls", "july-A10", "rot-A10", "c4-shape"])
def test_split_adjacency_path_vs_oracle(monkeypatch, kw, steps, G, B):
    """Big-E path forced at small N: fused kernel -> compact matrix in the handle's scratch -> k_adj_expand (nt) to [N,A,E,E]."""
    _knobs(monkeypatch, split=1)
    if G:
        _knobs(monkeypatch, g=G, block=B)
    cfg = gmpe.make_config(**kw)
    eng = _engine(cfg)
    assert eng.tuning()["split"] == 1
    eng.close()
    assert _rollout_vs_oracle(cfg, steps, seed=62) >= 1


@pytest.mark.parametrize("scen", ["navigation_graph", JULY, ROT])
def test_unspecialised_multiwave_tiles_vs_oracle(monkeypatch, scen):
    """GMPE_SPEC=0 at BLOCK = 256: all four waves take the block-synchronous order (no wave specialisation)."""
    _knobs(monkeypatch, spec=0, g=4, block=256)
    cfg = gmpe.make_config(scenario_name=scen, num_envs=45, num_agents=10, world_size=4.0, episode_length=9, seed=63)
    assert _rollout_vs_oracle(cfg, 22, seed=64) >= 45


def test_navigation_graph_configs0_three_agents_one_env():
    """BASELINE.json configs[0]: navigation_graph, 3 agents / 3 landmarks / 0 obstacles, ONE env (k_env<*, 3, SC_NAV, *>)."""
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=1, num_agents=3, world_size=2.0, episode_length=25, seed=3)
    assert cfg.num_entities == 6
    assert _rollout_vs_oracle(cfg, 60, seed=65) >= 2
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=37, num_agents=3, world_size=2.0, episode_length=11, seed=4)
    assert _rollout_vs_oracle(cfg, 30, seed=66) >= 37


def _full_size_slice(cfg_kw, n_total, n_slice, steps, act_seed, base=0, many=0):
    """Full-size launch vs an oracle run of n_slice of its envs (the first ones, or those from `base` on), every step; returns the
    last engine outputs."""
    import torch
    cfg = gmpe.make_config(num_envs=n_total, **cfg_kw)
    small = gmpe.make_config(num_envs=n_slice, env_id_base=base, **cfg_kw)
    eng, orc = _engine(cfg), ol.Oracle(small)
    eo, oo = eng.reset(), orc.reset()
    np.testing.assert_allclose(_np(eo.obs[base:base + n_slice]), oo[0], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(eo.node_obs[base:base + n_slice]), oo[2], rtol=0, atol=TOL)
    g = torch.Generator(device="cpu"); g.manual_seed(act_seed)
    A, E = cfg.num_agents, cfg.num_entities
    n_resets = 0
    for t in range(steps):
        act = torch.randint(0, cfg.n_actions, (n_total, A), generator=g, dtype=torch.int32)
        o = eng.step(act)
        oo = orc.step(act[base:base + n_slice].numpy())
        n_resets += int(oo[7].sum())
        lab = "t=%d" % t
        np.testing.assert_allclose(_np(o.obs[base:base + n_slice]), oo[0], rtol=0, atol=TOL, err_msg=lab + " obs")
        np.testing.assert_allclose(_np(o.node_obs[base:base + n_slice]), oo[2], rtol=0, atol=TOL, err_msg=lab + " node")
        adj = _np(o.adj[base:base + n_slice])
        np.testing.assert_allclose(adj, np.broadcast_to(oo[3][:, None], adj.shape), rtol=0, atol=TOL, err_msg=lab + " adj")
        np.testing.assert_array_equal(adj == 0, np.broadcast_to(oo[3][:, None] == 0, adj.shape), err_msg=lab + " adj mask")
        np.testing.assert_allclose(_np(o.reward[base:base + n_slice]), oo[4], rtol=0, atol=TOL, err_msg=lab + " rew")
        np.testing.assert_array_equal(_np(o.done[base:base + n_slice]).astype(bool), oo[5], err_msg=lab + " done")
        np.testing.assert_allclose(_np(o.info[base:base + n_slice]), oo[6], rtol=2e-6, atol=2e-5, err_msg=lab + " info")
    for f in ("x", "y", "s2", "s3"):
        np.testing.assert_allclose(eng.get(f)[base:base + n_slice], orc.get(f), rtol=0, atol=1e-9, err_msg=f)
    for f in ("status", "rng_ctr", "current_step", "goal_tracker", "n_agent_coll", "n_obst_coll"):
        np.testing.assert_array_equal(eng.get(f)[base:base + n_slice], orc.get(f), err_msg=f)
    if many:                                                             # then `many` open-loop steps in ONE gmpe_step_many call, final outputs + state
        acts = torch.randint(0, cfg.n_actions, (many, n_total, A), generator=g, dtype=torch.int32)
        o = eng.step_many(acts.to("cuda"), many)
        for k in range(many):
            oo = orc.step(acts[k, base:base + n_slice].numpy())
        np.testing.assert_allclose(_np(o.obs[base:base + n_slice]), oo[0], rtol=0, atol=TOL, err_msg="step_many obs")
        np.testing.assert_allclose(_np(o.node_obs[base:base + n_slice]), oo[2], rtol=0, atol=TOL, err_msg="step_many node")
        adj = _np(o.adj[base:base + n_slice])
        np.testing.assert_allclose(adj, np.broadcast_to(oo[3][:, None], adj.shape), rtol=0, atol=TOL, err_msg="step_many adj")
        np.testing.assert_allclose(_np(o.reward[base:base + n_slice]), oo[4], rtol=0, atol=TOL, err_msg="step_many rew")
        np.testing.assert_array_equal(_np(o.done[base:base + n_slice]).astype(bool), oo[5], err_msg="step_many done")
        for f in ("x", "y", "s2", "s3"):
            np.testing.assert_allclose(eng.get(f)[base:base + n_slice], orc.get(f), rtol=0, atol=1e-9, err_msg="step_many " + f)
        for f in ("status", "rng_ctr", "current_step"):
            np.testing.assert_array_equal(eng.get(f)[base:base + n_slice], orc.get(f), err_msg="step_many " + f)
    eng.check_errors()
    return eng, o, n_resets


def _graph_invariants(o, A, E, F):
    import torch
    adj = o.adj
    assert torch.equal(adj, adj.transpose(-1, -2)) and (torch.diagonal(adj, dim1=-2, dim2=-1) == 0).all()
    assert torch.equal(adj, adj[:, :1].expand_as(adj))                  # one matrix per env
    assert torch.isfinite(o.obs).all() and torch.isfinite(o.node_obs).all() and torch.isfinite(o.reward).all()
    typ = o.node_obs[..., F - 1]
    assert (typ[:, :, :A] == 0).all() and (typ[:, :, A:] >= 1).all()
    idx = torch.arange(A, device=adj.device)
    assert (o.node_obs[:, idx, idx, 0:4] == 0).all()                     # ego row: zero relative velocity / position
    rel = o.node_obs[:, idx, :, 2:4].double()
    d = torch.sqrt((rel ** 2).sum(-1)).float()
    row = adj[:, idx, idx, :]
    live = row != 0
    assert torch.allclose(row[live], d[live], atol=2e-5)


def test_full_size_c3_with_oracle_slice():
    """configs[2] at full size (4096 x 10, the steady-state instantiation) + a 512-env oracle slice, every step."""
    eng, o, n_resets = _full_size_slice(dict(scenario_name=JULY, num_agents=10, world_size=4.0, episode_length=25, seed=1234), 4096, 512, 30, 42)
    assert n_resets >= 512
    t = eng.tuning()
    assert t["block"] == 256 and t["G"] == 4 and t["nt"] == 0
    _graph_invariants(o, 10, 20, 8)


def test_full_size_c4_with_oracle_slice():
    """configs[3] at full size: 8192 envs x (32 agents + 8 obstacles + 4 walls), E = 72 — the launch bench.py --workload c4 times."""
    eng, o, n_resets = _full_size_slice(dict(scenario_name="navigation_graph", num_agents=32, num_obstacles=8, num_walls=4, world_size=8.0,
                                             episode_length=6, seed=1234), 8192, 256, 10, 43)
    assert n_resets >= 256
    t = eng.tuning()
    assert t["nt"] == 1 or t["split"] == 1                               # the big-launch path really ran
    _graph_invariants(o, 32, 72, 8)


@pytest.mark.parametrize("n_envs,ahead", [(2048, 0), (4096, 2)], ids=["shard-2048", "4096-bounded-run-ahead"])
def test_full_size_c5_shard_with_oracle_slice(n_envs, ahead):
    """configs[4], one GPU's shard of 8: 2048 envs x 64 agents, E = 128 (8.6 GB of adjacency per step) — the split pipeline; and twice
    that, where the compact-matrix scratch outgrows the Infinity Cache and the fused kernel is held to two chunks ahead of the expansion."""
    eng, o, n_resets = _full_size_slice(dict(scenario_name="navigation_graph", num_agents=64, world_size=12.0, episode_length=5, seed=1234),
                                        n_envs, 128, 8, 44, base=n_envs - 128 if ahead else 0, many=7)   # the big case checks the LAST chunk's envs; then 7 chained steps
    assert n_resets >= 128
    t = eng.tuning()
    assert t["split"] == 1 and t["ahead"] == ahead and t["chunks"] == 8     # 256-env chunks unbounded, 512-env chunks with the bound
    assert t["xstep"] == 1 and t["chunks_x"] == n_envs // 512 and t["ahead_x"] == 2   # gmpe_step_many: the steps' pipelines chained
    _graph_invariants(o, 64, 128, 8)


@pytest.mark.parametrize("chunks,ahead,xstep", [(1, 0, 1), (3, 1, 1), (8, 0, 1), (8, 2, 1), (8, 2, 0), (None, None, 1)])
def test_split_pipeline_across_steps_equals_step_loop(monkeypatch, chunks, ahead, xstep):
    """gmpe_step_many on the split path (one chunk pipeline per step, any chunk count): same final outputs and state as one
    gmpe_step per step."""
    import torch
    _knobs(monkeypatch, split=1, xstep=xstep)
    if chunks is not None:
        _knobs(monkeypatch, chunks=chunks, ahead=ahead)
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=37, num_agents=12, num_obstacles=3, num_walls=4, world_size=5.0,
                           episode_length=6, seed=97)
    e1, e2 = _engine(cfg), _engine(cfg)
    assert e1.tuning()["split"] == 1
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    acts = torch.randint(0, cfg.n_actions, (5, 37, 12), generator=g, device="cuda", dtype=torch.int32)
    for K in (1, 2, 9, 14):
        for k in range(K):
            o1 = e1.step(acts[k % 5])
        o2 = e2.step_many(acts, K)
        torch.cuda.synchronize()
        for key in ("obs", "agent_id", "node_obs", "adj", "reward", "done", "info"):
            assert torch.equal(getattr(o1, key), getattr(o2, key)), (K, key)
        _compare_state(e1, e2, "K=%d" % K)
    e1.check_errors(); e2.check_errors()


def test_full_size_c4_rollout_kernel_with_oracle_slice():
    """configs[3] as bench.py --workload c4 times it since round 2: the K steps in ONE launch of the run-time-size rollout kernel with
    nontemporal stores (k_env<256, 0, SC_NAV_WALLS, 2>, three tiles per CU, 8192 tiles queued behind 768 slots)."""
    import torch
    kw = dict(scenario_name="navigation_graph", num_agents=32, num_obstacles=8, num_walls=4, world_size=8.0, episode_length=6, seed=1234)
    cfg = gmpe.make_config(num_envs=8192, **kw)
    eng, orc = _engine(cfg), ol.Oracle(gmpe.make_config(num_envs=192, **kw))
    t = eng.tuning()
    assert t["roll"] == 1 and t["split"] == 0 and t["nt"] == 1
    eng.reset(); orc.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(45)
    K = 9
    acts = torch.randint(0, cfg.n_actions, (K, 8192, 32), generator=g, device="cuda", dtype=torch.int32)
    o = eng.step_many(acts, K)
    a = acts[:, :192].cpu().numpy()
    for k in range(K):
        oo = orc.step(a[k])
    np.testing.assert_allclose(_np(o.obs[:192]), oo[0], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.node_obs[:192]), oo[2], rtol=0, atol=TOL)
    adj = _np(o.adj[:192])
    np.testing.assert_allclose(adj, np.broadcast_to(oo[3][:, None], adj.shape), rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.reward[:192]), oo[4], rtol=0, atol=TOL)
    np.testing.assert_array_equal(_np(o.done[:192]).astype(bool), oo[5])
    np.testing.assert_allclose(_np(o.info[:192]), oo[6], rtol=2e-6, atol=2e-5)
    for f in ("rng_ctr", "current_step", "status", "n_obst_coll"):
        np.testing.assert_array_equal(eng.get(f)[:192], orc.get(f), err_msg=f)
    _graph_invariants(o, 32, 72, 8)
    eng.check_errors()
