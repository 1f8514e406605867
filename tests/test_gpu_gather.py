"""Compact node form + rollout-granularity gather (SURVEY §8e "state + one E×E per env").

* the engine's entity table (gmpe_outputs.entity_table) against the oracle's, and `gmpe_expand_node_obs` against the engine's OWN node_obs rows — bit for bit, for every
  scenario x feature type, through steps in which agents reach goals (the ordered-visibility rule for re-drawn velocities) and through resets;
* `DeviceRolloutBuffer` with node_form="table": one launch per rollout, rows expanded on demand == the rows-form buffer;
* `ShardedRolloutCollector` over RCCL with the real engine (world_size 1: slab binding, the collective, unpack + in-place expansion): the learner's arrays are the
  unsharded buffer's bit for bit over three alternating rollouts. The world_size-2 twin runs on CPU (gloo, tests/test_host_logic.py).

What it replaces: GraphSubprocVecEnv.step_wait's remote.recv() loop (onpolicy/envs/env_wrappers.py:996-1004) feeding GraphReplayBuffer.insert (onpolicy/utils/graph_buffer.py:168-251)."""
import os

import numpy as np
import pytest

import gmpe
import oracle_lib as ol
from test_gpu_parity import ROTFAM, _np

pytestmark = pytest.mark.gpu
JULY = "nav_metered_one_goal_graph_rotate_tube_july"
SCENS = [JULY, "navigation_graph"] + ROTFAM


def _queue(orc, targets, rng, N, A):
    """agents queued in front of the tube entrance, heading along it (goal reaches + heading re-draws follow within ~30 steps)"""
    tube = orc.get("tube"); e_ = tube[:, 5:7]; ent = tube[:, 1:3]
    k = np.arange(A)[None, :, None]
    pos = ent[:, None, :] - e_[:, None, :] * (0.1 + 0.2 * k) + rng.uniform(-0.1, 0.1, (N, A, 2))
    th = np.arctan2(e_[:, 1], e_[:, 0])[:, None] + rng.uniform(-0.2, 0.2, (N, A))
    for tgt in targets:
        tgt.set("x", pos[..., 0]); tgt.set("y", pos[..., 1]); tgt.set("s2", th); tgt.set("s3", np.full((N, A), 0.08))


@pytest.mark.parametrize("feat", ["relative", "global"])
@pytest.mark.parametrize("scen", SCENS)
def test_expand_node_obs_is_bit_identical_to_the_engines_rows(scen, feat):
    import torch
    from gmpe.engine import GmpeEngine, expand_adj, expand_node_obs
    nav = scen == "navigation_graph"
    N, A = 37, 4
    kw = dict(scenario_name=scen, num_envs=N, num_agents=A, world_size=2.4, episode_length=(8 if nav else 30), seed=91, graph_feat_type=feat)
    if nav:
        kw.update(num_obstacles=2, num_landmarks=5)
    cfg = gmpe.make_config(**kw)
    eng, orc = GmpeEngine(cfg, node_form="both"), ol.Oracle(cfg)
    W = cfg.entity_table_width
    assert W == eng.lib.gmpe_entity_table_width(cfg) and eng.out.entity_table.shape == (N, W)
    o = eng.reset(); orc.reset()
    tab_o = np.zeros((N, W)); orc.lib.gmpo_get_entity_table(orc.h, tab_o.ctypes.data)
    np.testing.assert_allclose(_np(o.entity_table), tab_o, rtol=0, atol=1e-9)
    assert torch.equal(expand_node_obs(cfg, o.entity_table), o.node_obs)
    assert torch.equal(expand_adj(cfg, o.entity_table, copies=A), o.adj)
    rng = np.random.RandomState(4)
    if not nav:
        _queue(orc, (eng, orc), rng, N, A)
    redraws, masked_rows = 0, 0
    for t in range(60):
        act = (rng.randint(0, cfg.n_actions, (N, A)) if nav else np.where(rng.rand(N, A) < 0.7, 14, rng.randint(0, 25, (N, A)))).astype(np.int32)
        st_before = eng.get("status").copy()
        o = eng.step(torch.as_tensor(act)); oo = orc.step(act)
        orc.lib.gmpo_get_entity_table(orc.h, tab_o.ctypes.data)
        np.testing.assert_allclose(_np(o.entity_table), tab_o, rtol=0, atol=1e-9, err_msg="entity table t=%d" % t)
        rows = expand_node_obs(cfg, o.entity_table)
        assert torch.equal(rows, o.node_obs), "t=%d: expanded rows differ from the engine's" % t
        np.testing.assert_allclose(_np(rows), oo[2], rtol=0, atol=1e-5)
        # the adjacency from the same table (positions + this step's mask words): the engine's own matrix bit for bit, masked rows included
        adj1 = expand_adj(cfg, o.entity_table)
        assert torch.equal(adj1, o.adj[:, 0]) and torch.equal(expand_adj(cfg, o.entity_table, copies=A), o.adj), "t=%d: expanded adjacency differs" % t
        masked_rows += int((_np(adj1).sum(axis=2) == 0).sum())
        redraws += int((eng.get("status").astype(bool) & ~st_before.astype(bool) & ~oo[7][:, None]).sum())
    if not nav:
        assert redraws >= 10                                        # newly-finished agents: ego i saw agent k's re-drawn velocity iff k <= i
        assert masked_rows >= 10                                    # ... and their rows / columns of the adjacency were zeroed
    # offset form: the rows of this rank's envs written into a larger global array
    big = torch.zeros((N + 9, A, cfg.num_entities, cfg.node_feats), device="cuda")
    expand_node_obs(cfg, o.entity_table, out=big, out_envs=N + 9, env_offset=5)
    assert torch.equal(big[5:5 + N], o.node_obs) and (big[:5] == 0).all() and (big[5 + N:] == 0).all()
    with pytest.raises(Exception):
        expand_node_obs(cfg, o.entity_table, out=big, out_envs=N + 9, env_offset=10)   # does not fit
    eng.check_errors()


@pytest.mark.parametrize("scen", [JULY, "navigation_graph", ROTFAM[1]])
def test_rollout_buffer_table_form_equals_rows_form(scen):
    """DeviceRolloutBuffer(node_form='table'): the rollout kernel writes entity-table slots instead of node rows (8x fewer bytes); .node_obs expands them."""
    import torch
    from gmpe.engine import GmpeEngine
    from gmpe.rollout import DeviceRolloutBuffer
    N, A, T = 45, 10, 9
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, world_size=4.0, episode_length=6, seed=12)
    e1, e2 = GmpeEngine(cfg, adj_compact=True), GmpeEngine(cfg, adj_compact=True, node_form="table")
    assert e2.out.node_obs is None
    b1, b2 = DeviceRolloutBuffer(e1, T), DeviceRolloutBuffer(e2, T)
    b1.warmup(); b2.warmup()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for rep in range(3):
        acts = torch.randint(0, cfg.n_actions, (T, N, A), generator=g, device="cuda", dtype=torch.int32)
        b1.collect(acts); b2.collect(acts)
        torch.cuda.synchronize()
        assert torch.equal(b2.node_obs, b1.node_obs), rep
        for name in ("obs", "_adj", "rewards", "dones", "masks", "active_masks", "agent_id"):
            assert torch.equal(getattr(b1, name), getattr(b2, name)), (rep, name)
        b1.after_update(); b2.after_update()
    # one launch per step writes the table too
    b1.insert_step(acts[0]); b2.insert_step(acts[0])
    assert torch.equal(b2.node_obs[:2], b1.node_obs[:2])


def test_expand_adj_odd_entity_count_and_64_agents():
    """E*E not a multiple of 4 (scalar path) and E = 128 (four mask words): expand_adj == the engine's adjacency."""
    import torch
    from gmpe.engine import GmpeEngine, expand_adj
    for kw in (dict(scenario_name="navigation_graph", num_envs=20, num_agents=3, num_obstacles=1, world_size=3.0, episode_length=7, seed=26),
               dict(scenario_name=JULY, num_envs=3, num_agents=64, world_size=30.0, episode_length=4, seed=23)):
        cfg = gmpe.make_config(**kw)
        eng = GmpeEngine(cfg, node_form="both", adj_compact=True)
        o = eng.reset()
        assert torch.equal(expand_adj(cfg, o.entity_table), o.adj)
        g = torch.Generator(); g.manual_seed(1)
        for t in range(9):
            o = eng.step(torch.randint(0, cfg.n_actions, (cfg.num_envs, cfg.num_agents), generator=g, dtype=torch.int32))
            assert torch.equal(expand_adj(cfg, o.entity_table), o.adj), (kw["num_agents"], t)
        if cfg.num_agents == 64:                                     # force masks into the upper words: agents 40.. done
            st = eng.get("status"); st[:, 40:] = 1; eng.set("status", st)
            gt = eng.get("goal_tracker"); gt[:, 40:] = np.arange(40, 64); eng.set("goal_tracker", gt)
            o = eng.step(torch.zeros((cfg.num_envs, 64), dtype=torch.int32))
            a = expand_adj(cfg, o.entity_table)
            assert torch.equal(a, o.adj) and (a[:, 45] == 0).all() and (a[:, 64 + 45] == 0).all() and (a[:, 3, :40] != 0).any()
        eng.check_errors()


@pytest.mark.parametrize("adj_form", [None, "none"])
@pytest.mark.parametrize("scen", [JULY, ROTFAM[0], "navigation_graph"])
def test_sharded_rollout_collector_real_engine_over_rccl(scen, adj_form):
    import torch
    import torch.distributed as dist
    from gmpe.engine import GmpeEngine
    from gmpe.rollout import DeviceRolloutBuffer
    from gmpe.sharding import ShardedRolloutCollector, rollout_bytes_per_env_step
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500), rank=0, world_size=1, device_id=torch.device("cuda", 0))
    N, A, T = 52, 10, 7
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, world_size=4.0, episode_length=5, seed=44)
    e_ref = GmpeEngine(cfg, adj_compact=True)
    ref = DeviceRolloutBuffer(e_ref, T)
    col = ShardedRolloutCollector(GmpeEngine(cfg, adj_compact=True, node_form="table", adj_form=adj_form), T, 1)
    ref.warmup(); col.warmup()
    assert col.slab_bytes < (0.45 if adj_form is None else 0.25) * (T + 1) * N * rollout_bytes_per_env_step(cfg, T, "rows")   # compact: < half of the rows-form bytes; table only: < a quarter
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for rep in range(3):                                                     # the two slabs alternate; slot 0 is carried from one to the other
        acts = torch.randint(0, cfg.n_actions, (T, N, A), generator=g, device="cuda", dtype=torch.int32)
        if rep:
            ref.after_update()
        ref.collect(acts)
        b = col.collect_and_gather_async(acts)
        u = col.unpack(b)
        torch.cuda.synchronize()
        for name, key in (("obs", "obs"), ("node_obs", "node_obs"), ("adj", "_adj"), ("rewards", "rewards"), ("masks", "masks"), ("active_masks", "active_masks")):
            assert torch.equal(u[name], getattr(ref, key)), (rep, name)
        assert torch.equal(u["dones"], ref.dones.bool()) and torch.equal(u["agent_id"], ref.agent_id)
    with pytest.raises(ValueError):
        ShardedRolloutCollector(GmpeEngine(cfg, adj_compact=True), T, 1)     # needs the table form
