"""The persistent rollout kernel (gmpe_rollout_steps, k_env<*, 0, SC, 2>): K steps in ONE launch must be bit-identical to K
launches of the step kernel — outputs of every step (slot-per-step placement), masks, final state, RNG counters — for every
scenario variant, with auto-resets inside the rollout, mixed tiles, single- and multi-wave tiles.

What it replaces: the runner's collect loop with a fixed action source (onpolicy/runner/shared/graph_mpe_runner.py:57-103 →
envs.step → GraphReplayBuffer.insert, onpolicy/utils/graph_buffer.py:168-251).
"""
import numpy as np
import pytest

import gmpe
import oracle_lib as ol
from test_gpu_parity import ROTFAM, TOL, _compare_state, _engine, _np

pytestmark = pytest.mark.gpu
JULY = "nav_metered_one_goal_graph_rotate_tube_july"
OUT_KEYS = ("obs", "agent_id", "node_obs", "adj", "reward", "done", "info")


def _slots(eng, T):
    """Slot-per-step output storage [T, ...] for every output of the engine + masks."""
    import torch
    st = {k: torch.zeros((T,) + tuple(getattr(eng.out, k).shape), dtype=getattr(eng.out, k).dtype, device="cuda") for k in OUT_KEYS}
    st["masks"] = torch.full((T, eng.N, eng.A), -1.0, device="cuda")
    st["active"] = torch.full((T, eng.N, eng.A), -1.0, device="cuda")
    return st


def _rollout_into_slots(eng, acts, K, T, first=0):
    from gmpe.engine import StepOutputs
    st = _slots(eng, T)
    slot0 = StepOutputs(**{k: st[k][0] for k in OUT_KEYS})
    strides = {k: st[k][0].numel() for k in OUT_KEYS}
    strides["masks"] = eng.N * eng.A
    eng.rollout(acts, K, slot0=slot0, num_slots=T, first_slot=first, strides=strides, masks=st["masks"], active_masks=st["active"])
    return st


CASES = [
    dict(scenario_name="navigation_graph", num_envs=40, num_agents=10, world_size=4.0, episode_length=7, seed=71),
    dict(scenario_name="navigation_graph", num_envs=23, num_agents=6, num_obstacles=3, num_walls=4, world_size=3.0, episode_length=6, seed=72),
    dict(scenario_name=JULY, num_envs=50, num_agents=10, world_size=4.0, episode_length=8, seed=73),
    dict(scenario_name=ROTFAM[0], num_envs=31, num_agents=10, world_size=4.0, episode_length=8, seed=74),
    dict(scenario_name=ROTFAM[1], num_envs=31, num_agents=4, world_size=2.4, episode_length=9, seed=75),
    dict(scenario_name=ROTFAM[2], num_envs=18, num_agents=10, world_size=4.0, episode_length=8, seed=76),
    dict(scenario_name="navigation_graph", num_envs=5, num_agents=32, num_obstacles=8, num_walls=4, world_size=8.0, episode_length=4, seed=77),
    dict(scenario_name=JULY, num_envs=3, num_agents=64, world_size=30.0, episode_length=3, seed=78),
    dict(scenario_name=JULY, num_envs=29, num_agents=10, world_size=4.0, episode_length=7, seed=79, formation_type="line"),       # distinct landmarks: the cached
    dict(scenario_name=ROTFAM[0], num_envs=29, num_agents=10, world_size=4.0, episode_length=7, seed=80, formation_type="circle"),  # landmark block of rollouts
]


@pytest.mark.parametrize("shape", [None, (2, 64), (3, 256), (6, 256)], ids=["auto", "G2-B64", "G3-B256", "G6-B256"])
@pytest.mark.parametrize("kw", CASES, ids=["%s-A%d-O%d%s" % (c["scenario_name"][:10], c["num_agents"], c.get("num_obstacles", 0), "-" + c["formation_type"] if "formation_type" in c else "") for c in CASES])
def test_rollout_kernel_equals_step_loop(monkeypatch, kw, shape):
    import torch
    if shape:
        if shape[0] * kw["num_agents"] > 64:
            pytest.skip("G*A > 64")
        monkeypatch.setenv("GMPE_G", str(shape[0])); monkeypatch.setenv("GMPE_BLOCK", str(shape[1]))
    monkeypatch.setenv("GMPE_SPLIT", "0")
    cfg = gmpe.make_config(**kw)
    e1, e2 = _engine(cfg), _engine(cfg)
    e1.reset(); e2.reset()
    N, A = cfg.num_envs, cfg.num_agents
    K, S, T = 19, 5, 19
    g = torch.Generator(device="cuda"); g.manual_seed(kw["seed"])
    acts = torch.randint(0, cfg.n_actions, (S, N, A), generator=g, device="cuda", dtype=torch.int32)
    ref = {k: [] for k in OUT_KEYS}
    dones = []
    for k in range(K):
        o = e1.step(acts[k % S])
        for key in OUT_KEYS:
            ref[key].append(getattr(o, key).clone())
        dones.append(o.done.clone())
    st = _rollout_into_slots(e2, acts, K, T)
    torch.cuda.synchronize()
    n_all_done = 0
    for k in range(K):
        for key in OUT_KEYS:
            assert torch.equal(st[key][k], ref[key][k]), (key, k)
        d = dones[k].bool()
        alld = d.all(dim=1, keepdim=True)
        n_all_done += int(alld.sum())
        assert torch.equal(st["masks"][k], (~d).float()), ("masks", k)
        assert torch.equal(st["active"][k], (~(d & ~alld)).float()), ("active_masks", k)
    assert n_all_done >= N                                   # auto-resets happened inside the rollout
    _compare_state(e1, e2, "after rollout")
    # a second rollout continues from the carried-back state; single slot this time (every step overwrites the same buffers)
    for k in range(7):
        o1 = e1.step(acts[k % S])
    o2 = e2.rollout(acts, 7)
    for key in OUT_KEYS:
        assert torch.equal(getattr(o1, key), getattr(o2, key)), key
    _compare_state(e1, e2, "after second rollout")
    e1.check_errors(); e2.check_errors()


def test_rollout_kernel_vs_oracle_every_step():
    """The rollout kernel against the CPU oracle directly (not only against the step kernel)."""
    import torch
    cfg = gmpe.make_config(scenario_name=JULY, num_envs=64, num_agents=10, world_size=4.0, episode_length=9, seed=81)
    eng, orc = _engine(cfg), ol.Oracle(cfg)
    eng.reset(); orc.reset()
    K = 24
    rng = np.random.RandomState(5)
    acts = rng.randint(0, 25, (K, 64, 10)).astype(np.int32)
    st = _rollout_into_slots(eng, torch.as_tensor(acts, device="cuda"), K, K)
    for k in range(K):
        oo = orc.step(acts[k])
        np.testing.assert_allclose(_np(st["obs"][k]), oo[0], rtol=0, atol=TOL, err_msg="obs %d" % k)
        np.testing.assert_allclose(_np(st["node_obs"][k]), oo[2], rtol=0, atol=TOL, err_msg="node %d" % k)
        np.testing.assert_allclose(_np(st["adj"][k]), np.broadcast_to(oo[3][:, None], st["adj"][k].shape), rtol=0, atol=TOL, err_msg="adj %d" % k)
        np.testing.assert_allclose(_np(st["reward"][k]), oo[4], rtol=0, atol=TOL, err_msg="rew %d" % k)
        np.testing.assert_array_equal(_np(st["done"][k]).astype(bool), oo[5], err_msg="done %d" % k)
        np.testing.assert_allclose(_np(st["info"][k]), oo[6], rtol=2e-6, atol=2e-5, err_msg="info %d" % k)
    _compare_state(eng, orc, "end")


def test_rollout_slots_wrap_and_tape_mode():
    """first_slot > 0 with wrap-around (slot = (first + k) % num_slots), and the RNG tape (parity mode) inside a rollout."""
    import torch
    cfg = gmpe.make_config(scenario_name=ROTFAM[0], num_envs=20, num_agents=5, world_size=4.0, episode_length=5, seed=91)
    e1, e2 = _engine(cfg), _engine(cfg)
    tape = np.random.RandomState(2).rand(20, 4096)
    e1.set_tape(tape); e2.set_tape(tape)
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    acts = torch.randint(0, 25, (4, 20, 5), generator=g, device="cuda", dtype=torch.int32)
    T, K, first = 6, 10, 4
    ref = {}
    for k in range(K):
        o = e1.step(acts[k % 4])
        ref[(first + k) % T] = {key: getattr(o, key).clone() for key in OUT_KEYS}      # later steps overwrite wrapped slots
    st = _rollout_into_slots(e2, acts, K, T, first=first)
    for s in range(T):
        for key in OUT_KEYS:
            assert torch.equal(st[key][s], ref[s][key]), (key, s)
    _compare_state(e1, e2, "tape")
    e1.check_errors(); e2.check_errors()


def test_step_many_takes_the_rollout_kernel_and_falls_back(monkeypatch):
    """gmpe_step_many = one launch of the rollout kernel by default; GMPE_ROLL=0 keeps the launch loop. Same results."""
    import torch
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=70, num_agents=10, seed=8, episode_length=9)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    acts = torch.randint(0, 5, (7, 70, 10), generator=g, device="cuda", dtype=torch.int32)
    e1 = _engine(cfg); assert e1.tuning()["roll"] == 1
    monkeypatch.setenv("GMPE_ROLL", "0")
    e2 = _engine(cfg); assert e2.tuning()["roll"] == 0
    e1.reset(); e2.reset()
    o1, o2 = e1.step_many(acts, 23), e2.step_many(acts, 23)
    for k in OUT_KEYS:
        assert torch.equal(getattr(o1, k), getattr(o2, k)), k
    _compare_state(e1, e2, "roll vs loop")
    for K in (1, 2, 1, 3):                                   # degenerate rollouts: a single step is load -> step -> write-back in one pass of the loop
        o1, o2 = e1.step_many(acts, K), e2.step_many(acts, K)
        for k in OUT_KEYS:
            assert torch.equal(getattr(o1, k), getattr(o2, k)), (K, k)
        _compare_state(e1, e2, "K=%d" % K)


@pytest.mark.parametrize("compact", [False, True])
def test_device_rollout_buffer_collect_equals_insert_loop(compact):
    """DeviceRolloutBuffer.collect (one launch for the whole T-step rollout) == T insert_step calls."""
    import torch
    from gmpe.engine import GmpeEngine
    from gmpe.rollout import DeviceRolloutBuffer
    N, A, T = 36, 10, 12
    cfg = gmpe.make_config(num_envs=N, num_agents=A, episode_length=5, seed=19)
    e1, e2 = GmpeEngine(cfg, adj_compact=compact), GmpeEngine(cfg, adj_compact=compact)
    b1, b2 = DeviceRolloutBuffer(e1, T), DeviceRolloutBuffer(e2, T)
    b1.warmup(); b2.warmup()
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    acts = torch.randint(0, 25, (T, N, A), generator=g, device="cuda", dtype=torch.int32)
    for rep in range(2):
        for t in range(T):
            b1.insert_step(acts[t])
        b2.collect(acts)
        torch.cuda.synchronize()
        for name in ("obs", "node_obs", "_adj", "agent_id", "rewards", "dones", "masks", "active_masks", "info"):
            assert torch.equal(getattr(b1, name), getattr(b2, name)), (rep, name)
        assert b1.step == b2.step == 0
        assert torch.equal(e1.out.obs, e2.out.obs) and e1.out.obs.data_ptr() == b1.obs[T].data_ptr()
        b1.after_update(); b2.after_update()
    # partial collects: 5 + 7 steps
    for t in range(T):
        b1.insert_step(acts[t])
    b2.collect(acts[:5], 5); assert b2.step == 5
    b2.collect(acts[5:], 7); assert b2.step == 0
    for name in ("obs", "node_obs", "_adj", "rewards", "masks", "active_masks"):
        assert torch.equal(getattr(b1, name), getattr(b2, name)), name


@pytest.mark.parametrize("scen", ["navigation_graph", JULY] + ROTFAM)
def test_full_size_rollout_properties_and_slice(scen):
    """The launches bench.py times (c2 by default, c3 / c3r / c3p2 / c3p3 with --workload: 4096 x 10, K steps in ONE launch of the exact-size
    rollout kernel at its own tile shape): a 512-env oracle slice of the final step + the graph invariants."""
    import torch
    kw = dict(scenario_name=scen, num_agents=10, world_size=4.0, episode_length=25, seed=1234)
    cfg = gmpe.make_config(num_envs=4096, **kw)
    eng, orc = _engine(cfg), ol.Oracle(gmpe.make_config(num_envs=512, **kw))
    eng.reset(); orc.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(42)
    K = 60
    acts = torch.randint(0, cfg.n_actions, (K, 4096, 10), generator=g, device="cuda", dtype=torch.int32)
    o = eng.step_many(acts, K)
    a = acts[:, :512].cpu().numpy()
    for k in range(K):
        oo = orc.step(a[k])
    np.testing.assert_allclose(_np(o.obs[:512]), oo[0], rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.node_obs[:512]), oo[2], rtol=0, atol=TOL)
    assert eng.tuning()["ap"] == 10 and eng.tuning()["G_roll"] == (4 if scen == "navigation_graph" else 6)
    np.testing.assert_allclose(_np(o.adj[:512]), np.broadcast_to(oo[3][:, None], (512, 10, 20, 20)), rtol=0, atol=TOL)
    np.testing.assert_allclose(_np(o.reward[:512]), oo[4], rtol=0, atol=TOL)
    np.testing.assert_array_equal(_np(o.done[:512]).astype(bool), oo[5])
    np.testing.assert_array_equal(eng.get("rng_ctr")[:512], orc.get("rng_ctr"))
    np.testing.assert_array_equal(eng.get("current_step")[:512], orc.get("current_step"))
    adj = o.adj
    assert torch.equal(adj, adj.transpose(-1, -2)) and torch.equal(adj, adj[:, :1].expand_as(adj))
    eng.check_errors()


def test_rollout_nontemporal_store_variant_equals_step_loop(monkeypatch):
    """Rollouts that fill more slots than the Infinity Cache holds use nontemporal graph stores (GMPE_ROLLNT forces it at small size)."""
    import torch
    monkeypatch.setenv("GMPE_ROLLNT", "1")
    for scen, F in ((JULY, 8), (ROTFAM[0], 7), ("navigation_graph", 8)):
        cfg = gmpe.make_config(scenario_name=scen, num_envs=45, num_agents=10, world_size=4.0, episode_length=6, seed=55)
        e1, e2 = _engine(cfg), _engine(cfg)
        e1.reset(); e2.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(4)
        acts = torch.randint(0, cfg.n_actions, (3, 45, 10), generator=g, device="cuda", dtype=torch.int32)
        ref = []
        for k in range(13):
            o = e1.step(acts[k % 3])
            ref.append({key: getattr(o, key).clone() for key in OUT_KEYS})
        st = _rollout_into_slots(e2, acts, 13, 13)
        for k in range(13):
            for key in OUT_KEYS:
                assert torch.equal(st[key][k], ref[k][key]), (scen, key, k)
        _compare_state(e1, e2, scen)


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_fused_force_pass_with_goals_reached_mid_rollout(monkeypatch, fuse):
    """Rollouts of the exact-size navigation_graph tile compute the next step's contact forces inside the distance pass and keep the landmark x landmark block of the
    adjacency across steps (distance_force_pass; GMPE_FUSE=0: the classic passes). Agents parked on their goals reach them in the first steps, so the masks of reached
    landmarks must persist on the cached block, colliding agents exercise the forces, short episodes put resets inside the rollout: every step's outputs must equal
    the step loop's bit for bit, and the oracle's."""
    import torch
    monkeypatch.setenv("GMPE_FUSE", fuse)
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=37, num_agents=10, world_size=4.0, episode_length=9, seed=123)
    e1, e2, orc = _engine(cfg), _engine(cfg), ol.Oracle(cfg)
    e1.reset(); e2.reset(); orc.reset()
    # park agents 0..3 of every second env next to their own landmarks (inside goal_thresh) and put agents 4 / 5 on top of each other's edge (contact force)
    lm = e1.get("landmarks"); x = e1.get("x"); y = e1.get("y")
    for n in range(0, 37, 2):
        for a in range(4):
            x[n, a] = lm[n, a, 0] + 0.01 * (a + 1); y[n, a] = lm[n, a, 1] - 0.02
        x[n, 5] = x[n, 4] + 0.3; y[n, 5] = y[n, 4] + 0.1
    for e in (e1, e2, orc):
        e.set("x", x); e.set("y", y)
    K, T = 21, 21
    rng = np.random.RandomState(9)
    acts_np = rng.randint(0, cfg.n_actions, (K, 37, 10)).astype(np.int32)
    acts = torch.as_tensor(acts_np, device="cuda")
    ref = {k: [] for k in OUT_KEYS}
    for k in range(K):
        o = e1.step(acts[k])
        for key in OUT_KEYS:
            ref[key].append(getattr(o, key).clone())
    st = _rollout_into_slots(e2, acts, K, T)
    torch.cuda.synchronize()
    for k in range(K):
        for key in OUT_KEYS:
            assert torch.equal(st[key][k], ref[key][k]), (key, k)
        oo = orc.step(acts_np[k])
        np.testing.assert_allclose(_np(st["adj"][k]), np.broadcast_to(oo[3][:, None], st["adj"][k].shape), rtol=0, atol=TOL, err_msg="adj %d" % k)
        np.testing.assert_allclose(_np(st["obs"][k]), oo[0], rtol=0, atol=TOL, err_msg="obs %d" % k)
        np.testing.assert_array_equal(_np(st["done"][k]).astype(bool), oo[5], err_msg="done %d" % k)
    # masked landmark rows really occurred: some landmark x landmark block entries are zero off the diagonal in the first steps
    blk = _np(st["adj"][1])[0, 0, 10:, 10:]
    assert (blk[np.triu_indices(10, 1)] == 0).any()
    _compare_state(e1, e2, "after fused rollout")
    assert e2.tuning()["ap"] == 10
    e1.check_errors(); e2.check_errors()


def _bench_shape_rollout(kw, n_envs, n_slice, K, n_slots):
    """The launch bench.py times (bench.py run_k_steps, `rollout` mode with slots): eng.rollout(actions, K, slot0 = slot 0 of [n_slots, ...] storage for every
    output, strides = one slot) — NO masks, first_slot 0, the nontemporal path chosen by slot volume. Every surviving slot (the last n_slots steps) of the first
    n_slice envs is compared with the oracle, then the final state and the RNG counters."""
    import torch
    from gmpe.engine import StepOutputs
    cfg = gmpe.make_config(num_envs=n_envs, **kw)
    eng, orc = _engine(cfg), ol.Oracle(gmpe.make_config(num_envs=n_slice, **kw))
    eng.reset(); orc.reset()
    A, E = cfg.num_agents, cfg.num_entities
    g = torch.Generator(device="cuda"); g.manual_seed(42)
    acts = torch.randint(0, cfg.n_actions, (K, n_envs, A), generator=g, device="cuda", dtype=torch.int32)
    o = eng.out
    slots = {k: torch.empty((n_slots,) + tuple(getattr(o, k).shape), dtype=getattr(o, k).dtype, device="cuda") for k in OUT_KEYS}
    step_bytes = sum(getattr(o, k).numel() * getattr(o, k).element_size() for k in OUT_KEYS)
    assert step_bytes * n_slots > (256 << 20)                      # past the Infinity Cache: gmpe_rollout_steps takes the nontemporal store path (gmpe_step.hip, by slot volume)
    eng.rollout(acts, K, slot0=StepOutputs(**{k: v[0] for k, v in slots.items()}), num_slots=n_slots, strides={k: v[0].numel() for k, v in slots.items()})
    torch.cuda.synchronize()
    a = acts[:, :n_slice].cpu().numpy()
    n_resets = 0
    for k in range(K):
        oo = orc.step(a[k])
        n_resets += int(oo[7].sum())
        if k < K - n_slots:
            continue                                               # overwritten by step k + n_slots (the slots wrap)
        s = k % n_slots
        lab = "step %d slot %d" % (k, s)
        np.testing.assert_allclose(_np(slots["obs"][s][:n_slice]), oo[0], rtol=0, atol=TOL, err_msg=lab + " obs")
        np.testing.assert_allclose(_np(slots["node_obs"][s][:n_slice]), oo[2], rtol=0, atol=TOL, err_msg=lab + " node")
        adj = _np(slots["adj"][s][:n_slice])
        ref = np.broadcast_to(oo[3][:, None], adj.shape)
        np.testing.assert_allclose(adj, ref, rtol=0, atol=TOL, err_msg=lab + " adj")
        np.testing.assert_array_equal(adj == 0, ref == 0, err_msg=lab + " adj mask")
        np.testing.assert_allclose(_np(slots["reward"][s][:n_slice]), oo[4], rtol=6e-8, atol=TOL, err_msg=lab + " rew")
        np.testing.assert_array_equal(_np(slots["done"][s][:n_slice]).astype(bool), oo[5], err_msg=lab + " done")
        np.testing.assert_array_equal(_np(slots["agent_id"][s][:n_slice])[..., 0], oo[1][..., 0], err_msg=lab + " ids")
        np.testing.assert_allclose(_np(slots["info"][s][:n_slice]), oo[6], rtol=2e-6, atol=2e-5, err_msg=lab + " info")
    from test_gpu_parity import STATE_F, STATE_I
    for f in STATE_F + ["tube", "landmarks", "obstacles", "goal_min_time", "delta_spacing", "prev_proj"]:
        np.testing.assert_allclose(eng.get(f)[:n_slice], orc.get(f), rtol=0, atol=1e-9, err_msg="final " + f)
    for f in STATE_I:
        np.testing.assert_array_equal(eng.get(f)[:n_slice], orc.get(f), err_msg="final " + f)
    # all envs, not only the slice: the slots' adjacency is symmetric, zero-diagonal and one matrix per env in every surviving slot
    for s in range(min(n_slots, 3)):
        adj = slots["adj"][s]
        assert torch.equal(adj, adj.transpose(-1, -2)) and torch.equal(adj, adj[:, :1].expand_as(adj))
    eng.check_errors()
    return eng, n_resets


@pytest.mark.parametrize("scen", ["navigation_graph", JULY], ids=["c2", "c3"])
def test_timed_launch_shape_full_size_26_slots_vs_oracle(scen):
    """the EXACT launch the driver's bench line is timed on — 4096 x 10, one rollout launch into slot-per-step storage [26, ...]
    (2.5 GB: nontemporal, output-order adjacency stores), K = 60 so the slots wrap twice — against a 512-env oracle slice in every surviving slot."""
    eng, n_resets = _bench_shape_rollout(dict(scenario_name=scen, num_agents=10, world_size=4.0, episode_length=25, seed=1234), 4096, 512, 60, 26)
    assert n_resets >= 2 * 512                                     # two all-env reset steps inside the launch
    assert eng.tuning()["ap"] == 10 and eng.tuning()["roll"] == 1


def test_timed_launch_shape_c4_slots_vs_oracle():
    """The c4 line's launch (bench.py --workload c4): 8192 envs x (32 agents + 8 obstacles + 4 walls), rollout into as many slots as the bench's memory rule
    allows (26 x 6.1 GB = 158 GB where it fits), slots wrapping; a 64-env oracle slice in every surviving slot."""
    import torch
    kw = dict(scenario_name="navigation_graph", num_agents=32, num_obstacles=8, num_walls=4, world_size=8.0, episode_length=25, seed=1234)
    cfg = gmpe.make_config(num_envs=8192, **kw)
    A, E = 32, 72
    step_bytes = 8192 * (A * 13 * 4 + A * 4 + A * E * 8 * 4 + A * E * E * 4 + A * 4 + A + A * 18 * 4)
    budget = min(int(torch.cuda.mem_get_info()[0] * 0.75), 200 << 30)       # bench.py's rule
    n_slots = 26 if step_bytes * 26 <= budget else int(max(4, budget // step_bytes))
    K = n_slots + 4
    eng, n_resets = _bench_shape_rollout(kw, 8192, 64, K, n_slots)
    assert n_resets >= 64


def test_prepared_rollout_is_the_same_launch_and_refuses_a_closed_engine():
    """engine.prepare_rollout: argument structs built once, a call is one C call — same bits as rollout(); launching after close() raises instead of passing a freed handle."""
    import torch
    from gmpe._lib import GmpeError
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=33, num_agents=10, world_size=4.0, episode_length=6, seed=5)
    e1, e2 = _engine(cfg), _engine(cfg)
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    acts = torch.randint(0, cfg.n_actions, (4, 33, 10), generator=g, device="cuda", dtype=torch.int32)
    launch = e2.prepare_rollout(acts, 9)
    for rep in range(3):
        o1 = e1.rollout(acts, 9); launch()
        torch.cuda.synchronize()
        for key in OUT_KEYS:
            assert torch.equal(getattr(o1, key), getattr(e2.out, key)), (rep, key)
    _compare_state(e1, e2, "prepared launch")
    e2.close()
    with pytest.raises(GmpeError, match="closed"):
        launch()


@pytest.mark.parametrize("scen,ws,tape_len,expect", [("navigation_graph", 2.5, 0, "clean"), (JULY, 3.0, 0, "clean"), ("navigation_graph", 2.0, 0, "forced"), (JULY, 2.5, 0, "forced"),
                                                     ("navigation_graph", 4.0, 170, "tape"), (JULY, 4.0, 150, "tape")],
                         ids=["nav-ws2.5", "july-ws3", "nav-ws2-forced", "july-ws2.5-forced", "nav-tape-runs-out", "july-tape-runs-out"])
def test_batched_placement_crowded_worlds_and_short_tapes_vs_oracle(scen, ws, tape_len, expect):
    """The exact-size rollout kernels of navigation_graph / July place a resetting env's entities in batches of A attempts from draws the streaming waves
    prefilled (reset_world_coop<SC, true>). Crowded worlds make a reset consume MORE draws than the 200-draw buffer holds (evaluated in place) and hit the
    GMPE_MAX_TRIES forced accepts (error bit 2, which the oracle's literal loop sets for the same envs); a short RNG tape runs out inside the rollout (error bit 1,
    draws past its end read 0.5). Placements, RNG counters, error flags and every output must equal the oracle's sequential loops (…_july.py:440-515)."""
    import torch
    N, A, K = 44, 10, 11
    if expect == "tape" and scen == "navigation_graph":
        K = 8        # the tape runs out in the LAST step's auto-reset: past its end every draw is 0.5, the agents coincide and the next step's contact forces are 0 / 0 in engine and oracle alike
    cfg = gmpe.make_config(scenario_name=scen, num_envs=N, num_agents=A, world_size=ws, episode_length=4, seed=311)
    eng, orc = _engine(cfg), ol.Oracle(cfg)
    assert eng.tuning()["ap"] == 10 and eng.tuning()["block_roll"] == 256           # the instantiation with the batched placement
    if tape_len:
        tape = np.random.RandomState(12).rand(N, tape_len)
        eng.set_tape(tape); orc.set_tape(tape)
    eng.reset(); orc.reset()
    rng = np.random.RandomState(6)
    acts = rng.randint(0, cfg.n_actions, (K, N, A)).astype(np.int32)
    st = _rollout_into_slots(eng, torch.as_tensor(acts, device="cuda"), K, K)
    biggest_reset, prev = 0, orc.get("rng_ctr").copy()
    for k in range(K):
        oo = orc.step(acts[k])
        now = orc.get("rng_ctr")
        biggest_reset, prev = max(biggest_reset, int((now - prev).max())), now.copy()   # draws of one step: <= A heading re-draws + an auto-reset's placement
        np.testing.assert_allclose(_np(st["obs"][k]), oo[0], rtol=0, atol=TOL, err_msg="obs %d" % k)
        np.testing.assert_allclose(_np(st["node_obs"][k]), oo[2], rtol=0, atol=TOL, err_msg="node %d" % k)
        np.testing.assert_allclose(_np(st["adj"][k]), np.broadcast_to(oo[3][:, None], st["adj"][k].shape), rtol=0, atol=TOL, err_msg="adj %d" % k)
        np.testing.assert_allclose(_np(st["reward"][k]), oo[4], rtol=0, atol=TOL, err_msg="rew %d" % k)
        np.testing.assert_array_equal(_np(st["done"][k]).astype(bool), oo[5], err_msg="done %d" % k)
    _compare_state(eng, orc, "end")                                                  # positions, landmarks, rng_ctr and error_flags included
    ctr, err = orc.get("rng_ctr"), orc.get("error_flags")
    assert biggest_reset > 200 + A                                                   # an auto-reset inside the rollout drew past the prefilled buffer
    if expect == "tape":
        assert (err & 1).any() and (ctr > tape_len).any()                            # the tape did run out inside the rollout
    elif expect == "forced":
        assert (err & 2).any()                                                       # forced accepts happened
    else:
        assert not err.any()
