"""CPU-side tests: C-ABI surface, config POD layout, host mirrors of the reference interface,
sharding + gather (gloo, world_size 2), oracle self-consistency. No GPU compute."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import gmpe
from gmpe import config as gcfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "gmpe.h")
SO = os.path.join(ROOT, "contracts-marl-aam-corridors_amd", "libgmpe.so")


def _declared_functions():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gmpe_[a-z_0-9]+)\s*\(", txt)))


def _load_so():
    if not os.path.exists(SO):
        import __graft_entry__ as g
        g.build()
    import torch  # noqa: F401  libamdhip64 first (same runtime as torch)
    return C.CDLL(SO)


def test_library_exports_every_declared_symbol():
    lib = _load_so()
    fns = _declared_functions()
    assert len(fns) >= 15
    for f in fns:
        assert hasattr(lib, f), "libgmpe.so does not export %s declared in include/gmpe.h" % f
    assert lib.gmpe_abi_version() == gcfg.ABI_VERSION


def test_binding_symbol_list_matches_header():
    from gmpe import _lib
    assert sorted(_lib.SYMBOLS) == _declared_functions()


def test_config_pod_layout_matches_c_header():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "gmpe.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(gmpe_config), offsetof(gmpe_config, seed),
             offsetof(gmpe_config, world_size), offsetof(gmpe_config, dt), offsetof(gmpe_config, ang_rate_opt),
             offsetof(gmpe_config, sensitivity), offsetof(gmpe_config, walls), sizeof(gmpe_outputs));
      printf("%d %d\n", (int)GMPE_F_ERROR_FLAGS, (int)GMPE_F_COUNT);
      printf("%zu %zu %zu %zu %zu\n", offsetof(gmpe_config, graph_feat_type), offsetof(gmpe_config, agent_size), offsetof(gmpe_config, action_force_scale),
             sizeof(gmpe_rollout), sizeof(gmpe_tuning));
      printf("%zu\n", offsetof(gmpe_config, formation_type));
      return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        cpath = os.path.join(td, "t.c")
        open(cpath, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, cpath])
        out = subprocess.check_output([exe]).decode().split()
    G = gcfg.GmpeConfig
    from gmpe._lib import GmpeOutputs
    assert [int(x) for x in out[:8]] == [C.sizeof(G), G.seed.offset, G.world_size.offset, G.dt.offset,
                                         G.ang_rate_opt.offset, G.sensitivity.offset, G.walls.offset,
                                         C.sizeof(GmpeOutputs)]
    assert int(out[8]) == gcfg.FIELDS["error_flags"][0] and int(out[9]) == len(gcfg.FIELDS)
    from gmpe._lib import GmpeRollout, GmpeTuning
    assert [int(x) for x in out[10:15]] == [G.graph_feat_type.offset, G.agent_size.offset, G.action_force_scale.offset,
                                            C.sizeof(GmpeRollout), C.sizeof(GmpeTuning)]
    assert int(out[15]) == G.formation_type.offset                       # ABI 3


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a GPU the product path raises instead of computing."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib = _load_so()
    lib.gmpe_last_error.restype = C.c_char_p
    cfg = gmpe.make_config(num_agents=3)
    h = C.c_void_p()
    rc = lib.gmpe_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -3 and b"no CPU fallback" in lib.gmpe_last_error()
    from gmpe._lib import GmpeError
    from gmpe.engine import GmpeEngine
    with pytest.raises(GmpeError):
        GmpeEngine(cfg)


def test_bench_refuses_to_run_without_a_gpu():
    """No CPU fallback anywhere on the product path: bench.py must fail, not time the oracle, when no GPU is visible."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not any(ln.startswith("{") for ln in out.stdout.splitlines())


def test_create_rejects_bad_configs():
    lib = _load_so()
    lib.gmpe_last_error.restype = C.c_char_p
    h = C.c_void_p()
    cfg = gmpe.make_config(num_agents=3)
    cfg.abi_version = 99
    assert lib.gmpe_create(C.byref(cfg), 0, C.byref(h)) == -1
    cfg = gmpe.make_config(num_agents=3)
    cfg.num_agents = 65
    assert lib.gmpe_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert lib.gmpe_create(None, 0, C.byref(h)) == -1


def test_config_from_args_mirrors_reference_fields():
    import argparse
    a = argparse.Namespace(scenario_name="nav_metered_one_goal_graph_rotate_tube_july", dynamics_type="air_taxi",
                           world_size=4, num_agents=10, num_landmarks=10, num_scripted_agents=0, num_obstacles=0,
                           num_walls=0, collaborative=False, max_speed=2, collision_rew=5, formation_rew=1,
                           goal_rew=5, use_dones=False, episode_length=25, num_env_steps=10000,
                           n_rollout_threads=128, render_episodes=None, fair_wt=1, fair_rew=1,
                           formation_type="point", total_actions=5, zeroshift=5, graph_feat_type="relative",
                           discrete_action=True, use_safety_filter=False, seed=3)
    c = gmpe.config_from_args(a)
    assert (c.num_envs, c.num_agents, c.num_entities, c.obs_dim, c.n_actions) == (128, 10, 20, 19, 25)
    assert c.dt == 1.0 and abs(c.sep_dist - 0.4572) < 1e-12 and c.goal_thresh == 0.35
    assert gcfg.algorithmic_bytes_per_env_step(c) == 24250
    # graph_feat_type='global' (…_july.py:1672-1691) is a node-row variant: F = 7; use_safety_filter is a vec-env concern (hook slot)
    a.graph_feat_type = "global"
    cg = gmpe.config_from_args(a)
    assert cg.graph_feat_type == 1 and cg.node_feats == 7 and gcfg.algorithmic_bytes_per_env_step(cg) == 24250 - 4 * 10 * 20
    # formation_type 'line' / 'circle' (…_july.py:492-495) reach the POD; anything else raises like the reference's commented-out branch
    for name, code in (("point", 0), ("line", 1), ("circle", 2)):
        a.formation_type = name
        assert gmpe.config_from_args(a).formation_type == code
    a.formation_type = "random"
    with pytest.raises(NotImplementedError):
        gmpe.config_from_args(a)
    a.formation_type = "point"
    a.graph_feat_type = "relative"
    a.scenario_name = "two_phase_graph"
    a.graph_feat_type = "global"                             # the rot_inv family has the same _get_entity_feat_global (rot_inv.py:1668-1687)
    assert gmpe.config_from_args(a).node_feats == 7
    a.graph_feat_type = "relative"
    c3 = gmpe.config_from_args(a)
    assert (c3.obs_dim, c3.node_feats, c3.n_actions) == (15, 7, 25)
    a.scenario_name = "nav_metered_one_goal_graph_sequential_split_tube"      # a scenario file this engine does not build
    with pytest.raises(NotImplementedError):
        gmpe.config_from_args(a)
    c2 = gmpe.make_config(scenario_name="navigation_graph", num_agents=32, num_obstacles=8, num_walls=4, world_size=8.0)
    assert c2.num_entities == 72 and c2.obs_dim == 13 and c2.n_actions == 5 and c2.dt == 0.1
    assert [c2.walls[i].orient for i in range(4)] == [0, 0, 1, 1]
    # force-path constant families: multiagent/core.py:542-548 vs classic onpolicy/envs/mpe/core.py:125-130
    assert (c2.contact_family, c2.contact_force, c2.contact_margin, c2.wall_contact_force, c2.agent_mass, c2.action_force_scale) == (0, 300.0, 0.02, 220.0, 1.0, 1.0)
    cc = gmpe.make_config(scenario_name="navigation_graph", num_agents=4, contact_family="classic", agent_size=0.15, collider_size=0.2, agent_accel=3.0)
    assert (cc.contact_family, cc.contact_force, cc.contact_margin, cc.wall_contact_force, cc.wall_contact_margin) == (1, 100.0, 1e-3, 100.0, 1e-3)
    assert (cc.agent_size, cc.collider_size, cc.agent_mass, cc.action_force_scale, cc.sensitivity) == (0.15, 0.2, 1.0, 3.0, 3.0)
    with pytest.raises(NotImplementedError):
        gmpe.make_config(contact_family="classic")              # kinematic scenario: no force path


def test_spaces_are_duck_type_compatible():
    from gmpe.spaces import Box, Discrete
    b, d = Box(-np.inf, np.inf, (19,), np.float32), Discrete(25)
    # onpolicy/utils/util.py:32-52
    assert b.__class__.__name__ == "Box" and b.shape == (19,)
    assert d.__class__.__name__ == "Discrete" and d.n == 25


def test_lazy_infos_materialise_on_access():
    import torch
    from gmpe.vec_env import LazyInfos
    K = len(gcfg.INFO_KEYS)
    raw = torch.arange(2 * 3 * K, dtype=torch.float32).reshape(2, 3, K)
    li = LazyInfos(raw, 2, 3)
    assert li._host is None and len(li) == 2
    d = li[1][2]
    assert list(d.keys()) == gcfg.INFO_KEYS[:17] and d["individual_reward"] == raw[1, 2, 0].item()   # July / navigation_graph keys
    r = LazyInfos(raw, 2, 3, include_phase=True)[1][2]                                               # rot_inv adds 'Phase_reached' (:835)
    assert list(r.keys()) == gcfg.INFO_KEYS and r["Phase_reached"] == raw[1, 2, 17].item()
    assert "Min_time_to_goal" not in LazyInfos(raw, 2, 3, include_min_time=False)[0][0]
    assert [len(x) for x in li] == [3, 3]
    # keys the runner prints (graph_mpe_runner.py:174-187)
    for k in ("Distance_mean", "Distance_variance", "Mean_by_variance", "Dist_to_goal", "individual_reward", "Num_agent_collisions"):
        assert k in li[0][0]


def test_shard_ranges_cover_all_envs():
    from gmpe.sharding import shard_range
    for n, w in ((4096, 8), (8192, 3), (10, 4), (7, 7)):
        r = [shard_range(n, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))


class _OracleBackedEngine(object):
    """Host stand-in with GmpeEngine's surface (cfg / out / rebind / step / adj_compact / device) whose `step` fills the BOUND
    output tensors from the CPU oracle — so the gather path (slab views, rebind, collective, unpack) runs on CPU with the real
    shapes and values of a real config. Test infrastructure only; the GPU twin is tests/test_gpu_vec_env.py::test_rollout_gather_real_engine."""

    def __init__(self, cfg, node_form="rows", do_reset=True, adj_form=None):
        import torch
        import oracle_lib as ol
        from gmpe.engine import StepOutputs
        self.cfg, self.adj_compact, self.device, self.node_form = cfg, True, torch.device("cpu"), node_form
        self.adj_form = "none" if adj_form == "none" else "compact"
        N, A, E, D, F = cfg.num_envs, cfg.num_agents, cfg.num_entities, cfg.obs_dim, cfg.node_feats
        self.N, self.A = N, A
        self.orc = ol.Oracle(cfg)
        if do_reset:
            self.orc.reset()
        self.out = StepOutputs(obs=torch.zeros(N, A, D), agent_id=torch.zeros(N, A, 1, dtype=torch.int32),
                               node_obs=torch.zeros(N, A, E, F) if node_form != "table" else None,
                               entity_table=torch.zeros(N, cfg.entity_table_width, dtype=torch.float64) if node_form != "rows" else None,
                               adj=None if adj_form == "none" else torch.zeros(N, E, E), reward=torch.zeros(N, A), done=torch.zeros(N, A, dtype=torch.uint8), info=None)

    def rebind(self, o):
        for k in ("obs", "node_obs", "entity_table", "adj", "reward", "done"):
            a, b = getattr(o, k), getattr(self.out, k)
            assert (a is None) == (b is None), k
            assert a is None or (a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous()), k
        self.out = o

    def tuning(self):
        return {"split": 1, "roll": 0}                       # DeviceRolloutBuffer.collect then takes its insert_step loop (no rollout kernel on the CPU)

    def _fill(self, obs, node, adj):
        import torch
        o = self.out
        o.obs.copy_(torch.from_numpy(obs))
        if o.adj is not None:
            o.adj.copy_(torch.from_numpy(adj))
        if o.node_obs is not None:
            o.node_obs.copy_(torch.from_numpy(node))
        if o.entity_table is not None:
            o.entity_table.copy_(torch.from_numpy(self.orc.entity_table()))
        if o.agent_id is not None:
            o.agent_id.copy_(torch.arange(self.A, dtype=torch.int32).view(1, self.A, 1).expand(self.N, self.A, 1))

    def reset(self):
        obs, ids, node, adj = self.orc.reset()
        self._fill(obs, node, adj)
        return self.out

    def step(self, act):
        import torch
        obs, ids, node, adj, rew, done, info, did = self.orc.step(act.numpy() if torch.is_tensor(act) else act)
        self._fill(obs, node, adj)
        o = self.out
        o.reward.copy_(torch.from_numpy(rew)); o.done.copy_(torch.from_numpy(done.astype(np.uint8)))
        return o

    def masks_from_dones(self, done, masks, active_masks):
        """GraphReplayBuffer.insert's mask rules (graph_buffer.py:223-251, graph_mpe_runner.py:395-405) — what gmpe_masks_from_dones computes on the device"""
        d = done.bool()
        alld = d.all(dim=1, keepdim=True)
        masks.copy_((~d).float().view(masks.shape)); active_masks.copy_((~(d & ~alld)).float().view(active_masks.shape))


def _gather_worker(rank, world, port, q, scen, mode):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    from gmpe.sharding import RolloutGather, shard_range
    NT, A = 12, 4
    kw = dict(scenario_name=scen, num_agents=A, world_size=4.0, episode_length=4, seed=5)
    lo, hi = shard_range(NT, world, rank)
    eng = _OracleBackedEngine(gmpe.make_config(num_envs=hi - lo, env_id_base=lo, **kw))
    full = ol.Oracle(gmpe.make_config(num_envs=NT, **kw))
    full.reset()
    rg = RolloutGather(eng, world, mode=mode)
    rng = np.random.RandomState(1)
    ok = True
    for t in range(9):                                        # two auto-resets inside; slabs alternate
        act = rng.randint(0, eng.cfg.n_actions, (NT, A)).astype(np.int32)
        ref = full.step(act)
        rg.step_and_gather(act[lo:hi])
        u = rg.unpack()
        if mode == "gather" and rank != 0:
            ok = ok and u is None
            continue
        F = eng.cfg.node_feats
        ok = ok and u["node_obs"].shape == (NT, A, eng.cfg.num_entities, F) and u["done"].dtype == torch.bool
        for k, r in (("obs", ref[0]), ("node_obs", ref[2]), ("adj", ref[3]), ("reward", ref[4])):
            ok = ok and np.array_equal(u[k].numpy(), r.astype(np.float32))
        ok = ok and np.array_equal(u["done"].numpy(), ref[5])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("scen,mode", [("nav_metered_one_goal_graph_rotate_tube_july", "gather"), ("nav_graph_metered_single_corridor_rot_inv", "gather"),
                                       ("two_phase_graph", "all_gather"), ("navigation_graph", "all_gather")])
def test_rollout_gather_world_size_2_gloo(scen, mode):
    """Sharded oracle-backed engines + the gather == the unsharded run, for F = 8 and F = 7 scenarios, both collective modes."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (hash((scen, mode)) % 50)
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q, scen, mode)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    assert res == [(0, True), (1, True)]


def test_rollout_gather_rejects_unequal_shards_and_wrong_layout():
    from gmpe.sharding import slab_layout, slab_views
    import torch
    lay, nbytes = slab_layout(5, 3, 6, 13, 7)
    assert nbytes % 16 == 0 and all(lo % 16 == 0 for lo, _ in lay.values())
    v = slab_views(torch.zeros(nbytes, dtype=torch.uint8), lay, 5, 3, 6, 13, 7)
    assert v["node_obs"].shape == (5, 3, 6, 7) and v["done"].dtype == torch.uint8 and v["adj"].shape == (5, 6, 6)
    with pytest.raises(TypeError):
        slab_layout(5, 3, 6, 13)                              # F has no default any more (round-1 bug: F = 8 assumed)


# ---------------------------------------------------------------- rollout-granularity gather of the compact slab (world_size 2, gloo)
def _numpy_expand(cfg, table, out=None, out_envs=None, env_offset=0):
    """CPU stand-in for gmpe_expand_node_obs (TEST infrastructure for the gloo rehearsal only; the HIP kernel is checked bit for bit against the engine's own rows in
    tests/test_gpu_gather.py): node rows from entity tables, the three row variants of gmpe_kernel.h stream_graph_fn."""
    import torch
    t = table.numpy()
    A, L, E, F, W = cfg.num_agents, cfg.num_landmarks, cfg.num_entities, cfg.node_feats, cfg.entity_table_width
    lead, n = t.shape[:-2], t.shape[-2]
    T = t.reshape(-1, W)
    ex, ey = T[:, :E], T[:, E:2 * E]
    vox, voy, vnx, vny = (T[:, 2 * E + q * A:2 * E + (q + 1) * A] for q in range(4))
    B = T.shape[0]
    ego = np.arange(A)[None, :, None]; k = np.arange(E)[None, None, :]
    kag = k < A; kk = np.where(kag, k, 0)
    post = k <= ego
    take = lambda a: np.take_along_axis(a[:, None, :].repeat(A, 1), np.broadcast_to(kk, (B, A, E)), axis=2)
    kvx = np.where(kag, np.where(post, take(vnx), take(vox)), 0.0); kvy = np.where(kag, np.where(post, take(vny), take(voy)), 0.0)
    kx = np.broadcast_to(ex[:, None, :], (B, A, E)); ky = np.broadcast_to(ey[:, None, :], (B, A, E))
    gxa = np.take_along_axis(ex[:, None, :].repeat(A, 1), np.broadcast_to(A + kk, (B, A, E)), axis=2)
    gya = np.take_along_axis(ey[:, None, :].repeat(A, 1), np.broadcast_to(A + kk, (B, A, E)), axis=2)
    px, py = ex[:, :A, None], ey[:, :A, None]; evx, evy = vnx[:, :, None], vny[:, :, None]
    typ = np.where(kag, 0.0, np.where(k < A + L, 1.0, 2.0)) * np.ones((B, A, E))
    rot = cfg.scenario in gcfg.ROT_FAMILY
    f32 = np.float32
    if cfg.graph_feat_type == 1:
        rows = np.stack([kvx.astype(f32), kvy.astype(f32), kx.astype(f32), ky.astype(f32), np.where(kag, gxa, kx).astype(f32), np.where(kag, gya, ky).astype(f32), typ.astype(f32)], -1)
    elif rot:
        cs, sn = T[:, 2 * E + 4 * A:2 * E + 5 * A, None], T[:, 2 * E + 5 * A:2 * E + 6 * A, None]
        two = cfg.scenario == gcfg.SCENARIO_TWO_PHASE
        wx = W - (E + 31) // 32 - 2                           # two_phase: exit x, y sit right before the mask words
        gx = (T[:, wx, None, None].astype(f32) if two else gxa.astype(f32)) * np.ones((B, A, E), f32)
        gy = (T[:, wx + 1, None, None].astype(f32) if two else gya.astype(f32)) * np.ones((B, A, E), f32)
        rvx = (kvx.astype(f32) - evx.astype(f32)).astype(np.float64); rvy = (kvy.astype(f32) - evy.astype(f32)).astype(np.float64)
        rpx = (kx.astype(f32) - px.astype(f32)).astype(np.float64); rpy = (ky.astype(f32) - py.astype(f32)).astype(np.float64)
        rgx = (gx - px.astype(f32)).astype(np.float64); rgy = (gy - py.astype(f32)).astype(np.float64)
        r2 = lambda vx, vy: (cs * vx + sn * vy, -sn * vx + cs * vy)
        o0, o1 = r2(rvx, rvy); o2, o3 = r2(rpx, rpy); o4, o5 = r2(rgx, rgy)
        o4 = np.where(kag, o4, o2); o5 = np.where(kag, o5, o3)
        rows = np.stack([o0, o1, o2, o3, o4, o5, typ], -1).astype(f32)
    else:
        rows = np.stack([kvx - evx, kvy - evy, kx - px, ky - py, np.where(kag, gxa, kx) - px, np.where(kag, gya, ky) - py, np.where(kag, 0.0, 1.0) * np.ones((B, A, E)), typ], -1).astype(f32)
    rows = torch.from_numpy(rows.reshape(lead + (n, A, E, F)))
    if out is None:
        return rows
    out[..., env_offset:env_offset + n, :, :, :] = rows
    return out


def _numpy_expand_adj(cfg, table, copies=1, out=None, out_envs=None, env_offset=0):
    """CPU stand-in for gmpe_expand_adj (gloo rehearsal only): f32(sqrt(dx^2 + dy^2)) of pos[min] - pos[max], zero diagonal, masked rows / columns zeroed."""
    import torch
    t = table.numpy()
    E, W = cfg.num_entities, cfg.entity_table_width
    lead, n = t.shape[:-2], t.shape[-2]
    T = t.reshape(-1, W)
    ex, ey = T[:, :E], T[:, E:2 * E]
    words = T[:, W - (E + 31) // 32:].astype(np.uint64)
    k = np.arange(E)
    m = ((words[:, k // 32] >> (k % 32).astype(np.uint64)) & 1).astype(bool)
    lo, hi = np.minimum(k[:, None], k[None, :]), np.maximum(k[:, None], k[None, :])
    dx, dy = ex[:, lo] - ex[:, hi], ey[:, lo] - ey[:, hi]
    d = np.sqrt(dx * dx + dy * dy).astype(np.float32)
    d[:, k, k] = 0.0
    d[m[:, :, None] | m[:, None, :]] = 0.0
    rows = torch.from_numpy(d.reshape(lead + (n, E, E)))
    if out is None:
        return rows
    out[..., env_offset:env_offset + n, :, :] = rows
    return out


def _collector_worker(rank, world, port, q, scen, adj_form=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    from gmpe.sharding import ShardedRolloutCollector, shard_range
    NT, A, T = 10, 4, 6
    kw = dict(scenario_name=scen, num_agents=A, world_size=4.0, episode_length=4, seed=8)
    lo, hi = shard_range(NT, world, rank)
    eng = _OracleBackedEngine(gmpe.make_config(num_envs=hi - lo, env_id_base=lo, **kw), node_form="table", do_reset=False, adj_form=adj_form)
    full = ol.Oracle(gmpe.make_config(num_envs=NT, **kw))
    col = ShardedRolloutCollector(eng, T, world, expand=_numpy_expand, expand_adj=_numpy_expand_adj)
    ok0 = ("_adj" in col.layout) == (adj_form != "none")
    col.warmup()
    r0 = full.reset()
    rng = np.random.RandomState(2)
    ok = ok0
    F = eng.cfg.node_feats
    prev_last = None
    for rep in range(3):                                      # slabs alternate, slot 0 carried across; auto-resets inside every rollout
        acts = rng.randint(0, eng.cfg.n_actions, (T, NT, A)).astype(np.int32)
        ref = [full.step(acts[t]) for t in range(T)]
        b = col.collect_and_gather_async(torch.from_numpy(acts[:, lo:hi].copy()))
        u = col.unpack(b)
        if rank != 0:
            ok = ok and u is None
            continue
        first = (r0[0], r0[2], r0[3]) if rep == 0 else prev_last
        obs = np.stack([first[0]] + [r[0] for r in ref]).astype(np.float32)
        node = np.stack([first[1]] + [r[2] for r in ref])
        adj = np.stack([first[2]] + [r[3] for r in ref]).astype(np.float32)
        ok = ok and u["node_obs"].shape == (T + 1, NT, A, eng.cfg.num_entities, F)
        ok = ok and np.array_equal(u["obs"].numpy(), obs) and np.array_equal(u["adj"].numpy(), adj)
        ok = ok and np.allclose(u["node_obs"].numpy(), node, rtol=0, atol=1e-6)
        ok = ok and np.array_equal(u["rewards"].numpy()[..., 0], np.stack([r[4] for r in ref]).astype(np.float32))
        dn = np.stack([r[5] for r in ref])
        ok = ok and np.array_equal(u["dones"].numpy(), dn)
        alld = dn.all(axis=2, keepdims=True)
        ok = ok and np.array_equal(u["masks"].numpy()[1:, ..., 0], (~dn).astype(np.float32))
        ok = ok and np.array_equal(u["active_masks"].numpy()[1:, ..., 0], (~(dn & ~alld)).astype(np.float32))
        prev_last = (ref[-1][0], ref[-1][2], ref[-1][3])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("scen,adj_form", [("nav_metered_one_goal_graph_rotate_tube_july", None), ("two_phase_graph", None), ("navigation_graph", None),
                                           ("nav_metered_one_goal_graph_rotate_tube_july", "none"), ("nav_graph_metered_single_corridor_rot_inv", "none"), ("navigation_graph", "none")])
def test_sharded_rollout_collector_world_size_2_gloo(scen, adj_form):
    """Per-rollout gather of the compact slab (obs + entity table + one ExE adj + rewards / dones / masks — or, adj_form 'none', without the adjacency, which the learner
    rebuilds from the table), two ranks: the learner's unpacked arrays == the unsharded run's [T+1, N, ...] rollout, over three alternating slabs. Replaces
    env_wrappers.py:996-1004 at rollout granularity."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 2000) + (hash((scen, adj_form)) % 50)
    ps = [ctx.Process(target=_collector_worker, args=(r, 2, port, q, scen, adj_form)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=240) for _ in range(2))
    [p.join(60) for p in ps]
    assert res == [(0, True), (1, True)]


def test_rollout_slab_is_compact():
    """bytes a rank ships per env-step, rows form (round 3) vs compact rollout slab."""
    from gmpe.sharding import rollout_bytes_per_env_step, rollout_slab_layout
    c3 = gmpe.make_config(num_envs=4096, num_agents=10)
    rows, compact = rollout_bytes_per_env_step(c3, 25, "rows"), rollout_bytes_per_env_step(c3, 25, "compact")
    assert rows == 8810 and compact < 3300 and rows / compact > 2.6
    lay, total = rollout_slab_layout(c3, 25)
    assert total % 16 == 0 and all(o % 16 == 0 for o, _, _, _ in lay.values()) and "node_obs" not in lay and lay["entity_table"][3] == (26, 4096, 81)
    assert abs(total / (25 * 4096) - compact) < 8
    # adj_form 'none': the adjacency is rebuilt on the learner from the table's positions + mask words, the slab drops it
    table_only = rollout_bytes_per_env_step(c3, 25, "table")
    lay2, total2 = rollout_slab_layout(c3, 25, with_adj=False)
    assert "_adj" not in lay2 and table_only < 1650 and rows / table_only > 5.3 and abs(total2 / (25 * 4096) - table_only) < 8


# ---------------------------------------------------------------- oracle self-consistency (CPU)
def test_oracle_sharded_equals_unsharded_and_is_deterministic():
    import oracle_lib as ol
    N, A = 24, 5
    full = ol.Oracle(gmpe.make_config(num_envs=N, num_agents=A, seed=9, episode_length=6))
    parts = [ol.Oracle(gmpe.make_config(num_envs=N // 2, num_agents=A, seed=9, episode_length=6, env_id_base=g * (N // 2)))
             for g in range(2)]
    rf = full.reset(); rp = [p.reset() for p in parts]
    np.testing.assert_array_equal(rf[0], np.concatenate([r[0] for r in rp]))
    rng = np.random.RandomState(0)
    for t in range(15):
        act = rng.randint(0, 25, (N, A))
        of = full.step(act)
        op = [p.step(act[g * (N // 2):(g + 1) * (N // 2)]) for g, p in enumerate(parts)]
        for k in (0, 2, 3, 4, 5):
            np.testing.assert_array_equal(of[k], np.concatenate([o[k] for o in op]), err_msg="t=%d out %d" % (t, k))


def test_oracle_philox_stream_known_values():
    import oracle_lib as ol
    lib = ol.load()
    u = [lib.gmpo_philox_uniform(1234, 7, k) for k in range(2000)]
    assert all(0.0 <= x < 1.0 for x in u) and len(set(u)) == 2000
    assert abs(np.mean(u) - 0.5) < 0.03
    assert lib.gmpo_philox_uniform(1234, 7, 5) == u[5] != lib.gmpo_philox_uniform(1234, 8, 5)


def test_oracle_navigation_graph_invariants():
    """navigation_graph composition (this project's restatement): structural properties."""
    import oracle_lib as ol
    cfg = gmpe.make_config(scenario_name="navigation_graph", num_envs=16, num_agents=6, num_obstacles=3, num_walls=4,
                           world_size=3.0, episode_length=10, seed=2)
    o = ol.Oracle(cfg)
    obs, ids, node, adj = o.reset()
    E = cfg.num_entities
    assert obs.shape == (16, 6, 13) and node.shape == (16, 6, E, 8) and adj.shape == (16, E, E)
    np.testing.assert_array_equal(adj, adj.transpose(0, 2, 1))
    assert (np.einsum("nii->ni", adj) == 0).all()
    assert (node[:, :, :6, 7] == 0).all() and (node[:, :, 6:12, 7] == 1).all() and (node[:, :, 12:, 7] == 2).all()
    rng = np.random.RandomState(3)
    for t in range(25):
        obs, ids, node, adj, rew, done, info, did = o.step(rng.randint(0, 5, (16, 6)))
        assert (rew >= -20 - 1e-9).all() and (rew <= 25 + 1e-9).all()
        assert np.isfinite(obs).all() and np.isfinite(node).all() and np.isfinite(adj).all()
        # speed clamp (integrate_state core.py:836-841)
        sp = np.hypot(o.get("s2"), o.get("s3"))
        assert (sp <= cfg.max_speed + 1e-12).all()
    assert (o.get("error_flags") == 0).all()


def test_oracle_under_sanitizers():
    """AddressSanitizer + UBSan on the CPU build of the oracle (the GPU pool has no ASan)."""
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan-run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("checksum") == 5 and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


def test_kernels_do_not_spill():
    """Every instantiation of the fused kernel must fit the register file (a spill costs 3x on this kernel:
    the inlined per-agent code is large and a non-inlined closure silently moves its state to scratch)."""
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "contracts-marl-aam-corridors_amd", "csrc"), "resource-usage"],
                         capture_output=True, text=True, timeout=600)
    txt = out.stdout + out.stderr
    names = re.findall(r"Function Name: (\S*k_env\S*)", txt)
    scratch = re.findall(r"Function Name: \S*k_env\S*.*?ScratchSize \[bytes/lane\]: (\d+)", txt, flags=re.S)
    assert len(names) >= 96 and len(scratch) == len(names)      # 6 scenario variants x (3 tile shapes x 3 exact sizes + the steady-state one + 3 rollout ones)
    # Zero everywhere except two exact-size instantiations that sit at their 128-VGPR cap with a few dwords in cold paths and were
    # measured FASTER on one box than their spill-free alternatives (profiles/README.md): the July step kernels (3 dwords; c3 closed
    # loop 26.27 us vs 26.95 run-time sizes / 27.05 at three waves per SIMD) and navigation_graph's rollout kernel (5 dwords; c2
    # 17.0 us per step vs 18.3 at three tiles per CU without a spill / 19.9 run-time sizes).
    # Round 4: the compile-time-G variants (k_env<..., GC>) of those two are spill-free; rot_inv's GC = 4 step kernel is compiled for four waves per SIMD and keeps one dword
    # in scratch — measured 6 % faster than three waves without it (profiles/r04_ab_rot_family_four_waves_abk.log).
    # The steady-state step kernels (FL = 1) are scheduled for ILP (Makefile STEPFLAGS): July's run-time-G one then keeps six dwords in scratch, rot_inv's GC = 4 one three — both measured
    # faster that way (closed loop -1.2 % / -0.9 %, profiles/r04_ab_max_ilp*_abk.log).
    allowed = lambda n: (24 if "Li256ELi10ELi2ELi1ELi0E" in n else 16) if re.search(r"ELi10ELi2ELi[01]E", n) else (24 if "Li256ELi10ELi0ELi2E" in n else (12 if "Li256ELi10ELi3ELi1ELi4E" in n else 0))
    bad = [(n, x) for n, x in zip(names, scratch) if int(x) > allowed(n)]
    assert not bad, bad


def test_bench_multi_rank_branch_inits_the_process_group_before_any_gpu_work():
    """bench.py's multi-rank branch must call init_process_group before anything allocates on / launches to the GPU
    (engine creation, torch allocations, RNG); pinned on the source order of main()."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "main")
    first_init, first_gpu = None, None
    gpu_calls = {"GmpeEngine", "randint", "empty", "Generator", "synchronize", "reset", "step"}
    for node in ast.walk(main):
        if isinstance(node, ast.Call):
            f = node.func
            name = f.attr if isinstance(f, ast.Attribute) else getattr(f, "id", "")
            if name == "init_process_group":
                first_init = node.lineno if first_init is None else min(first_init, node.lineno)
            elif name in gpu_calls:
                first_gpu = node.lineno if first_gpu is None else min(first_gpu, node.lineno)
    assert first_init is not None and first_gpu is not None and first_init < first_gpu, (first_init, first_gpu)
    # and the rendezvous is the 127.0.0.1 one the launcher passes (env), never a hostname lookup
    assert "MASTER_ADDR" not in src or "127.0.0.1" in src


def test_bench_cpu_baseline_runs_at_host_scale():
    """bench.py's cpu_baseline leg: W = usable host cores worker threads over env shards of the C oracle, aggregate,
    per-core and single-thread figures, core counts stated."""
    import bench
    seen, usable, quota = bench.host_cpu_budget()
    assert 1 <= usable <= seen
    cb = bench.cpu_baseline(bench.WORKLOADS["c3"], budget_s=1.5, single_s=0.7, envs_per_worker=128)
    assert cb["kind"] == "port" and cb["cores"] == usable and cb["host_cores"] == seen and cb["unit"] == "env-steps/s"
    assert cb["value"] > 0 and cb["single_thread"] > 0 and abs(cb["per_core"] * cb["cores"] - cb["value"]) < 1e-6 * cb["value"]
    if usable >= 4:
        assert cb["value"] > 1.5 * cb["single_thread"]          # the C call releases the GIL: threads really run in parallel


def test_plain_c_consumer_of_the_abi_compiles_and_links(tmp_path):
    """examples/c_abi_consumer.c — a consumer that is neither Python nor torch — builds with gcc -std=c11 -Wall -Werror against include/gmpe.h and links against
    libgmpe.so + the HIP runtime (it RUNS in tests/test_gpu_c_consumer.py)."""
    from c_consumer_build import build_c_consumer
    exe = build_c_consumer(str(tmp_path))
    assert os.path.exists(exe)
    needed = subprocess.check_output(["ldd", exe]).decode()
    assert "libgmpe.so" in needed and "libamdhip64" in needed and "torch" not in needed
