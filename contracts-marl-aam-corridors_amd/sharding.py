"""Env-range sharding across the GPUs of one node + the rollout gather.

Envs never interact (the reference runs them in separate processes, env_wrappers.py:968-975), so
rank g of G owns the contiguous range [g*N/G, (g+1)*N/G) and `step` needs NO communication; the
per-env RNG streams are keyed by the GLOBAL env id, so a sharded run reproduces the unsharded one
env by env. The only exchange is collecting the rollout slab on the learner rank — what
`GraphSubprocVecEnv.step_wait` does with `remote.recv()` from every worker process
(onpolicy/envs/env_wrappers.py:996-1004) — as ONE RCCL gather (backend "nccl" on ROCm) of the COMPACT
form: obs, node_obs, one ExE adj per env, reward, done. That is A-times smaller than the
materialised [N,A,E,E] adjacency, and a gather to the learner moves 1/world of what an all_gather would:
over xGMI (point-to-point links) the 7 peers send into the root in parallel.

The engine writes its outputs straight into the slab (`engine.rebind` onto slab views): no pack copies.
Two slabs alternate so that the gather of step k runs on RCCL's stream while step k+1 is computed.
"""
import torch

from .engine import StepOutputs


def shard_range(n_total, world, rank):
    """Contiguous env range of `rank`; remainders go to the first ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_F32 = ("obs", "node_obs", "adj", "reward")


def slab_layout(n_envs, A, E, D, F):
    """Byte offsets of the compact per-rank rollout slab: four float32 sections (16-byte aligned) and the uint8 `done` tail.
    F = node features per row (8; 7 in the rot_inv family) — taken from the engine's config, never assumed."""
    sizes = [("obs", 4 * n_envs * A * D), ("node_obs", 4 * n_envs * A * E * F), ("adj", 4 * n_envs * E * E),
             ("reward", 4 * n_envs * A), ("done", n_envs * A)]
    off, out = 0, {}
    for k, s in sizes:
        out[k] = (off, off + s)
        off = (off + s + 15) // 16 * 16
    return out, off


def slab_views(slab, layout, n_envs, A, E, D, F):
    """Typed views of one rank's slab (a contiguous uint8 tensor) with the engine's output shapes."""
    shp = {"obs": (n_envs, A, D), "node_obs": (n_envs, A, E, F), "adj": (n_envs, E, E), "reward": (n_envs, A)}
    v = {k: slab[layout[k][0]:layout[k][1]].view(torch.float32).view(shp[k]) for k in _F32}
    v["done"] = slab[layout["done"][0]:layout["done"][1]].view(n_envs, A)
    return v


def unpack_gathered(gathered, world, n_envs, A, E, D, F):
    """gathered: uint8 [world, slab_bytes] -> dict of global arrays in env order (rank-major). The float sections are
    returned per rank and concatenated (a strided byte view cannot be re-typed in place)."""
    layout, nbytes = slab_layout(n_envs, A, E, D, F)
    g = gathered.reshape(world, nbytes)
    per = [slab_views(g[r], layout, n_envs, A, E, D, F) for r in range(world)]
    out = {k: torch.cat([p[k] for p in per], dim=0) for k in _F32}
    out["done"] = torch.cat([p["done"] for p in per], dim=0) != 0
    return out


class RolloutGather(object):
    """step + gather of the compact slab to the learner rank `dst` (mode="gather", default) or to every rank
    (mode="all_gather"). Every rank must hold the same number of envs (checked at construction). The engine must write the
    compact adjacency (`GmpeEngine(..., adj_compact=True)`): the [N,A,E,E] form is a broadcast view the learner makes for free."""

    def __init__(self, engine, world, rank=None, group=None, dst=0, mode="gather"):
        import torch.distributed as dist
        if mode not in ("gather", "all_gather"):
            raise ValueError("mode must be 'gather' or 'all_gather'")
        if not engine.adj_compact:
            raise ValueError("RolloutGather needs an engine created with adj_compact=True (one ExE matrix per env is shipped)")
        self.dist, self.group, self.mode, self.dst = dist, group, mode, int(dst)
        self.engine, self.world = engine, int(world)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        c = engine.cfg
        self.dims = (c.num_envs, c.num_agents, c.num_entities, c.obs_dim, c.node_feats)
        self.layout, self.slab_bytes = slab_layout(*self.dims)
        dev = engine.device
        # equal shard sizes: all_gather_into_tensor / gather need one slab size
        n = torch.tensor([c.num_envs], dtype=torch.int64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
        lo, hi = n.clone(), n.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group); dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if int(lo.item()) != int(hi.item()):
            raise ValueError("RolloutGather: every rank must hold the same number of envs (got %d..%d)" % (int(lo.item()), int(hi.item())))
        self.slabs = [torch.zeros(self.slab_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        root = self.mode == "all_gather" or self.rank == self.dst
        self.gathered = [torch.zeros((self.world, self.slab_bytes), dtype=torch.uint8, device=dev) if root else None for _ in range(2)]
        self._outs = []
        for s in self.slabs:
            v = slab_views(s, self.layout, *self.dims)
            o = StepOutputs(obs=v["obs"], agent_id=engine.out.agent_id, node_obs=v["node_obs"], adj=v["adj"], reward=v["reward"],
                            done=v["done"], info=engine.out.info)
            for k in _F32 + ("done",):
                assert getattr(o, k).numel() == getattr(engine.out, k).numel(), k      # layout vs the engine's real shapes
            self._outs.append(o)
        self._flip = 0
        self._work = [None, None]

    def _issue(self, b):
        d, g = self.dist, self.group
        if self.mode == "all_gather":
            if d.get_backend(g) == "nccl":                       # RCCL over xGMI on the GPU node
                return d.all_gather_into_tensor(self.gathered[b].view(-1), self.slabs[b], group=g, async_op=True)
            return d.all_gather(list(self.gathered[b].unbind(0)), self.slabs[b], group=g, async_op=True)   # gloo (CPU rehearsal / tests)
        dst_global = self.dst if g is None else d.get_global_rank(g, self.dst)
        lst = list(self.gathered[b].unbind(0)) if self.rank == self.dst else None
        return d.gather(self.slabs[b], lst, dst=dst_global, group=g, async_op=True)

    def step_and_gather_async(self, action_idx):
        """Step into the next slab and start its gather; returns the buffer index. The previous gather on that slab is waited for
        first (depth-2 pipeline: the gather of step k overlaps the computation of step k+1)."""
        b = self._flip
        self._flip ^= 1
        if self._work[b] is not None:
            self._work[b].wait(); self._work[b] = None
        self.engine.rebind(self._outs[b])
        self.engine.step(action_idx)
        self._work[b] = self._issue(b)
        return b

    def wait(self, b):
        if self._work[b] is not None:
            self._work[b].wait(); self._work[b] = None
        return self.gathered[b]

    def step_and_gather(self, action_idx):
        """Blocking form: -> gathered uint8 [world, slab_bytes] on the learner rank (None elsewhere in mode="gather")."""
        return self.wait(self.step_and_gather_async(action_idx))

    def unpack(self, b=None):
        """Global arrays of the last gathered step (learner rank / every rank for all_gather)."""
        b = (self._flip ^ 1) if b is None else b
        g = self.wait(b)
        if g is None:
            return None
        return unpack_gathered(g, self.world, *self.dims)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Gather at ROLLOUT granularity, compact node form (SURVEY §8e "state + one E×E per env")
# ---------------------------------------------------------------------------------------------------------------------------------------
_ROLLOUT_SECTIONS = ("obs", "entity_table", "_adj", "rewards", "dones", "masks", "active_masks")     # agent_id is arange(A): regenerated by the learner, not shipped


def rollout_slab_layout(cfg, episode_length, with_adj=True):
    """Byte layout of one rank's rollout slab: the env-side arrays of a T-step rollout (rollout.storage_spec, node features as the fp64 entity table, and — unless the
    engine writes no adjacency at all (adj_form 'none': the matrix is rebuilt from the table on the learner) — one ExE adjacency per env) back to back, every section
    16-byte aligned. -> ({name: (offset, nbytes, dtype, shape)}, total bytes)."""
    from .rollout import storage_spec
    spec = storage_spec(cfg, episode_length, adj_compact=True, node_form="table", with_adj=with_adj)
    off, out = 0, {}
    for name in _ROLLOUT_SECTIONS:
        if name not in spec:
            continue
        dt, shape = spec[name]
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dt).element_size()
        out[name] = (off, nbytes, dt, tuple(int(s) for s in shape))
        off = (off + nbytes + 15) // 16 * 16
    return out, off


def rollout_slab_views(slab, layout):
    """Typed views of one rollout slab (contiguous uint8 tensor)."""
    return {name: slab[o:o + nb].view(dt).view(shape) for name, (o, nb, dt, shape) in layout.items()}


def rollout_bytes_per_env_step(cfg, episode_length=25, form="compact"):
    """Bytes a rank ships per env-step. form: "rows" = the round-3 slab (obs + node_obs rows + one ExE adj + reward + done), "compact" = this
    module's rollout slab (obs + entity table + one ExE adj + reward + done + masks), "table" = the same without the adjacency (adj_form 'none': rebuilt from the table
    on the learner), "materialised" = what GraphSubprocVecEnv's workers pickle (A adjacency copies)."""
    A, E, D, F, W = cfg.num_agents, cfg.num_entities, cfg.obs_dim, cfg.node_feats, cfg.entity_table_width
    if form == "rows":
        return 4 * A * D + 4 * A * E * F + 4 * E * E + 4 * A + A
    if form == "materialised":
        return 4 * A * D + 4 * A * E * F + 4 * A * E * E + 4 * A + A
    T = float(episode_length)
    return (4 * A * D + 8 * W + (4 * E * E if form == "compact" else 0) + 8 * A) * (T + 1) / T + 4 * A + A


class ShardedRolloutCollector(object):
    """The runner's collect loop (graph_mpe_runner.py:57-103) on every rank's env shard + ONE collective per T-step rollout that moves the rank's rollout slab to
    the learner rank — what replaces GraphSubprocVecEnv.step_wait's per-step `remote.recv()` loop (env_wrappers.py:996-1004) for a GraphReplayBuffer-shaped consumer
    (graph_buffer.py:168-251).

    * Launch shape: `DeviceRolloutBuffer.collect` = one launch of the persistent rollout kernel per T steps (the headline launch shape), writing straight into the slab.
    * Slab: obs + the fp64 ENTITY TABLE (the per-entity state node_obs is a pure function of: ~8x fewer bytes than the [A,E,F] rows) + one ExE adjacency per env +
      rewards / dones / masks; the learner rebuilds the node rows bit for bit with gmpe_expand_node_obs straight into its global [T+1, world*N, A, E, F] array.
      With an engine created with adj_form="none" the adjacency is not shipped either (it is a function of the table's positions + mask words: gmpe_expand_adj) —
      the slab is then obs + table + rewards / dones / masks, 1.5 KB per env-step at c2 / c3 instead of 8.6-8.8 KB of rows.
    * Two slabs alternate: the gather of rollout k runs on RCCL's stream while rollout k+1 is collected (slot 0 of the next rollout is carried over first).
    Every rank must hold the same number of envs. The engine must be created with adj_compact=True, node_form="table"."""

    def __init__(self, engine, episode_length, world, rank=None, group=None, dst=0, expand=None, expand_adj=None):
        import torch.distributed as dist
        from .rollout import DeviceRolloutBuffer
        if not engine.adj_compact or getattr(engine, "node_form", "rows") != "table":
            raise ValueError("ShardedRolloutCollector needs an engine created with adj_compact=True, node_form='table' (it ships one ExE matrix and the entity table per env-step)")
        self.dist, self.group, self.dst = dist, group, int(dst)
        self.engine, self.world, self.T = engine, int(world), int(episode_length)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        self.cfg = engine.cfg
        self.with_adj = getattr(engine, "adj_form", "compact") != "none"
        self.layout, self.slab_bytes = rollout_slab_layout(self.cfg, self.T, self.with_adj)
        dev = engine.device
        n = torch.tensor([self.cfg.num_envs], dtype=torch.int64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
        lo, hi = n.clone(), n.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group); dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if int(lo.item()) != int(hi.item()):
            raise ValueError("ShardedRolloutCollector: every rank must hold the same number of envs (got %d..%d)" % (int(lo.item()), int(hi.item())))
        self.slabs = [torch.zeros(self.slab_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.bufs = [DeviceRolloutBuffer(engine, self.T, storage=rollout_slab_views(s, self.layout)) for s in self.slabs]
        self.gathered = [torch.zeros((self.world, self.slab_bytes), dtype=torch.uint8, device=dev) if self.rank == self.dst else None for _ in range(2)]
        self._work = [None, None]
        self._flip = 0
        self._last = None
        self._expand, self._expand_adj = expand, expand_adj

    def warmup(self):
        """GMPERunner.warmup (graph_mpe_runner.py:213-238): the reset observations go to slot 0 of the first rollout."""
        self.bufs[0].warmup()
        self._flip, self._last = 0, None

    def _issue(self, b):
        d, g = self.dist, self.group
        dst_global = self.dst if g is None else d.get_global_rank(g, self.dst)
        lst = list(self.gathered[b].unbind(0)) if self.rank == self.dst else None
        return d.gather(self.slabs[b], lst, dst=dst_global, group=g, async_op=True)

    def collect_and_gather_async(self, action_sets):
        """One WHOLE T-step rollout into the next slab (ONE launch; step k reads action_sets[k % S]), then start its gather; returns the slab index. The previous gather
        of that slab is waited for first. (Partial rollouts have no place here: the slab that travels is a complete [T+1, N, ...] rollout.)"""
        b = self._flip
        self._flip ^= 1
        if self._work[b] is not None:
            self._work[b].wait(); self._work[b] = None
        if self._last is not None and self._last != b:
            self.bufs[b].carry_from(self.bufs[self._last])          # after_update across the two slabs: slot 0 <- the previous rollout's last slot
        elif self._last == b:
            self.bufs[b].after_update()
        self.bufs[b].collect(action_sets)
        self._last = b
        self._work[b] = self._issue(b)
        return b

    def wait(self, b):
        if self._work[b] is not None:
            self._work[b].wait(); self._work[b] = None
        return self.gathered[b]

    def unpack(self, b, out=None):
        """Learner rank: global arrays of rollout slab `b` in env order (rank-major) — obs [T+1, W*N, A, D], node_obs [T+1, W*N, A, E, F] (expanded from every rank's
        entity table straight into place), adj [T+1, W*N, E, E] (the [.., A, E, E] form is its broadcast view), rewards [T, W*N, A, 1], dones bool [T, W*N, A], masks,
        active_masks, agent_id. `out`: dict of preallocated tensors to fill (a learner's GraphReplayBuffer storage). None on the other ranks."""
        g = self.wait(b)
        if g is None:
            return None
        c, Wd = self.cfg, self.world
        N, A, E = c.num_envs, c.num_agents, c.num_entities
        per = [rollout_slab_views(g[r], self.layout) for r in range(Wd)]
        out = {} if out is None else out
        dev = g.device

        def dest(name, dt, shape):
            if name not in out:
                out[name] = torch.empty(shape, dtype=dt, device=dev)
            return out[name]
        for name, key in (("obs", "obs"), ("adj", "_adj"), ("rewards", "rewards"), ("masks", "masks"), ("active_masks", "active_masks")):
            if key not in self.layout:
                continue
            _, _, dt, shape = self.layout[key]
            dst = dest(name, dt, (shape[0], Wd * N) + shape[2:])
            for r in range(Wd):
                dst[:, r * N:(r + 1) * N].copy_(per[r][key])
        dn = dest("dones", torch.bool, (self.T, Wd * N, A))
        for r in range(Wd):
            dn[:, r * N:(r + 1) * N].copy_(per[r]["dones"] != 0)
        T1 = self.T + 1
        node = dest("node_obs", torch.float32, (T1, Wd * N, A, E, c.node_feats))
        expand = self._expand
        if expand is None:
            from .engine import expand_node_obs as expand
        for r in range(Wd):
            expand(c, per[r]["entity_table"].contiguous(), out=node, out_envs=Wd * N, env_offset=r * N)
        if not self.with_adj:                                        # the adjacency was not shipped: rebuild it from the tables (bit-identical, gmpe_expand_adj)
            adj = dest("adj", torch.float32, (T1, Wd * N, E, E))
            expand_a = self._expand_adj
            if expand_a is None:
                from .engine import expand_adj as expand_a
            for r in range(Wd):
                expand_a(c, per[r]["entity_table"].contiguous(), out=adj, out_envs=Wd * N, env_offset=r * N)
        if "agent_id" not in out:
            out["agent_id"] = torch.arange(A, dtype=torch.int32, device=dev).view(1, 1, A, 1).expand(T1, Wd * N, A, 1)
        return out
