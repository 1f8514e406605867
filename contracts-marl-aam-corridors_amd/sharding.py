"""Env-range sharding across the GPUs of one node + the rollout gather.

Envs never interact (the reference runs them in separate processes, env_wrappers.py:968-975), so
rank g of G owns the contiguous range [g*N/G, (g+1)*N/G) and `step` needs NO communication; the
per-env RNG streams are keyed by the GLOBAL env id, so a sharded run reproduces the unsharded one
env by env. The only exchange is collecting the rollout slab on the learner rank(s): one RCCL
all_gather (backend "nccl" on ROCm) of the COMPACT form — obs, node_obs, one ExE adj per env,
reward, done — which is A-times smaller than the materialised [N,A,E,E] adjacency.
"""
import torch


def shard_range(n_total, world, rank):
    """Contiguous env range of `rank`; remainders go to the first ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slab_layout(n_envs, A, E, D, F=8):
    """Offsets (in float32 elements) of the compact per-rank rollout slab."""
    sizes = [("obs", n_envs * A * D), ("node_obs", n_envs * A * E * F), ("adj", n_envs * E * E),
             ("reward", n_envs * A), ("done", n_envs * A)]
    off, out = 0, {}
    for k, s in sizes:
        out[k] = (off, off + s)
        off += s
    return out, off


def pack_slab(out, slab, layout):
    """Copy one step's outputs (device tensors) into the flat float32 slab."""
    for k in ("obs", "node_obs", "adj", "reward"):
        lo, hi = layout[k]
        slab[lo:hi].copy_(getattr(out, k).reshape(-1))
    lo, hi = layout["done"]
    slab[lo:hi].copy_(out.done.reshape(-1).to(torch.float32))
    return slab


def unpack_gathered(gathered, world, n_envs, A, E, D, F=8):
    """gathered: [world, slab_len] -> dict of global arrays in env order (rank-major)."""
    layout, _ = slab_layout(n_envs, A, E, D, F)
    g = gathered.reshape(world, -1)
    view = lambda k, shp: g[:, layout[k][0]:layout[k][1]].reshape((world * n_envs,) + shp)
    return {"obs": view("obs", (A, D)), "node_obs": view("node_obs", (A, E, F)), "adj": view("adj", (E, E)),
            "reward": view("reward", (A,)), "done": view("done", (A,)) > 0.5}


class RolloutGather(object):
    """step + all_gather of the compact slab. Equal shard sizes per rank (all_gather_into_tensor)."""

    def __init__(self, engine, world, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.engine, self.world = engine, world
        c = engine.cfg
        if not engine.adj_compact:
            # gather the single ExE matrix: take ego 0's copy
            self._adj_of = lambda o: o.adj[:, 0]
        else:
            self._adj_of = lambda o: o.adj
        self.layout, self.slab_len = slab_layout(c.num_envs, c.num_agents, c.num_entities, c.obs_dim)
        dev = engine.device
        self.slab = torch.empty(self.slab_len, dtype=torch.float32, device=dev)
        self.gathered = torch.empty(world * self.slab_len, dtype=torch.float32, device=dev)

    def step_and_gather(self, action_idx):
        o = self.engine.step(action_idx)

        class _O(object):
            pass
        v = _O()
        v.obs, v.node_obs, v.adj, v.reward, v.done = o.obs, o.node_obs, self._adj_of(o), o.reward, o.done
        pack_slab(v, self.slab, self.layout)
        if self.dist.get_backend(self.group) == "nccl":          # RCCL over xGMI on the GPU node
            self.dist.all_gather_into_tensor(self.gathered, self.slab, group=self.group)
        else:                                                    # gloo (CPU rehearsal / tests)
            parts = list(self.gathered.reshape(self.world, -1).unbind(0))
            self.dist.all_gather(parts, self.slab, group=self.group)
        return self.gathered

    def unpack(self):
        c = self.engine.cfg
        return unpack_gathered(self.gathered, self.world, c.num_envs, c.num_agents, c.num_entities, c.obs_dim)
