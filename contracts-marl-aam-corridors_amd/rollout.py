"""DeviceRolloutBuffer — env-side arrays of GraphReplayBuffer kept in HBM.

Mirrors the storage the runner fills from the env (onpolicy/utils/graph_buffer.py:84-164 shapes,
:168-251 insert, :253-283 after_update; masks as computed in
onpolicy/runner/shared/graph_mpe_runner.py:384-428): `[T+1, N, A, ...]` float32/int32 slots for
obs / share_obs / node_obs / adj / agent_id / share_agent_id / masks / active_masks and `[T, N, A, 1]`
rewards. The engine's output pointers are re-bound to slot t+1 before every step, so the kernel writes
the rollout in place: no host hop and no device-to-device copy of the 16 KB/env adjacency.
Policy-side fields (rnn states, values, log-probs, actions) stay with the learner.
"""
import torch

from .config import NODE_FEATS
from .engine import StepOutputs


def storage_spec(cfg, episode_length, adj_compact, node_form, with_adj=True):
    """name -> (dtype, shape) of the env-side arrays of a rollout (graph_buffer.py:84-164 shapes). With node_form "table" the node features are kept as the fp64
    entity table [T+1, N, W] they are a pure function of (include/gmpe.h gmpe_outputs.entity_table) instead of the [T+1, N, A, E, F] rows."""
    N, A, E, D = cfg.num_envs, cfg.num_agents, cfg.num_entities, cfg.obs_dim
    T, T1 = int(episode_length), int(episode_length) + 1
    f32 = torch.float32
    spec = {"obs": (f32, (T1, N, A, D))}
    if node_form != "table":
        spec["node_obs"] = (f32, (T1, N, A, E, cfg.node_feats))
    if node_form != "rows":
        spec["entity_table"] = (torch.float64, (T1, N, cfg.entity_table_width))
    if with_adj:
        spec["_adj"] = (f32, (T1, N, E, E) if adj_compact else (T1, N, A, E, E))
    spec["agent_id"] = (torch.int32, (T1, N, A, 1))
    spec["rewards"] = (f32, (T, N, A, 1))
    spec["dones"] = (torch.uint8, (T, N, A))
    spec["masks"] = (f32, (T1, N, A, 1))
    spec["active_masks"] = (f32, (T1, N, A, 1))
    return spec


class DeviceRolloutBuffer(object):
    def __init__(self, engine, episode_length, use_centralized_V=True, storage=None):
        """storage: optional dict name -> caller-owned tensor for some or all of the arrays of storage_spec (e.g. views of ONE byte slab that a collective ships
        as a whole, sharding.ShardedRolloutCollector); anything missing is allocated here."""
        self.engine = engine
        self.T = int(episode_length)
        self.use_centralized_V = bool(use_centralized_V)
        c, dev = engine.cfg, engine.device
        self.node_form = getattr(engine, "node_form", "rows")
        storage = dict(storage or {})
        self.adj_form = getattr(engine, "adj_form", "compact" if engine.adj_compact else "full")
        self._adj = None
        for name, (dt, shape) in storage_spec(c, self.T, engine.adj_compact, self.node_form, with_adj=self.adj_form != "none").items():
            t = storage.pop(name, None)
            if t is None:
                t = (torch.ones if name in ("masks", "active_masks") else torch.zeros)(shape, dtype=dt, device=dev)
            elif tuple(t.shape) != tuple(shape) or t.dtype != dt or t.device != dev or not t.is_contiguous():
                raise ValueError("storage[%r] must be a contiguous %s tensor of shape %s on %s" % (name, dt, tuple(shape), dev))
            elif name in ("masks", "active_masks"):
                t.fill_(1.0)
            setattr(self, name if name != "node_obs" else "_node_obs", t)
        if storage:
            raise ValueError("unknown storage entries: %s" % sorted(storage))
        if self.node_form == "table":
            self._node_obs = None
        if self.node_form == "rows":
            self.entity_table = None
        self.info = torch.zeros_like(engine.out.info) if engine.out.info is not None else None
        self.step = 0
        self._prepared = {}

    @property
    def node_obs(self):
        """[T+1, N, A, E, F]. With node_form "table" the rows are expanded from the entity tables on demand (gmpe_expand_node_obs: bit-identical to the rows the engine
        would have written) — a fresh tensor every call; a learner keeps the result, a rank that only ships its rollout never asks."""
        if self._node_obs is not None:
            return self._node_obs
        return self.engine.expand_node_obs(self.entity_table)

    # ------------------------------------------------------------------ views with the reference's shapes
    @property
    def adj(self):
        """[T+1, N, A, E, E]; a zero-copy broadcast when the engine writes the compact matrix; rebuilt from the entity tables (gmpe_expand_adj, bit-identical) when
        the engine writes no adjacency at all (adj_form 'none')."""
        if self._adj is None:
            a = self.engine.expand_adj(self.entity_table)
            T1, N, E, _ = a.shape
            return a[:, :, None].expand(T1, N, self.engine.A, E, E)
        if self.engine.adj_compact:
            T1, N, E, _ = self._adj.shape
            return self._adj[:, :, None].expand(T1, N, self.engine.A, E, E)
        return self._adj

    @property
    def share_obs(self):
        """graph_mpe_runner.py:408-413: every agent sees the concatenation of all agents' obs."""
        T1, N, A, D = self.obs.shape
        if not self.use_centralized_V:
            return self.obs
        return self.obs.reshape(T1, N, 1, A * D).expand(T1, N, A, A * D)

    @property
    def share_agent_id(self):
        T1, N, A, _ = self.agent_id.shape
        if not self.use_centralized_V:
            return self.agent_id
        return self.agent_id.reshape(T1, N, 1, A).expand(T1, N, A, A)

    # ------------------------------------------------------------------ filling
    def _bind(self, slot, reward_slot):
        e = self.engine
        o = StepOutputs(obs=self.obs[slot], agent_id=self.agent_id[slot], node_obs=None if self._node_obs is None else self._node_obs[slot],
                        entity_table=None if self.entity_table is None else self.entity_table[slot], adj=None if self._adj is None else self._adj[slot],
                        reward=self.rewards[reward_slot].view(e.N, e.A) if reward_slot is not None else e.out.reward,
                        done=self.dones[reward_slot] if reward_slot is not None else e.out.done, info=self.info)
        e.rebind(o)

    def warmup(self):
        """GMPERunner.warmup (graph_mpe_runner.py:213-238): reset outputs go to slot 0."""
        self._bind(0, None)
        self.engine.reset()
        self.step = 0

    def insert_step(self, action_idx):
        """One env step written straight into slot step+1 (+ masks), GraphReplayBuffer.insert semantics."""
        t = self.step
        self._bind(t + 1, t)
        self.engine.step(action_idx)
        # masks[dones] = 0; active_masks[dones] = 0 except where the whole env is done — one small kernel instead of eight torch ops
        self.engine.masks_from_dones(self.dones[t], self.masks[t + 1], self.active_masks[t + 1])
        self.step = (t + 1) % self.T
        return self.engine.out

    def insert_external(self, obs, agent_id, node_obs, adj, rewards, dones):
        """GraphReplayBuffer.insert's env-side arguments for step outputs produced elsewhere (a replayed log, another engine):
        device or host tensors in the engine's shapes ([N,A,D], [N,A,1], [N,A,E,F], [N,E,E] or [N,A,E,E], [N,A], [N,A] bool/u8).
        Same slot placement and mask rules as insert_step (graph_buffer.py:223-251, graph_mpe_runner.py:395-405)."""
        t, e = self.step, self.engine
        dev = e.device
        put = lambda dst, src: dst.copy_(torch.as_tensor(src).to(device=dev, dtype=dst.dtype).reshape(dst.shape))
        if self._node_obs is None:
            raise ValueError("insert_external takes node_obs rows: the buffer's engine must keep them (node_form 'rows' or 'both')")
        put(self.obs[t + 1], obs); put(self.agent_id[t + 1], agent_id); put(self._node_obs[t + 1], node_obs)
        if self._adj is None:
            raise ValueError("insert_external takes an adjacency: the buffer's engine must keep one (adj_form not 'none')")
        a = torch.as_tensor(adj)
        if e.adj_compact and a.dim() == 4:
            a = a[:, 0]                                   # the A per-agent matrices are one matrix (…_july.py:1625)
        elif not e.adj_compact and a.dim() == 3:
            a = a[:, None].expand(-1, e.A, -1, -1)
        put(self._adj[t + 1], a); put(self.rewards[t], rewards); put(self.dones[t], torch.as_tensor(dones).to(torch.uint8))
        e.masks_from_dones(self.dones[t], self.masks[t + 1], self.active_masks[t + 1])
        self.step = (t + 1) % self.T

    def collect(self, action_sets, num_steps=None):
        """The runner's collect loop with a fixed action source (graph_mpe_runner.py:57-103: `for step in range(episode_length)`:
        envs.step -> buffer.insert) as ONE launch of the persistent rollout kernel: step k reads action_sets[k % S] and writes slot
        step+k+1 of every array in place, masks / active_masks included (gmpe_rollout_steps). Same results as `num_steps`
        insert_step calls. Falls back to that loop on the split big-E path."""
        K = self.T - self.step if num_steps is None else int(num_steps)
        e = self.engine
        if e.tuning()["split"] or not e.tuning()["roll"]:
            for k in range(K):
                self.insert_step(action_sets[k % action_sets.shape[0]])
            return e.out
        NA = e.N * e.A
        key = (action_sets.data_ptr(), int(action_sets.shape[0]), K, self.step)
        launch = self._prepared.get(key) if hasattr(e, "prepare_rollout") else None
        if launch is not None:                                   # the same rollout as an earlier call: one C call, no views / structs rebuilt (engine.prepare_rollout)
            launch[0]()
            last = (self.step + K - 1) % self.T
            self._bind(last + 1, last)
            self.step = (self.step + K) % self.T
            return e.out
        slot0 = StepOutputs(obs=self.obs[1], agent_id=self.agent_id[1], node_obs=None if self._node_obs is None else self._node_obs[1],
                            entity_table=None if self.entity_table is None else self.entity_table[1], adj=None if self._adj is None else self._adj[1],
                            reward=self.rewards[0].view(e.N, e.A), done=self.dones[0], info=self.info)
        strides = dict(obs=self.obs[0].numel(), agent_id=NA, node_obs=0 if self._node_obs is None else self._node_obs[0].numel(),
                       entity_table=0 if self.entity_table is None else self.entity_table[0].numel(), adj=0 if self._adj is None else self._adj[0].numel(),
                       reward=NA, done=NA, info=0, masks=NA)
        if hasattr(e, "prepare_rollout"):
            launch = e.prepare_rollout(action_sets, K, slot0=slot0, num_slots=self.T, first_slot=self.step, strides=strides,
                                       masks=self.masks[1], active_masks=self.active_masks[1])
            if len(self._prepared) >= 8:
                self._prepared.clear()
            self._prepared[key] = (launch, action_sets)         # keeps the action tensor alive: its address is part of the key
            launch()
        else:
            e.rollout(action_sets, K, slot0=slot0, num_slots=self.T, first_slot=self.step, strides=strides,
                      masks=self.masks[1], active_masks=self.active_masks[1])
        # the engine's "current outputs" are the last slot written, as after insert_step
        last = (self.step + K - 1) % self.T
        self._bind(last + 1, last)
        self.step = (self.step + K) % self.T
        return e.out

    def after_update(self):
        """graph_buffer.py:253-283: the last slot becomes slot 0 of the next rollout."""
        for buf in self._carried():
            buf[0].copy_(buf[-1])

    def _carried(self):
        return [b for b in (self.obs, self._node_obs, self.entity_table, self._adj, self.agent_id, self.masks, self.active_masks) if b is not None]

    def carry_from(self, other):
        """after_update across TWO buffers that alternate (sharding.ShardedRolloutCollector): slot 0 of this one = the last slot of `other`, and the engine's
        outputs are re-bound to this buffer's storage by the next collect / insert_step."""
        for dst, src in zip(self._carried(), other._carried()):
            dst[0].copy_(src[-1])
        self.step = 0
