"""GmpeEngine — thin Python owner of one gmpe_handle (one per GPU) + the torch output buffers.

PyTorch is plumbing only: device memory (`torch.empty(..., device=...)`), streams and
`torch.distributed`. All arithmetic happens in the HIP kernels behind include/gmpe.h.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import FIELDS, INFO_KEYS, NODE_FEATS, GmpeConfig, algorithmic_bytes_per_env_step


class StepOutputs(object):
    """Device-resident outputs of one step/reset (torch tensors on the engine's device)."""
    __slots__ = ("obs", "agent_id", "node_obs", "adj", "reward", "done", "info", "entity_table")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class GmpeEngine(object):
    def __init__(self, cfg, device=0, adj_compact=False, with_info=True, node_form="rows", adj_form=None):
        """node_form: "rows" — the engine writes node_obs [N,A,E,F] (what GraphSubprocVecEnv hands the runner); "table" — it writes the fp64 entity table
        [N,W] instead (include/gmpe.h gmpe_outputs.entity_table: the state the rows are a pure function of, ~8x fewer bytes — the form a rank ships to the learner,
        expanded there by expand_node_obs); "both" — both outputs.
        adj_form: None — by `adj_compact` ([N,E,E] or [N,A,E,E]); "none" — no adjacency output at all: the matrix is a function of the entity table too (positions +
        mask words, expand_adj), so a rank that ships the table need not write or ship it (needs node_form "table" or "both")."""
        if not isinstance(cfg, GmpeConfig):
            raise TypeError("cfg must be a gmpe.config.GmpeConfig")
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GmpeError("no MI355X visible to torch: the engine has no CPU fallback")
        self.cfg = cfg
        self.device = torch.device("cuda", int(device))
        self.N, self.A = cfg.num_envs, cfg.num_agents
        self.E, self.D = cfg.num_entities, cfg.obs_dim
        self.adj_compact = bool(adj_compact)
        if node_form not in ("rows", "table", "both"):
            raise ValueError("node_form must be 'rows', 'table' or 'both'")
        self.node_form = node_form
        if adj_form not in (None, "none"):
            raise ValueError("adj_form must be None (by adj_compact) or 'none'")
        if adj_form == "none" and node_form == "rows":
            raise ValueError("adj_form='none' needs the entity table (node_form 'table' or 'both'): the adjacency is rebuilt from it")
        self.adj_form = "none" if adj_form == "none" else ("compact" if adj_compact else "full")
        self.h = C.c_void_p()
        _lib.check(self.lib.gmpe_create(C.byref(cfg), self.device.index, C.byref(self.h)), "gmpe_create")
        N, A, E, D = self.N, self.A, self.E, self.D
        dev = self.device
        self.out = StepOutputs(
            obs=torch.empty((N, A, D), dtype=torch.float32, device=dev),
            agent_id=torch.empty((N, A, 1), dtype=torch.int32, device=dev),
            node_obs=torch.empty((N, A, E, cfg.node_feats), dtype=torch.float32, device=dev) if node_form != "table" else None,
            entity_table=torch.empty((N, cfg.entity_table_width), dtype=torch.float64, device=dev) if node_form != "rows" else None,
            adj=None if adj_form == "none" else torch.empty((N, E, E) if adj_compact else (N, A, E, E), dtype=torch.float32, device=dev),
            reward=torch.empty((N, A), dtype=torch.float32, device=dev),
            done=torch.empty((N, A), dtype=torch.uint8, device=dev),
            info=torch.empty((N, A, len(INFO_KEYS)), dtype=torch.float32, device=dev) if with_info else None)
        self._tape = None
        self._ovr = None
        self._o = self._pack(self.out)

    # ------------------------------------------------------------------ plumbing
    def _pack(self, o):
        p = lambda t: None if t is None else t.data_ptr()
        return _lib.GmpeOutputs(p(o.obs), p(o.agent_id), p(o.node_obs), p(o.adj), p(o.reward), p(o.done),
                                p(o.info), int(self.adj_compact), 0, p(o.entity_table))

    def rebind(self, outputs):
        """Point the engine at other caller-owned output tensors (same shapes/dtypes, same device)."""
        ref = self.out
        for k in StepOutputs.__slots__:
            a, b = getattr(ref, k), getattr(outputs, k)
            if (a is None) != (b is None):
                raise ValueError("output %r: presence differs" % k)
            if a is not None and (tuple(a.shape) != tuple(b.shape) or a.dtype != b.dtype or b.device != self.device or not b.is_contiguous()):
                raise ValueError("output %r must be a contiguous %s tensor of shape %s on %s" % (k, a.dtype, tuple(a.shape), self.device))
        self.out = outputs
        self._o = self._pack(outputs)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmpe_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ hot path
    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, dtype=torch.uint8, device=self.device).contiguous()
        _lib.check(self.lib.gmpe_reset(self.h, None if m is None else m.data_ptr(), C.byref(self._o),
                                       self._stream()), "gmpe_reset")
        return self.out

    def step(self, action_idx):
        """action_idx: int32 device tensor [N,A]."""
        a = action_idx
        if a.dtype != torch.int32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.int32).contiguous()
        if a.numel() != self.N * self.A:
            raise ValueError("action_idx must have N*A = %d elements" % (self.N * self.A))
        _lib.check(self.lib.gmpe_step(self.h, a.data_ptr(), C.byref(self._o), self._stream()), "gmpe_step")
        return self.out

    def step_envs(self, action_idx, env_lo, env_hi, stream=None):
        """Step the envs [env_lo, env_hi) only (whole-batch `action_idx` [N,A] and outputs; rows outside the range untouched), on `stream`
        (a torch.cuda.Stream; default: the current one). Ranges are independent: a runner can double-buffer halves of the batch."""
        a = action_idx
        if a.dtype != torch.int32 or not a.is_contiguous() or a.device != self.device or a.numel() != self.N * self.A:
            raise ValueError("action_idx must be a contiguous int32 device tensor with N*A = %d elements" % (self.N * self.A))
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        _lib.check(self.lib.gmpe_step_envs(self.h, a.data_ptr(), C.byref(self._o), int(env_lo), int(env_hi), st), "gmpe_step_envs")
        return self.out

    def step_many_ranges(self, action_sets, num_steps, parts=2):
        """num_steps open-loop steps with the batch cut into `parts` env ranges, each on its own side stream (gmpe_step_many_envs)."""
        a = action_sets
        self._check_action_sets(a)
        _lib.check(self.lib.gmpe_step_many_envs(self.h, a.data_ptr(), int(num_steps), int(a.shape[0]), C.byref(self._o), int(parts), self._stream()),
                   "gmpe_step_many_envs")
        return self.out

    def _check_action_sets(self, a):
        if a.dtype != torch.int32 or not a.is_contiguous() or a.device != self.device or a.dim() != 3:
            raise ValueError("action_sets must be a contiguous int32 device tensor [S, N, A]")
        if a.shape[1] * a.shape[2] != self.N * self.A:
            raise ValueError("action_sets must be [S, %d, %d]" % (self.N, self.A))

    def step_many_prepare(self, action_sets, num_steps):
        """Record the launches of step_many(action_sets, num_steps) into a hipGraph (once, off the step path); later
        step_many calls with the same tensor / count / outputs replay it with one graph launch."""
        self._check_action_sets(action_sets)
        _lib.check(self.lib.gmpe_step_many_prepare(self.h, action_sets.data_ptr(), int(num_steps), int(action_sets.shape[0]),
                                                   C.byref(self._o)), "gmpe_step_many_prepare")

    def step_many(self, action_sets, num_steps):
        """Open-loop rollout: `num_steps` steps enqueued by one C call — one launch of the persistent rollout kernel unless the
        handle's tuning says otherwise; step k uses action_sets[k % len] (int32 device tensor [S, N, A]). Returns the outputs of
        the LAST step."""
        a = action_sets
        self._check_action_sets(a)
        _lib.check(self.lib.gmpe_step_many(self.h, a.data_ptr(), int(num_steps), int(a.shape[0]), C.byref(self._o),
                                           self._stream()), "gmpe_step_many")
        return self.out

    def rollout(self, action_sets, num_steps, slot0=None, num_slots=1, first_slot=0, strides=None, masks=None, active_masks=None):
        """K steps in ONE launch (gmpe_rollout_steps, the persistent rollout kernel). Step k reads action_sets[k % S] and writes
        output slot (first_slot + k) % num_slots: `slot0` = StepOutputs of slot 0 (default: the engine's own buffers),
        `strides` = dict output-name -> elements between consecutive slots (default 0). Bit-identical to `num_steps` step() calls."""
        self.prepare_rollout(action_sets, num_steps, slot0, num_slots, first_slot, strides, masks, active_masks)()
        return self.out

    def prepare_rollout(self, action_sets, num_steps, slot0=None, num_slots=1, first_slot=0, strides=None, masks=None, active_masks=None):
        """The launch of rollout(...) with these arguments as a callable: the argument structs (gmpe_rollout, gmpe_outputs) are built once, a call is ONE C call
        (gmpe_rollout_steps on the current stream). For collect loops that launch the same rollout every episode — building slot views and structs in Python costs
        ~30 us per launch, 7 % of a 20-step rollout at c2. The tensors are referenced, not copied: their contents are read / written at launch time."""
        a = action_sets
        self._check_action_sets(a)
        o = self._o if slot0 is None else self._pack(slot0)
        st = strides or {}
        r = _lib.GmpeRollout(int(num_steps), int(a.shape[0]), int(num_slots), int(first_slot),
                             int(st.get("obs", 0)), int(st.get("agent_id", 0)), int(st.get("node_obs", 0)), int(st.get("adj", 0)),
                             int(st.get("reward", 0)), int(st.get("done", 0)), int(st.get("info", 0)), int(st.get("masks", 0)),
                             None if masks is None else masks.data_ptr(), None if active_masks is None else active_masks.data_ptr(),
                             int(st.get("entity_table", 0)))
        keep = (a, slot0, masks, active_masks)                           # the tensors behind the raw pointers stay alive with the callable
        eng, fn, ap, rp, op, stream, check = self, self.lib.gmpe_rollout_steps, a.data_ptr(), C.byref(r), C.byref(o), self._stream, _lib.check

        def launch(_keep=keep, _r=r, _o=o):
            h = eng.h                                                    # read at launch time: a closed engine must fail loudly, not hand the C side a freed handle
            if not h:
                raise _lib.GmpeError("prepared rollout launched after the engine was closed")
            check(fn(h, ap, rp, op, stream()), "gmpe_rollout_steps")
        return launch

    def tuning(self):
        """What gmpe_create chose (tile shape, store flavour, split / rollout paths) as a dict."""
        t = _lib.GmpeTuning()
        _lib.check(self.lib.gmpe_get_tuning(self.h, C.byref(t)), "gmpe_get_tuning")
        return {k: int(getattr(t, k)) for k, _ in _lib.GmpeTuning._fields_ if k != "reserved"}

    def step_many_loop(self, action_sets, num_steps):
        """step_many as one kernel launch per step (the closed-loop launch shape; replays a prepared hipGraph when there is one)."""
        a = action_sets
        self._check_action_sets(a)
        _lib.check(self.lib.gmpe_step_many_launches(self.h, a.data_ptr(), int(num_steps), int(a.shape[0]), C.byref(self._o),
                                                    self._stream()), "gmpe_step_many_launches")
        return self.out

    def step_onehot(self, onehot):
        """onehot: float32 device tensor [N,A,n_actions] (argmax fused into the kernel)."""
        a = onehot
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        if a.numel() != self.N * self.A * self.cfg.n_actions:
            raise ValueError("onehot must be [N,A,%d]" % self.cfg.n_actions)
        _lib.check(self.lib.gmpe_step_onehot(self.h, a.data_ptr(), C.byref(self._o), self._stream()),
                   "gmpe_step_onehot")
        return self.out

    # ------------------------------------------------------------------ state access
    def get(self, name):
        fid, dt, shp = FIELDS[name]
        a = np.empty(shp(self.cfg), dtype=dt)
        _lib.check(self.lib.gmpe_get_field(self.h, fid, a.ctypes.data_as(C.c_void_p), a.nbytes), "gmpe_get_field")
        return a

    def set(self, name, value):
        fid, dt, shp = FIELDS[name]
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=dt), shp(self.cfg)))
        _lib.check(self.lib.gmpe_set_field(self.h, fid, a.ctypes.data_as(C.c_void_p), a.nbytes), "gmpe_set_field")

    def get_state(self):
        return {k: self.get(k) for k in FIELDS}

    def set_state(self, state):
        for k, v in state.items():
            self.set(k, v)

    def set_tape(self, tape):
        """Parity mode: f64 [N, len] of [0,1) samples replayed instead of Philox (None to disable)."""
        if tape is None:
            self._tape = None
            _lib.check(self.lib.gmpe_set_rng_tape(self.h, None, 0), "gmpe_set_rng_tape")
            return
        t = torch.as_tensor(np.asarray(tape, dtype=np.float64).reshape(self.N, -1), device=self.device).contiguous()
        torch.cuda.synchronize(self.device)
        self._tape = t
        _lib.check(self.lib.gmpe_set_rng_tape(self.h, t.data_ptr(), t.shape[1]), "gmpe_set_rng_tape")

    def set_control_override(self, ctrl=None, use=None):
        """Safety-filter hook slot (multiagent/core.py:692-736): `ctrl` float64 device tensor [N,A,2] integrated instead of the decoded
        action wherever `use` (uint8 [N,A]; None = everywhere) is non-zero. None removes the hook. The tensors are read at step time
        (the engine keeps references)."""
        if ctrl is None:
            self._ovr = None
            _lib.check(self.lib.gmpe_set_control_override(self.h, None, None), "gmpe_set_control_override")
            return
        if ctrl.dtype != torch.float64 or tuple(ctrl.shape) != (self.N, self.A, 2) or not ctrl.is_contiguous() or ctrl.device != self.device:
            raise ValueError("ctrl must be a contiguous float64 device tensor [N, A, 2]")
        if use is not None and (use.dtype != torch.uint8 or use.numel() != self.N * self.A or not use.is_contiguous() or use.device != self.device):
            raise ValueError("use must be a contiguous uint8 device tensor [N, A]")
        self._ovr = (ctrl, use)
        _lib.check(self.lib.gmpe_set_control_override(self.h, ctrl.data_ptr(), None if use is None else use.data_ptr()), "gmpe_set_control_override")

    def state_tensor(self, name):
        """Zero-copy torch view of a state field in HBM (gmpe_field_device_ptr), e.g. for an on-device safety filter that reads
        x / y / s2 / s3 before the step. Same stream as the engine's launches; do not resize or free."""
        fid, dt, shp = FIELDS[name]
        p = C.c_void_p()
        _lib.check(self.lib.gmpe_field_device_ptr(self.h, fid, C.byref(p)), "gmpe_field_device_ptr")
        shape = tuple(int(x) for x in shp(self.cfg))

        class _Blob(object):
            __cuda_array_interface__ = {"shape": shape, "typestr": np.dtype(dt).str, "data": (int(p.value), False), "version": 2, "strides": None}
        return torch.as_tensor(_Blob(), device=self.device)

    def check_errors(self):
        e = self.get("error_flags")
        if (e & 1).any():
            raise _lib.GmpeError("RNG tape exhausted in %d env(s)" % int((e & 1).astype(bool).sum()))
        if (e & 2).any():
            raise _lib.GmpeError("reset placement gave up in %d env(s) (world too small for the agents)"
                                 % int((e & 2).astype(bool).sum()))

    # ------------------------------------------------------------------ edges (learner-side process_adj)
    def edges_from_adj(self, adj, max_edge_dist, inclusive=False, cap=None, index64=False):
        """process_adj's edge set (gnn_new.py:329-358) of a materialised [B, E, E] (or [N, A, E, E]) adjacency: -> (edge_index [2, M] int32 — int64
        with index64=True, what torch.nonzero / PyG message passing use —, edge_attr [M], M). Prefer edges_from_adj_compact when the
        engine writes the compact adjacency: it reads 1/A of the bytes."""
        adj = adj.reshape(-1, adj.shape[-2], adj.shape[-1]).contiguous()
        B, E = adj.shape[0], adj.shape[-1]
        cap = int(cap if cap is not None else B * E * E)
        if cap > 2 ** 31 - 1:
            raise ValueError("edges_from_adj: cap %d does not fit the int32 ABI argument; pass an explicit cap (or use edges_from_adj_compact)" % cap)
        ei = torch.empty((2, cap), dtype=torch.int32, device=self.device)
        ea = torch.empty((cap,), dtype=torch.float32, device=self.device)
        ne = torch.zeros((1,), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.gmpe_edges_from_adj(self.h, adj.data_ptr(), B, E, float(max_edge_dist), int(inclusive),
                                                ei.data_ptr(), ea.data_ptr(), cap, ne.data_ptr(), self._stream()),
                   "gmpe_edges_from_adj")
        m = int(ne.item())
        ei = ei[:, :min(m, cap)]
        return (ei.long() if index64 else ei), ea[:min(m, cap)], m

    def edges_from_adj_compact(self, adj_compact, copies, max_edge_dist, inclusive=False, cap=None, index64=True):
        """process_adj's edge set (gnn_new.py:329-358) for the [N*copies, E, E] batch the runner feeds the GNN, computed from the
        compact [N, E, E] adjacency: -> (edge_index [2, M] int64 (or int32), edge_attr [M], M)."""
        adj = adj_compact.reshape(-1, adj_compact.shape[-2], adj_compact.shape[-1]).contiguous()
        N, E = adj.shape[0], adj.shape[-1]
        ne = torch.zeros((1,), dtype=torch.int32, device=self.device)

        def run(ei, ea, cap):
            _lib.check(self.lib.gmpe_edges_from_adj_compact(self.h, adj.data_ptr(), N, int(copies), E, float(max_edge_dist), int(inclusive),
                                                            int(bool(index64)), ei.data_ptr(), ea.data_ptr(), cap, ne.data_ptr(), self._stream()),
                       "gmpe_edges_from_adj_compact")
            m = int(ne.item())
            if m >= 2 ** 31 - 1:
                raise _lib.GmpeError("edges_from_adj_compact: more than 2^31 - 2 edges (the count saturates, include/gmpe.h): split the batch")
            return m
        if cap is None and N * copies * E * E > 2 ** 28:
            # the worst case (every pair an edge) would be tens of GB at the big configs (c5 shard: 34 GB of int64 ids): size the buffers from a
            # count-only call (cap = 0 writes nothing) instead
            dummy = torch.empty((2,), dtype=torch.int64, device=self.device)
            cap = run(dummy, dummy, 0)
        cap = int(cap if cap is not None else N * copies * E * E)
        if cap == 0:
            # an empty edge set is a valid answer (max_edge_dist = 0, or every node masked): zero-size tensors have a null data_ptr the C ABI would
            # reject as a missing argument, so count with a one-element scratch and hand back empty tensors
            dummy = torch.empty((2,), dtype=torch.int64, device=self.device)
            m = run(dummy, dummy, 0)
            return (torch.empty((2, 0), dtype=torch.int64 if index64 else torch.int32, device=self.device),
                    torch.empty((0,), dtype=torch.float32, device=self.device), m)
        ei = torch.empty((2, cap), dtype=torch.int64 if index64 else torch.int32, device=self.device)
        ea = torch.empty((cap,), dtype=torch.float32, device=self.device)
        m = run(ei, ea, cap)
        return ei[:, :min(m, cap)], ea[:min(m, cap)], m

    def expand_node_obs(self, table, out=None, out_envs=None, env_offset=0):
        """node_obs rows from entity tables, bit-identical to what the engine writes (gmpe_expand_node_obs; module-level `expand_node_obs` needs no engine)."""
        return expand_node_obs(self.cfg, table, out=out, out_envs=out_envs, env_offset=env_offset)

    def expand_adj(self, table, copies=1, out=None, out_envs=None, env_offset=0):
        """The adjacency of every env-step from its entity table, bit-identical to the engine's own output (gmpe_expand_adj)."""
        return expand_adj(self.cfg, table, copies=copies, out=out, out_envs=out_envs, env_offset=env_offset)

    def masks_from_dones(self, done, masks, active_masks):
        """masks / active_masks (f32 [N,A,...], contiguous, N*A elements) from a uint8 [N,A] done tensor, one tiny kernel."""
        _lib.check(self.lib.gmpe_masks_from_dones(self.h, done.data_ptr(), masks.data_ptr() if masks is not None else None,
                                                  active_masks.data_ptr() if active_masks is not None else None, self._stream()),
                   "gmpe_masks_from_dones")

    # ------------------------------------------------------------------ timing hooks
    def timing(self, enable):
        _lib.check(self.lib.gmpe_timing_enable(self.h, int(bool(enable))), "gmpe_timing_enable")

    def timing_read(self, reset=True):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self.lib.gmpe_timing_read(self.h, C.byref(ms), C.byref(n), int(reset)), "gmpe_timing_read")
        return ms.value, n.value

    def region_mark(self, which):
        """HIP event on the launch stream: which=0 before the first launch of a region, 1 after the last."""
        _lib.check(self.lib.gmpe_timing_mark(self.h, int(which), self._stream()), "gmpe_timing_mark")

    def region_ms(self):
        ms = C.c_double()
        _lib.check(self.lib.gmpe_timing_region_ms(self.h, C.byref(ms)), "gmpe_timing_region_ms")
        return ms.value

    @property
    def bytes_per_env_step(self):
        return algorithmic_bytes_per_env_step(self.cfg)


def expand_node_obs(cfg, table, out=None, out_envs=None, env_offset=0):
    """Learner side of the compact gather: float64 entity tables [..., n, W] (device tensor, contiguous; leading dims = blocks such as the T+1 slots of a rollout) ->
    node_obs float32 [..., out_envs, A, E, F], rows of the table's n envs written at env_offset .. env_offset + n of every block (default: out_envs = n). The rows are
    bit-identical to the engine's own node_obs (same arithmetic, gmpe_step.hip k_node_expand). No handle needed: the learner rank may own no envs."""
    lib = _lib.load()
    if table.dtype != torch.float64 or not table.is_contiguous() or not table.is_cuda or table.dim() < 2 or table.shape[-1] != cfg.entity_table_width:
        raise ValueError("table must be a contiguous float64 device tensor [..., n, %d]" % cfg.entity_table_width)
    n = int(table.shape[-2])
    blocks = int(table.numel() // (n * table.shape[-1])) if n else 0
    out_envs = n if out_envs is None else int(out_envs)
    A, E, F = cfg.num_agents, cfg.num_entities, cfg.node_feats
    shape = tuple(table.shape[:-2]) + (out_envs, A, E, F)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=table.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != table.device:
        raise ValueError("out must be a contiguous float32 tensor of shape %s on %s" % (shape, table.device))
    if blocks and n:
        _lib.check(lib.gmpe_expand_node_obs(C.byref(cfg), table.device.index, table.data_ptr(), blocks, n, out.data_ptr(), out_envs, int(env_offset),
                                            C.c_void_p(torch.cuda.current_stream(table.device).cuda_stream)), "gmpe_expand_node_obs")
    return out


def expand_adj(cfg, table, copies=1, out=None, out_envs=None, env_offset=0):
    """float64 entity tables [..., n, W] -> adjacency float32 [..., out_envs, E, E] (copies = 1) or [..., out_envs, copies, E, E] (copies = A: the materialised
    per-agent form), rows of the table's n envs written at env_offset .. env_offset + n of every block. f32(sqrt(dx^2 + dy^2)) with the engine's own expression and
    this step's mask words: bit-identical to the adjacency the engine writes (gmpe_step.hip k_adj_from_table). No handle needed."""
    lib = _lib.load()
    if table.dtype != torch.float64 or not table.is_contiguous() or not table.is_cuda or table.dim() < 2 or table.shape[-1] != cfg.entity_table_width:
        raise ValueError("table must be a contiguous float64 device tensor [..., n, %d]" % cfg.entity_table_width)
    n = int(table.shape[-2])
    blocks = int(table.numel() // (n * table.shape[-1])) if n else 0
    out_envs = n if out_envs is None else int(out_envs)
    E = cfg.num_entities
    shape = tuple(table.shape[:-2]) + ((out_envs, E, E) if copies == 1 else (out_envs, int(copies), E, E))
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=table.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != table.device:
        raise ValueError("out must be a contiguous float32 tensor of shape %s on %s" % (shape, table.device))
    if blocks and n:
        _lib.check(lib.gmpe_expand_adj(C.byref(cfg), table.device.index, table.data_ptr(), blocks, n, out.data_ptr(), out_envs, int(env_offset), int(copies),
                                       C.c_void_p(torch.cuda.current_stream(table.device).cuda_stream)), "gmpe_expand_adj")
    return out
