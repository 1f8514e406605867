"""BatchedGraphMPEVecEnv — same surface as GraphSubprocVecEnv, backed by the HIP engine.

Mirrors onpolicy/envs/env_wrappers.py:959-1037 (class GraphSubprocVecEnv) and its base ShareVecEnv
(:28-141): `num_envs`, the eight space lists, `reset(num_current_episode)`,
`step_async/step_wait/step(actions, num_current_episode)`, `close()`, `render()`.
Where the reference forks N processes that each run MultiAgentGraphEnv.step
(multiagent/environment.py:1021-1063) and ship pickled fp64 arrays through pipes, this class makes
ONE kernel launch per step for all N envs and hands the runner NumPy views in the dtypes its
buffer stores (float32 / int32 / bool; onpolicy/utils/graph_buffer.py:84-114).
"""
import numpy as np
import torch

from . import _lib
from .config import INFO_KEYS, NODE_FEATS, ROT_FAMILY, config_from_args
from .engine import GmpeEngine, StepOutputs
from .spaces import Box, Discrete


class LazyInfos(object):
    """Sequence of N per-env info lists, materialised from the device counters only when read.

    The runner touches infos only every `log_interval` episodes (graph_mpe_runner.py:166-189,
    base_runner.py:194-290: `for info in infos: info[agent_id][key]`); building 4096x10 dicts of 17
    keys every step would dominate the host, so the dicts are created on first access.
    """

    def __init__(self, info_dev, n_envs, n_agents, include_min_time=True, include_phase=False, owner=None, generation=0):
        """info_dev: the device tensor the step wrote its info rows into. The vec env alternates TWO such tensors (no per-step clone), so
        the rows stay valid until the step after next; `owner` / `generation` let a late read fail loudly instead of returning a later
        step's rows."""
        self._dev = info_dev
        self._host = None
        self._n, self._a = n_envs, n_agents
        self._owner, self._gen = owner, generation
        # (key, column) pairs: 'Min_time_to_goal' only with max_speed (…_july.py:826-828), 'Phase_reached' only in rot_inv (:835)
        self._keys = [(k, j) for j, k in enumerate(INFO_KEYS)
                      if (k != "Min_time_to_goal" or include_min_time) and (k != "Phase_reached" or include_phase)]

    def _fetch(self):
        if self._host is None:
            if self._owner is not None and self._owner._info_gen - self._gen >= 2:
                raise RuntimeError("infos of step %d read after step %d overwrote their device buffer: read infos (or call .as_array()) "
                                   "before the step after next" % (self._gen, self._owner._info_gen))
            self._host = self._dev.detach().cpu().numpy().astype(np.float64)
            self._dev = None
        return self._host

    def __len__(self):
        return self._n

    def __getitem__(self, e):
        if isinstance(e, slice):
            return [self[i] for i in range(*e.indices(self._n))]
        if e < 0:
            e += self._n
        if not 0 <= e < self._n:
            raise IndexError(e)
        h = self._fetch()
        return [{k: float(h[e, a, j]) for k, j in self._keys} for a in range(self._a)]

    def __iter__(self):
        for e in range(self._n):
            yield self[e]

    def as_array(self):
        """[N, A, 18] float64 in config.INFO_KEYS order (no dict construction)."""
        return self._fetch()


class BatchedGraphMPEVecEnv(object):
    """Drop-in for GraphSubprocVecEnv([get_env_fn(i) for i in range(n_rollout_threads)])."""
    closed = False
    viewer = None
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, all_args, num_envs=None, device=0, env_id_base=0, adj_broadcast_view=True, pinned_host=True, safety_filter=None,
                 eval_surface=False):
        """eval_surface: reproduce GraphDummyVecEnv instead (env_wrappers.py:903-956) — what train_mpe.py:36, 61 / eval_mpe.py:36 pick for
        n_rollout_threads == 1 and GMPERunner.render unpacks (graph_mpe_runner.py:621-622): `step` returns an 8-tuple whose last element is
        `reset_count` (1 if an env of the batch was auto-reset in this step, else 0; env_wrappers.py:923-936).
        safety_filter: the hook slot of `World.step`'s safety filter (multiagent/core.py:692-736). A callable
        `f(engine, actions_dev) -> (ctrl [N,A,2] float64 device tensor, use [N,A] uint8 device tensor or None)` called before every
        step; where `use` is set the engine integrates `ctrl` instead of the decoded action. The HJ / CBF filter of the reference
        (safety_filter.py: jax / cvxpy / value-function data) is not built, so `args.use_safety_filter` without a callable raises."""
        if getattr(all_args, "use_safety_filter", False) and safety_filter is None:
            raise NotImplementedError("use_safety_filter=True: the HJ/CBF filter is out of scope (DESIGN.md); pass safety_filter=callable "
                                      "to fill the hook slot")
        self._safety_filter = safety_filter
        self._eval_surface = bool(eval_surface)
        self.cfg = config_from_args(all_args, num_envs=num_envs, env_id_base=env_id_base)
        # The reference's per-agent adj arrays alias ONE E x E matrix per env, so the
        # engine writes that matrix once and the [N,A,E,E] result is a zero-copy broadcast view.
        self._compact = bool(adj_broadcast_view)
        self.engine = GmpeEngine(self.cfg, device=device, adj_compact=self._compact, with_info=True)
        c = self.cfg
        self.num_envs = c.num_envs
        self.num_agents = self.n = c.num_agents
        A, E, D = c.num_agents, c.num_entities, c.obs_dim
        f32 = np.float32
        # spaces: multiagent/environment.py:152-208 (obs/share_obs/action) and :986-1018 (graph)
        self.observation_space = [Box(-np.inf, np.inf, (D,), f32) for _ in range(A)]
        self.share_observation_space = [Box(-np.inf, np.inf, (A * D,), f32) for _ in range(A)]
        self.action_space = [Discrete(c.n_actions) for _ in range(A)]
        self.node_observation_space = [Box(-np.inf, np.inf, (E, self.cfg.node_feats), f32) for _ in range(A)]
        self.adj_observation_space = [Box(-np.inf, np.inf, (E, E), f32) for _ in range(A)]
        self.edge_observation_space = [Box(-np.inf, np.inf, (1,), f32) for _ in range(A)]
        self.agent_id_observation_space = [Box(-np.inf, np.inf, (1,), f32) for _ in range(A)]
        self.share_agent_id_observation_space = [Box(-np.inf, np.inf, (A * 1,), f32) for _ in range(A)]
        self.waiting = False
        self._pending = None
        self._errors_reported = False
        # info rows: two device buffers bound alternately (a LazyInfos keeps a reference to its step's buffer instead of a per-step clone);
        # sticky device error flags (tape exhausted / placement gave up) travel down with every hand-off and raise GmpeError here
        o = self.engine.out
        self._outs = [o, StepOutputs(**{k: (torch.empty_like(o.info) if k == "info" else getattr(o, k)) for k in StepOutputs.__slots__})]
        self._info_gen = 0
        self._err_dev = self.engine.state_tensor("error_flags")
        # Host hand-off for the NumPy runner: two alternating sets of pinned staging buffers, filled by asynchronous
        # D2H copies on the engine's stream (arrays returned by step t stay valid until step t+2; the runner copies
        # them into its replay buffer immediately, graph_buffer.py:223-236). pinned_host=False returns fresh arrays.
        self._pinned = bool(pinned_host)
        self._host = None
        self._flip = 0
        if self._pinned:
            o = self.engine.out
            mk = lambda t: torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
            self._host = [dict(obs=mk(o.obs), agent_id=mk(o.agent_id), node_obs=mk(o.node_obs), adj=mk(o.adj),
                               reward=mk(o.reward), done=mk(o.done), err=mk(self._err_dev)) for _ in range(2)]
        # Actions go up through pinned staging too: whatever dtype the runner hands over (np.eye(n)[a] is float64, indices are int64) is
        # converted straight INTO the pinned buffer (single-threaded NumPy, see _upload) and uploaded from there.
        dev = self.engine.device
        self._act_host = dict(onehot=torch.empty((c.num_envs, A, c.n_actions), dtype=torch.float32, pin_memory=self._pinned),
                              index=torch.empty((c.num_envs, A), dtype=torch.int32, pin_memory=self._pinned))
        self._act_dev = dict(onehot=torch.empty((c.num_envs, A, c.n_actions), dtype=torch.float32, device=dev),
                             index=torch.empty((c.num_envs, A), dtype=torch.int32, device=dev))

    # ------------------------------------------------------------------ helpers
    def _expand_adj(self, a):
        if self._compact:
            N, E = a.shape[0], a.shape[-1]
            a = np.broadcast_to(a[:, None], (N, self.num_agents, E, E))
        return a

    def _fetch(self, o, with_step_outputs):
        """Device outputs -> NumPy. Returns (obs, agent_id, node_obs, adj[, reward, done])."""
        keys = ("obs", "agent_id", "node_obs", "adj") + (("reward", "done") if with_step_outputs else ())
        if self._pinned:
            h = self._host[self._flip]
            self._flip ^= 1
            for k in keys:
                h[k].copy_(getattr(o, k), non_blocking=True)
            h["err"].copy_(self._err_dev, non_blocking=True)
            torch.cuda.current_stream(self.engine.device).synchronize()
            arrs = [h[k].numpy() for k in keys]
            err = h["err"].numpy()
        else:
            arrs = [getattr(o, k).detach().cpu().numpy() for k in keys]
            err = self._err_dev.cpu().numpy()
        if err.any():
            self._raise_errors()
        arrs[3] = self._expand_adj(arrs[3])
        return arrs

    def _raise_errors(self):
        """Sticky per-env error flags of the engine (include/gmpe.h error_flags) -> GmpeError. The reference has no counterpart: its
        rejection sampler spins forever in a world too small for its agents (…_july.py:462-486), the engine's is bounded."""
        self._errors_reported = True                  # close() will not raise the same sticky flags again
        self.engine.check_errors()

    # ------------------------------------------------------------------ GraphSubprocVecEnv surface
    def reset(self, num_current_episode=0):
        """-> (obs [N,A,D], agent_id [N,A,1], node_obs [N,A,E,F], adj [N,A,E,E])  (env_wrappers.py:1006-1013)"""
        o = self.engine.reset()
        return tuple(self._fetch(o, False))

    def step_async(self, actions, num_current_episode=None):
        """actions: [N, A, n_actions] one-hot (graph_mpe_runner.py:375-377), or [N, A] integer indices."""
        if self.waiting:
            raise RuntimeError("step_async called while a step is pending")
        if torch.is_tensor(actions):
            a = actions
        else:
            a = np.asarray(actions)
        if a.ndim == 3:
            if tuple(a.shape) != (self.num_envs, self.num_agents, self.cfg.n_actions):
                raise ValueError("actions must be [N=%d, A=%d, %d]" % (self.num_envs, self.num_agents, self.cfg.n_actions))
            t = self._upload(a, "onehot")
            self._filter(t)
            self._next_info()
            self._pending = self.engine.step_onehot(t)
        elif a.ndim == 2:
            if tuple(a.shape) != (self.num_envs, self.num_agents):
                raise ValueError("actions must be [N=%d, A=%d]" % (self.num_envs, self.num_agents))
            t = self._upload(a, "index")
            self._filter(t)
            self._next_info()
            self._pending = self.engine.step(t)
        else:
            raise ValueError("actions must be a one-hot [N,A,n_act] or an index [N,A] array")
        self.waiting = True

    def _next_info(self):
        self._info_gen += 1
        self.engine.rebind(self._outs[self._info_gen & 1])

    def _upload(self, a, kind):
        """Host (NumPy / CPU tensor, any numeric dtype) or device actions -> the engine's device tensor of the right dtype."""
        dst = self._act_dev[kind]
        if torch.is_tensor(a) and a.is_cuda:
            return a if (a.dtype == dst.dtype and a.is_contiguous() and a.device == dst.device) else dst.copy_(a)
        stage = self._act_host[kind]
        # dtype conversion straight into pinned memory — with NumPy on purpose: a torch CPU op of this size fans out over every host
        # core (128 OpenMP threads on the MI355X boxes), and under a CPU quota their spin-waiting gets the whole process throttled for the
        # rest of the scheduler period (~90 ms stalls once per ~30 steps, tools/hostpath3.py)
        np.copyto(stage.numpy(), a.numpy() if torch.is_tensor(a) else a, casting="unsafe")
        return dst.copy_(stage, non_blocking=True)

    def _filter(self, actions_dev):
        if self._safety_filter is not None:
            ctrl, use = self._safety_filter(self.engine, actions_dev)
            self.engine.set_control_override(ctrl, use)

    def step_wait(self):
        """-> 7-tuple (obs, agent_id, node_obs, adj, rewards [N,A], dones [N,A] bool, infos)
        (env_wrappers.py:996-1004). Envs whose agents were all done carry POST-reset observations with
        the terminal reward/done (graphworker, env_wrappers.py:865-873)."""
        if not self.waiting:
            raise RuntimeError("step_wait without step_async")
        o = self._pending
        self._pending, self.waiting = None, False
        obs, ids, node, adj, rew, done = self._fetch(o, True)    # synchronises the stream
        done = done.astype(bool)
        if self.cfg.collaborative:
            rew = rew[..., None]                       # `reward_n = [[reward]] * self.n` (environment.py:1056-1061) stacks to [N, A, 1]
        # 'Phase_reached' is an info key of the rot_inv family only (rot_inv.py:835); the July file's info_callback has 17 keys (…_july.py:806-828)
        infos = LazyInfos(o.info, self.num_envs, self.num_agents, include_min_time=self.cfg.max_speed > 0,
                          include_phase=self.cfg.scenario in ROT_FAMILY, owner=self, generation=self._info_gen)
        if self._eval_surface:
            reset_count = 1 if done.all(axis=1).any() else 0       # GraphDummyVecEnv.step_wait (env_wrappers.py:923-936)
            return obs, ids, node, adj, rew, done, infos, reset_count
        return obs, ids, node, adj, rew, done, infos

    def step(self, actions, num_current_episode=None):
        self.step_async(actions, num_current_episode)
        return self.step_wait()

    def reset_task(self):
        raise NotImplementedError("reset_task is not part of the GraphMPE path")

    def render(self, mode="rgb_array"):
        raise NotImplementedError("rendering (pyglet viewer) is out of scope of the step engine")

    def close(self):
        """Frees the handle unconditionally. Sticky device errors that no reset / step hand-off has raised yet are raised here — once: a close() in a
        `finally` / `except` clean-up path after a GmpeError must not replace the exception that is already propagating (it warns instead)."""
        if self.closed:
            return
        try:
            if not self._errors_reported:
                self.engine.check_errors()            # last chance to report sticky device errors (synchronises)
        except _lib.GmpeError:
            self._errors_reported = True
            raise
        except Exception as e:                        # a handle that is already in a failed state: do not mask the original failure
            import warnings
            warnings.warn("BatchedGraphMPEVecEnv.close: could not read the device error flags (%s)" % e)
        finally:
            self.engine.close()
            self.closed = True

    @property
    def unwrapped(self):
        return self

    # ------------------------------------------------------------------ extras (not in the reference)
    def step_device(self, action_idx):
        """Zero-copy variant: int32 device tensor in, device tensors out (engine.StepOutputs)."""
        return self.engine.step(action_idx)

    def seed(self, seed=None):
        raise NotImplementedError("seeds are part of the config (args.seed): per-env streams are keyed by "
                                  "(seed, env id), the counterpart of env.seed(seed + rank*1000) in train_mpe.py:31")


def make_train_env(all_args, device=0, eval_surface=False):
    """Counterpart of onpolicy/scripts/train_mpe.py:21-43 for env_name == 'GraphMPE'. `step` returns the GraphSubprocVecEnv 7-tuple for EVERY thread
    count, one included: that is what the collect / eval loops unpack (graph_mpe_runner.py:83, 490). The reference itself builds GraphDummyVecEnv
    for n_rollout_threads == 1 (train_mpe.py:34-36), whose 8-tuple (env_wrappers.py:920-936) those loops cannot unpack — a single-thread training
    run of the reference raises "too many values to unpack"; only GMPERunner.render wants the 8-tuple (graph_mpe_runner.py:621-622). Pass
    eval_surface=True (or use make_eval_env) to get that shape."""
    if getattr(all_args, "env_name", "GraphMPE") != "GraphMPE":
        raise NotImplementedError("only the GraphMPE route is built")
    return BatchedGraphMPEVecEnv(all_args, num_envs=all_args.n_rollout_threads, device=device, eval_surface=eval_surface)


def GraphMPEEnv(args, device=0):
    """multiagent/MPE_env.py:56-84 builds ONE env; here that is a batch of one."""
    assert "graph" in args.scenario_name, "Only use graph env for graph scenarios"
    return BatchedGraphMPEVecEnv(args, num_envs=1, device=device)


def make_eval_env(all_args, device=0):
    """Counterpart of onpolicy/scripts/train_mpe.py:46-68 / eval_mpe.py:21-43: n_eval_rollout_threads == 1 selects GraphDummyVecEnv,
    whose step returns the 8-tuple GMPERunner.render unpacks (graph_mpe_runner.py:621-622)."""
    if getattr(all_args, "env_name", "GraphMPE") != "GraphMPE":
        raise NotImplementedError("only the GraphMPE route is built")
    n = getattr(all_args, "n_eval_rollout_threads", 1)
    return BatchedGraphMPEVecEnv(all_args, num_envs=n, device=device, eval_surface=(n == 1))
