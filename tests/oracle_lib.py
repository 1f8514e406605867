"""ctypes binding of oracle/gmpe_oracle.c — TEST INFRASTRUCTURE (checker / CPU baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

import gmpe
from gmpe.config import FIELDS, INFO_KEYS, NODE_FEATS, GmpeConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "_build", "libgmpe_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, "gmpe_oracle.c"), os.path.join(ROOT, "include", "gmpe.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(q) for q in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return SO


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(build())
    P = C.c_void_p
    lib.gmpo_create.argtypes = [C.POINTER(GmpeConfig), C.POINTER(P)]
    lib.gmpo_destroy.argtypes = [P]
    lib.gmpo_get_field.argtypes = [P, C.c_int, P, C.c_size_t]
    lib.gmpo_set_field.argtypes = [P, C.c_int, P, C.c_size_t]
    lib.gmpo_set_rng_tape.argtypes = [P, P, C.c_int64]
    lib.gmpo_set_control_override.argtypes = [P, P, P]
    lib.gmpo_get_dist_cache.argtypes = [P, P]
    lib.gmpo_get_entity_table.argtypes = [P, P]
    lib.gmpo_entity_table_width.argtypes = [C.POINTER(GmpeConfig)]
    lib.gmpo_reset.argtypes = [P, P, P, P, P, P]
    lib.gmpo_step.argtypes = [P, P, P, P, P, P, P, P, P, P, C.c_int]
    lib.gmpo_update_graph.argtypes = [P, C.c_int, P, P, C.c_int]
    lib.gmpo_last_error.restype = C.c_char_p
    lib.gmpo_philox_uniform.restype = C.c_double
    lib.gmpo_philox_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    lib.gmpo_kinematic_step.argtypes = [P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, P, P]
    lib.gmpo_force_step.argtypes = [C.c_int, C.c_int, P, P, P, P, P, P, C.c_double, P, P, P, C.c_int, P,
                                    C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, P, P]
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle(object):
    """Sequential CPU restatement; outputs float64. adj is the single [N,E,E] matrix per env."""

    def __init__(self, cfg):
        self.lib = load()
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.lib.gmpo_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise RuntimeError("gmpo_create: %s" % self.lib.gmpo_last_error().decode())
        self.N, self.A, self.E, self.D = cfg.num_envs, cfg.num_agents, cfg.num_entities, cfg.obs_dim
        self._tape = None

    def close(self):
        if self.h:
            self.lib.gmpo_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def get(self, name):
        fid, dt, shp = FIELDS[name]
        a = np.empty(shp(self.cfg), dtype=dt)
        rc = self.lib.gmpo_get_field(self.h, fid, _p(a), a.nbytes)
        assert rc == 0, self.lib.gmpo_last_error()
        return a

    def set(self, name, value):
        fid, dt, shp = FIELDS[name]
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=dt), shp(self.cfg)))
        rc = self.lib.gmpo_set_field(self.h, fid, _p(a), a.nbytes)
        assert rc == 0, self.lib.gmpo_last_error()

    def set_tape(self, tape):
        if tape is None:
            self._tape = None
            self.lib.gmpo_set_rng_tape(self.h, None, 0)
            return
        t = np.ascontiguousarray(np.asarray(tape, dtype=np.float64).reshape(self.N, -1))
        self._tape = t
        self.lib.gmpo_set_rng_tape(self.h, _p(t), t.shape[1])

    def set_control_override(self, ctrl=None, use=None):
        """World.step's safety-filter slot: float64 [N,A,2] controls integrated where use [N,A] is non-zero (None: everywhere)."""
        self._ovr = None if ctrl is None else (np.ascontiguousarray(ctrl, dtype=np.float64), None if use is None else np.ascontiguousarray(use, dtype=np.uint8))
        if self._ovr is None:
            self.lib.gmpo_set_control_override(self.h, None, None)
        else:
            self.lib.gmpo_set_control_override(self.h, _p(self._ovr[0]), _p(self._ovr[1]))

    def _bufs(self):
        N, A, E, D = self.N, self.A, self.E, self.D
        return (np.zeros((N, A, D)), np.zeros((N, A, 1), np.int32), np.zeros((N, A, E, self.cfg.node_feats)),
                np.zeros((N, E, E)))

    def reset(self, mask=None):
        obs, ids, node, adj = self._bufs()
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        self.lib.gmpo_reset(self.h, _p(m), _p(obs), _p(ids), _p(node), _p(adj))
        return obs, ids, node, adj

    def step(self, act, auto_reset=True):
        N, A = self.N, self.A
        obs, ids, node, adj = self._bufs()
        rew = np.zeros((N, A)); done = np.zeros((N, A), np.uint8)
        info = np.zeros((N, A, len(INFO_KEYS))); did = np.zeros(N, np.uint8)
        a = np.ascontiguousarray(np.asarray(act, dtype=np.int32).reshape(N, A))
        self.lib.gmpo_step(self.h, _p(a), _p(obs), _p(ids), _p(node), _p(adj), _p(rew), _p(done), _p(info),
                           _p(did), int(auto_reset))
        return obs, ids, node, adj, rew, done.astype(bool), info, did.astype(bool)

    def entity_table(self):
        """[N, W] table of the last step / reset (what a rank ships instead of node_obs; include/gmpe.h gmpe_outputs.entity_table)."""
        t = np.zeros((self.N, self.cfg.entity_table_width))
        self.lib.gmpo_get_entity_table(self.h, _p(t))
        return t

    def dist_cache(self):
        d = np.zeros((self.N, self.E, self.E))
        self.lib.gmpo_get_dist_cache(self.h, _p(d))
        return d

    def update_graph(self, n=0):
        cap = self.E * self.E
        e = np.zeros((2, cap), np.int32); w = np.zeros(cap)
        m = self.lib.gmpo_update_graph(self.h, n, _p(e), _p(w), cap)
        return e[:, :m], w[:m]
