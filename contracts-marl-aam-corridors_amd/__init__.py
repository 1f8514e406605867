"""gmpe — MI355X-native batched GraphMPE step engine (package dir: contracts-marl-aam-corridors_amd/).

Host side is Python (the reference is Python): `vec_env.BatchedGraphMPEVecEnv` keeps the surface of
`GraphSubprocVecEnv` (onpolicy/envs/env_wrappers.py:959-1037) and drives hand-written HIP kernels
for gfx950 through the C ABI of include/gmpe.h (ctypes). There is no CPU fallback: without the
built libgmpe.so the engine raises.
"""
from . import config  # noqa: F401
from .config import make_config, config_from_args  # noqa: F401

__all__ = ["config", "make_config", "config_from_args"]


def __getattr__(name):
    # torch / HIP-dependent modules are imported on first use
    if name in ("engine", "vec_env", "spaces", "_lib", "sharding", "rollout"):
        import importlib
        return importlib.import_module("." + name, __name__)
    if name in ("BatchedGraphMPEVecEnv", "GraphMPEEnv", "make_train_env", "make_eval_env"):
        from . import vec_env
        return getattr(vec_env, name)
    if name == "GmpeEngine":
        from . import engine
        return engine.GmpeEngine
    raise AttributeError(name)
