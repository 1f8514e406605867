"""ctypes binding of libgmpe.so (include/gmpe.h). No CPU fallback: a missing library raises."""
import ctypes as C
import os

from .config import GmpeConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# GMPE_LIB: diagnostic override to A/B another build of the SAME library (tools/*.py); never a fallback. bench.py refuses to report
# a number with it set (unless --diag, which records it in the JSON line).
LIB_PATH = os.environ.get("GMPE_LIB") or os.path.join(_HERE, "libgmpe.so")
_lib = None

SYMBOLS = ["gmpe_abi_version", "gmpe_last_error", "gmpe_obs_dim", "gmpe_node_feats", "gmpe_num_entities", "gmpe_create",
           "gmpe_destroy", "gmpe_set_rng_tape", "gmpe_reset", "gmpe_step", "gmpe_step_many", "gmpe_step_many_prepare", "gmpe_step_onehot",
           "gmpe_field_bytes", "gmpe_get_field", "gmpe_set_field", "gmpe_edges_from_adj", "gmpe_masks_from_dones",
           "gmpe_timing_enable", "gmpe_timing_read", "gmpe_timing_mark", "gmpe_timing_region_ms",
           "gmpe_rollout_steps", "gmpe_get_tuning", "gmpe_step_many_launches", "gmpe_edges_from_adj_compact",
           "gmpe_set_control_override", "gmpe_field_device_ptr", "gmpe_step_envs", "gmpe_step_many_envs",
           "gmpe_entity_table_width", "gmpe_expand_node_obs", "gmpe_expand_adj"]


class GmpeOutputs(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("agent_id", C.c_void_p), ("node_obs", C.c_void_p),
                ("adj", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("info", C.c_void_p), ("adj_compact", C.c_int32), ("reserved", C.c_int32), ("entity_table", C.c_void_p)]


class GmpeRollout(C.Structure):
    """gmpe_rollout (include/gmpe.h): K steps in one launch, step k -> output slot (first_slot + k) % num_slots."""
    _fields_ = [("num_steps", C.c_int32), ("num_action_sets", C.c_int32), ("num_slots", C.c_int32), ("first_slot", C.c_int32),
                ("stride_obs", C.c_int64), ("stride_agent_id", C.c_int64), ("stride_node_obs", C.c_int64), ("stride_adj", C.c_int64),
                ("stride_reward", C.c_int64), ("stride_done", C.c_int64), ("stride_info", C.c_int64), ("stride_masks", C.c_int64),
                ("masks", C.c_void_p), ("active_masks", C.c_void_p), ("stride_entity_table", C.c_int64)]


class GmpeTuning(C.Structure):
    _fields_ = [("G", C.c_int32), ("block", C.c_int32), ("nt", C.c_int32), ("spec", C.c_int32), ("split", C.c_int32),
                ("roll", C.c_int32), ("ap", C.c_int32), ("lds_bytes", C.c_int32), ("diag_build", C.c_int32),
                ("G_roll", C.c_int32), ("block_roll", C.c_int32), ("chunks", C.c_int32), ("ahead", C.c_int32), ("xstep", C.c_int32), ("chunks_x", C.c_int32), ("ahead_x", C.c_int32),
                ("lds_bytes_roll", C.c_int32)]


class GmpeError(RuntimeError):
    pass


def load():
    """Load libgmpe.so. torch must be imported first so that the engine binds to the SAME HIP
    runtime (libamdhip64.so.7) torch uses; two runtimes in one process cannot share device memory."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GmpeError("libgmpe.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C contracts-marl-aam-corridors_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64 first)
    lib = C.CDLL(LIB_PATH)
    P, I = C.c_void_p, C.c_int
    lib.gmpe_last_error.restype = C.c_char_p
    lib.gmpe_obs_dim.argtypes = [C.POINTER(GmpeConfig)]
    lib.gmpe_node_feats.argtypes = [C.POINTER(GmpeConfig)]
    lib.gmpe_num_entities.argtypes = [C.POINTER(GmpeConfig)]
    lib.gmpe_create.argtypes = [C.POINTER(GmpeConfig), I, C.POINTER(P)]
    lib.gmpe_destroy.argtypes = [P]
    lib.gmpe_set_rng_tape.argtypes = [P, P, C.c_int64]
    lib.gmpe_reset.argtypes = [P, P, C.POINTER(GmpeOutputs), P]
    lib.gmpe_step.argtypes = [P, P, C.POINTER(GmpeOutputs), P]
    lib.gmpe_step_envs.argtypes = [P, P, C.POINTER(GmpeOutputs), C.c_int32, C.c_int32, P]
    lib.gmpe_step_many_envs.argtypes = [P, P, C.c_int32, C.c_int32, C.POINTER(GmpeOutputs), C.c_int32, P]
    lib.gmpe_step_onehot.argtypes = [P, P, C.POINTER(GmpeOutputs), P]
    lib.gmpe_step_many.argtypes = [P, P, C.c_int32, C.c_int32, C.POINTER(GmpeOutputs), P]
    lib.gmpe_step_many_launches.argtypes = [P, P, C.c_int32, C.c_int32, C.POINTER(GmpeOutputs), P]
    lib.gmpe_step_many_prepare.argtypes = [P, P, C.c_int32, C.c_int32, C.POINTER(GmpeOutputs)]
    lib.gmpe_field_bytes.argtypes = [P, I, C.POINTER(C.c_size_t)]
    lib.gmpe_get_field.argtypes = [P, I, P, C.c_size_t]
    lib.gmpe_set_field.argtypes = [P, I, P, C.c_size_t]
    lib.gmpe_edges_from_adj.argtypes = [P, P, C.c_int32, C.c_int32, C.c_float, C.c_int32, P, P, C.c_int32, P, P]
    lib.gmpe_edges_from_adj_compact.argtypes = [P, P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, P, P, C.c_int64, P, P]
    lib.gmpe_masks_from_dones.argtypes = [P, P, P, P, P]
    lib.gmpe_set_control_override.argtypes = [P, P, P]
    lib.gmpe_field_device_ptr.argtypes = [P, I, C.POINTER(P)]
    lib.gmpe_timing_enable.argtypes = [P, C.c_int32]
    lib.gmpe_timing_read.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]
    lib.gmpe_timing_mark.argtypes = [P, C.c_int32, P]
    lib.gmpe_timing_region_ms.argtypes = [P, C.POINTER(C.c_double)]
    lib.gmpe_rollout_steps.argtypes = [P, P, C.POINTER(GmpeRollout), C.POINTER(GmpeOutputs), P]
    lib.gmpe_get_tuning.argtypes = [P, C.POINTER(GmpeTuning)]
    lib.gmpe_entity_table_width.argtypes = [C.POINTER(GmpeConfig)]
    lib.gmpe_expand_node_obs.argtypes = [C.POINTER(GmpeConfig), I, P, C.c_int64, C.c_int64, P, C.c_int64, C.c_int64, P]
    lib.gmpe_expand_adj.argtypes = [C.POINTER(GmpeConfig), I, P, C.c_int64, C.c_int64, P, C.c_int64, C.c_int64, C.c_int32, P]
    from .config import ABI_VERSION
    if lib.gmpe_abi_version() != ABI_VERSION:
        raise GmpeError("libgmpe.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise GmpeError("%s failed (%d): %s" % (what, rc, load().gmpe_last_error().decode()))
