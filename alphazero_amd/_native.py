"""ctypes binding of libaz_mcts.so (the C ABI declared in include/az_mcts.h).

The library is the only compute path of this package: if it is missing or fails
to load, importing this module raises -- there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AZ_MCTS_LIB selects another build of the same library (diagnostic builds under tools/probes)
LIB_PATH = os.environ.get("AZ_MCTS_LIB") or os.path.join(_HERE, "libaz_mcts.so")

MAX_LAYERS = 8
MAX_ACTIONS = 64
MAX_ATOMS = 64
MAX_WIDTH = 1024

VARIANT_MUZERO, VARIANT_SAMPLED, VARIANT_GUMBEL = 0, 1, 2
QT_PARENT_SIBLINGS, QT_COMPLETED_MIX, QT_MIN_MAX = 0, 1, 2
ENV_CARTPOLE, ENV_MOUNTAINCAR, ENV_ACROBOT, ENV_ACROBOT_NOTERM = 1, 2, 3, 4
NOISE_THREEFRY, NOISE_INJECTED, NOISE_ZERO = 0, 1, 2
VARIANT_STOCHASTIC = 3


class NativeLibraryError(RuntimeError):
    pass


class SearchDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("A", C.c_int32), ("K", C.c_int32),
        ("variant", C.c_int32), ("max_depth", C.c_int32), ("qtransform", C.c_int32),
        ("pb_c_init", C.c_float), ("pb_c_base", C.c_float), ("q_epsilon", C.c_float),
        ("value_scale", C.c_float), ("maxvisit_init", C.c_float),
        ("rescale_values", C.c_int32), ("use_mixed_value", C.c_int32),
        ("min_value", C.c_float), ("max_value", C.c_float),
        ("max_num_considered", C.c_int32), ("noise_mode", C.c_int32),
        ("noise_depth", C.c_int32), ("key", C.c_uint32 * 2),
        ("kernel_form", C.c_int32), ("num_decision_actions", C.c_int32),
    ]


TREE_ARRAYS = ("node_visits", "raw_values", "node_values", "parents", "action_from_parent",
               "children_index", "children_prior_logits", "children_visits", "children_rewards",
               "children_discounts", "children_values")


class TreePtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TREE_ARRAYS + ("emb_state", "emb_time")]


class SearchInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("invalid_actions", "root_gumbel", "table", "noise", "is_decision", "sim_keys")]


class Injected(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("prior_logits", "value", "reward", "discount")]


class MlpParams(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("sizes", C.c_int32 * (MAX_LAYERS + 1)),
                ("w", C.c_void_p * MAX_LAYERS), ("b", C.c_void_p * MAX_LAYERS)]


class Model(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("gamma", C.c_float), ("obs_dim", C.c_int32),
                ("num_actions", C.c_int32), ("num_atoms", C.c_int32),
                ("policy_layers", C.c_int32), ("value_layers", C.c_int32),
                ("policy_sizes", C.c_int32 * (MAX_LAYERS + 1)),
                ("value_sizes", C.c_int32 * (MAX_LAYERS + 1)), ("hidden_bf16", C.c_int32),
                ("packed", C.c_void_p), ("norm_mean", C.c_void_p), ("norm_std", C.c_void_p)]


class Replay(C.Structure):
    """az_replay (include/az_replay.h)."""
    _fields_ = [("rows", C.c_void_p), ("priority", C.c_void_p), ("weight", C.c_void_p),
                ("cum", C.c_void_p), ("partial", C.c_void_p), ("owner", C.c_void_p),
                ("cap", C.c_int32), ("D", C.c_int32)]


class ReplayState(C.Structure):
    """az_replay_state (include/az_replay.h)."""
    _fields_ = [("head", C.c_int32), ("insert_position", C.c_int32),
                ("sample_position", C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float
_K2 = C.POINTER(C.c_uint32 * 2)

BREAKOUT_WORDS, BREAKOUT_OBS, BREAKOUT_ACTIONS = 20, 400, 3   # include/az_minatar.h

# name -> (restype, argtypes); every symbol declared in include/az_mcts.h, az_replay.h, az_minatar.h
SIGNATURES = {
    "az_replay_partial_floats": (C.c_int64, [_I]),
    "az_replay_init": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState), _P]),
    "az_replay_insert": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState), _P, _I, _P]),
    "az_replay_sample_prioritized": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState), _K2, _F,
                                               _I, _P, _P, _P]),
    "az_replay_sample_prioritized_noalloc": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState),
                                                       _K2, _F, _I, _P, _P, _P]),
    "az_replay_sample_uniform": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState), _K2, _I, _P,
                                           _P, _P]),
    "az_replay_set_priorities": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState), _P, _P, _I,
                                           _P]),
    "az_replay_export": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState), _P, _P]),
    "az_replay_scan": (C.c_int, [C.POINTER(Replay), C.POINTER(ReplayState), _P, _P, _P]),
    "az_last_error": (C.c_char_p, []),
    "az_device_count": (C.c_int, []),
    "az_build_arch": (C.c_char_p, []),
    "az_considered_table": (C.c_int, [_I, _I, _P]),
    "az_sim_keys": (C.c_int, [_K2, _I, _P]),
    "az_tree_init": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs), _P, _P, _P, _P, _P]),
    "az_search_injected": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs),
                                     C.POINTER(SearchInputs), C.POINTER(Injected), _P, _P]),
    "az_simulate": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs), C.POINTER(SearchInputs),
                              _I, _P, _P, _P, _P, _P]),
    "az_expand_backward": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs), _P, _P, _P, _P,
                                     _P, _P, _P, _P]),
    "az_model_packed_floats": (C.c_int64, [C.POINTER(MlpParams), C.POINTER(MlpParams)]),
    "az_model_pack": (C.c_int, [C.POINTER(MlpParams), C.POINTER(MlpParams), _P, _P]),
    "az_model_packed_floats_bf16": (C.c_int64, [C.POINTER(MlpParams), C.POINTER(MlpParams)]),
    "az_model_pack_bf16": (C.c_int, [C.POINTER(MlpParams), C.POINTER(MlpParams), _P, _P]),
    "az_forward": (C.c_int, [C.POINTER(Model), _I, _P, _P, _P, _P]),
    "az_recurrent_step": (C.c_int, [C.POINTER(Model), _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "az_search_fused": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs),
                                  C.POINTER(SearchInputs), C.POINTER(Model), _P, _P]),
    "az_root_muzero": (C.c_int, [_I, _I, _P, _P, _F, _P, _P, _P]),
    "az_root_sampled": (C.c_int, [_I, _I, _P, _P, _F, _P, _I, _F, _F, _P, _P, _P, _P]),
    "az_root_gumbel": (C.c_int, [_I, _I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "az_dirichlet": (C.c_int, [_I, _I, _F, _P, _P, _P]),
    "az_epilogue_visits": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs), _F, _P, _P, _P,
                                     _P, _P]),
    "az_epilogue_gumbel": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs), _P, _P, _P, _P,
                                     _P, _P]),
    "az_summary": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs), _P, _P, _P, _P, _P]),
    "az_actor_post": (C.c_int, [C.POINTER(SearchDesc), C.POINTER(TreePtrs), _P, _P, _I, _P, _P,
                                _P, _P, _P]),
    "az_split_keys": (C.c_int, [_P, _I, _P, _P]),
    "az_random_sort_keys": (C.c_int, [_P, _I, _P, _P]),
    "az_breakout_reset": (C.c_int, [_I, _P, _P, _P, _P, _P]),
    "az_breakout_step": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "az_breakout_real_step": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "az_env_reset": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P]),
    "az_real_env_step": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "az_math_apply": (C.c_int, [_I, _I, _P, _P, _P]),
    "az_threefry": (C.c_int, [_I, _P, _P, _P, _P]),
}

_lib = None


def lib():
    """The loaded library; raises NativeLibraryError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build it with `make -C alphazero_amd/csrc` "
                "(or __graft_entry__.build()); this package has no fallback path")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().az_last_error().decode("utf-8", "replace")
        kind = {-1: ValueError, -2: NotImplementedError}.get(rc, RuntimeError)
        raise kind(f"libaz_mcts: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor handed to the native library must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
