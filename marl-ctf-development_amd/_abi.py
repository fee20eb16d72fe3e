"""ctypes mirror of include/ctf_env.h and the loader of the HIP shared library.

The product path has no CPU fallback: if ``libctf_hip.so`` is missing or does not load, every
entry point raises (``CtfLibraryError``) instead of routing anywhere else.
"""
import ctypes as C
import os

ABI_VERSION = 2
MAX_AGENTS = 16
MAX_GRID = 32
MAX_CELLS = MAX_GRID * MAX_GRID
MAX_CHANNELS = 16
N_ACTIONS = 9
N_METRICS = 13
MT_N = 624

ST_BAD_ACTION = 1
ST_NO_RESPAWN = 2
ST_SPAWN_EDGE = 4
ST_RNG_OVERRUN = 8
STEP_AUTO_RESET = 1
RNG_MT19937 = 0
RNG_COUNTER = 1
REVERSE_DEFAULT = 0xFFFFFFFF

# order of ctf_state_view.metrics rows == the reference's "agent_*" metric names (gridworld_ctf.py:456-468)
METRIC_NAMES = (
    "tag_count",
    "respawn_tag_count",
    "flag_pickups",
    "flag_captures",
    "flag_dispossessions",
    "blocks_laid",
    "blocks_mined",
    "blocks_laid_distance_from_own_flag",
    "blocks_laid_distance_from_opp_flag",
    "steps_defending_zone",
    "steps_attacking_zone",
    "steps_adj_teammate",
    "steps_adj_opponent",
)


class CtfConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_agents", C.c_int32),
        ("grid_size", C.c_int32),
        ("n_channels", C.c_int32),
        ("game_steps", C.c_int32),
        ("flip_axis", C.c_int32),
        ("home_flag_capture", C.c_int32),
        ("use_adjusted_rewards", C.c_int32),
        ("drop_flag_when_no_hp", C.c_int32),
        ("log_metrics", C.c_int32),
        ("n_opponents", C.c_int32 * 2),
        ("rng_mode", C.c_int32),
        ("reserved0", C.c_int32 * 5),
        ("heal_per_step", C.c_double),
        ("tag_probability", C.c_double),
        ("guardian_damage_multiplier", C.c_double),
        ("vault_hp_cost", C.c_double),
        ("vault_min_hp", C.c_double),
        ("reward_capture", C.c_double),
        ("reward_step", C.c_double),
        ("reward_tag", C.c_double),
        ("win_margin_scalar", C.c_double),
        ("loss_margin_scalar", C.c_double),
        ("opp_capture_punishment", C.c_double),
        ("type_hp", C.c_double * 4),
        ("type_damage", C.c_double * 4),
        ("agent_team", C.c_int8 * MAX_AGENTS),
        ("agent_type", C.c_int8 * MAX_AGENTS),
        ("opponents", (C.c_int8 * MAX_AGENTS) * 2),
        ("flag_pos", (C.c_int8 * 2) * 2),
        ("capture_pos", (C.c_int8 * 2) * 2),
        ("spawn_pos", (C.c_int8 * 2) * 2),
        ("start_pos", (C.c_int8 * 2) * MAX_AGENTS),
        ("tile_of_channel", C.c_uint8 * MAX_CHANNELS),
        ("init_grid", C.c_uint8 * MAX_CELLS),
    ]


class CtfStateView(C.Structure):
    _fields_ = [
        ("grid", C.c_uint8 * MAX_CELLS),
        ("pos", (C.c_int8 * 2) * MAX_AGENTS),
        ("hp", C.c_double * MAX_AGENTS),
        ("has_flag", C.c_uint8 * MAX_AGENTS),
        ("inventory", C.c_int32 * MAX_AGENTS),
        ("perm", C.c_uint8 * MAX_AGENTS),
        ("step_count", C.c_int32),
        ("done", C.c_int32),
        ("team_captures", C.c_int32 * 2),
        ("metrics", (C.c_int32 * MAX_AGENTS) * N_METRICS),
        ("visitation", (C.c_uint8 * MAX_CELLS) * MAX_AGENTS),
    ]


class CtfLibraryError(RuntimeError):
    pass


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# CTF_LIB_PATH: profiling builds only (tools/ablate.sh); the default is the in-tree library
LIB_PATH = os.environ.get("CTF_LIB_PATH") or os.path.join(_PKG_DIR, "csrc", "libctf_hip.so")

# every symbol include/ctf_env.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "ctf_create": (C.c_int, [C.POINTER(CtfConfig), C.c_int32, C.c_int32, C.POINTER(_P)]),
    "ctf_destroy": (None, [_P]),
    "ctf_n_envs": (C.c_int32, [_P]),
    "ctf_obs_bytes_per_env": (C.c_int64, [_P]),
    "ctf_meta_elems_per_env": (C.c_int64, [_P]),
    "ctf_seed": (C.c_int, [_P, _P, _P, _P]),
    "ctf_set_rng_state": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ctf_get_rng_state": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ctf_set_rng_states": (C.c_int, [_P, _P, _P, _P]),
    "ctf_get_rng_states": (C.c_int, [_P, _P, _P, _P]),
    "ctf_get_rng_counters": (C.c_int, [_P, _P, _P]),
    "ctf_set_rng_counters": (C.c_int, [_P, _P, _P]),
    "ctf_reset": (C.c_int, [_P, _P, _P]),
    "ctf_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint32, _P]),
    "ctf_observe": (C.c_int, [_P, _P, _P, C.c_uint32, _P]),
    "ctf_observe_kernel": (C.c_int32, [_P, _P]),
    "ctf_observe_stores_hinted": (C.c_int32, [_P, _P]),
    "ctf_observe_codes": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P]),
    "ctf_step_observe": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, _P]),
    "ctf_action_mask": (C.c_int, [_P, _P]),
    "ctf_get_state": (C.c_int, [_P, C.c_int32, C.POINTER(CtfStateView)]),
    "ctf_set_state": (C.c_int, [_P, C.c_int32, C.POINTER(CtfStateView)]),
    "ctf_host_step": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_int32), C.POINTER(C.c_uint32),
                                C.POINTER(CtfStateView), _P, _P, _P, _P, _P]),
    "ctf_export_counters": (C.c_int, [_P, _P, _P, _P, _P]),
    "ctf_status": (C.c_int, [_P, C.POINTER(C.c_uint32), _P]),
    "ctf_random_actions": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P]),
    "ctf_last_error": (C.c_char_p, []),
    "ctf_abi_version": (C.c_int32, []),
    "ctf_sizeof_config": (C.c_int32, []),
    "ctf_sizeof_state_view": (C.c_int32, []),
    # include/ctf_policy.h
    "ctf_policy_act_stride": (C.c_int32, [C.c_int32, C.c_int32]),
    "ctf_policy_features": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                      _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_features_train": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_front_dgrad": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_front_wgrad": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int32, _P]),
    "ctf_policy_front_backward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_linear_wgrad": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P]),
    "ctf_policy_head": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P,
                                  C.c_int32, _P]),
    "ctf_policy_fact_view_stride": (C.c_int32, [C.c_int32]),
    "ctf_policy_fact_row_stride": (C.c_int32, [C.c_int32]),
    "ctf_policy_fact_max_tiles": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "ctf_policy_fact_bucket": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_features_fact": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                           _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_view_gemm": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "ctf_policy_fc1_patch": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "ctf_policy_fc1_patch_head": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32,
                                            C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_fc1_dgrad": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "ctf_rollout_store_step": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32),
                                         C.c_int32, _P, _P, _P, _P, C.POINTER(C.c_uint8), C.c_uint32, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "ctf_policy_set_deterministic": (C.c_int, [C.c_int32, _P, C.c_int64]),
    "ctf_policy_deterministic_workspace": (C.c_int64, [C.c_int32]),
    "ctf_policy_last_error": (C.c_char_p, []),
}

_lib = None


def load_library(path=None):
    """dlopen libctf_hip.so (once) and type every entry point.  Raises CtfLibraryError when the
    library is absent — there is deliberately no fallback path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise CtfLibraryError(
            f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C {os.path.join(_PKG_DIR, 'csrc')}); there is no CPU fallback"
        )
    try:
        # torch (if the caller uses it) must come first so that its bundled libamdhip64.so.7 is the one
        # HIP runtime in the process; our library then binds to the already-loaded soname.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less callers use /opt/rocm's runtime
        pass
    lib = bind(path)
    _lib = lib
    return lib


def bind(path, mode=C.RTLD_GLOBAL, optional=()):
    """dlopen one build of the library and type its entry points (tools/ab_inproc.py loads several side by side; only that
    tool passes ``optional``: name prefixes a partial or older build may lack)."""
    try:
        lib = C.CDLL(path, mode=mode)
    except OSError as exc:
        raise CtfLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            if any(name.startswith(o) for o in optional):
                continue
            raise CtfLibraryError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.ctf_abi_version() != ABI_VERSION:
        raise CtfLibraryError(f"ABI version mismatch: library {lib.ctf_abi_version()} != binding {ABI_VERSION}")
    if lib.ctf_sizeof_config() != C.sizeof(CtfConfig) or lib.ctf_sizeof_state_view() != C.sizeof(CtfStateView):
        raise CtfLibraryError("struct layout mismatch between include/ctf_env.h and the ctypes mirror")
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        msg = lib.ctf_last_error()
        raise CtfLibraryError(f"ctf call failed ({rc}): {msg.decode() if msg else ''}")
