"""CPU oracle of the GridworldCtf hot path — TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg,
and there only as the checker / the reported CPU baseline.  The product package
(``marl-ctf-development_amd``) never imports this module.

Parity status: PINNED against the reference itself (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference in the build container).
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libctf_oracle.so")

_abi = importlib.import_module("marl-ctf-development_amd._abi")
CtfConfig, CtfStateView = _abi.CtfConfig, _abi.CtfStateView

_lib = None


def build(force=False):
    src = os.path.join(_HERE, "ctf_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        P = C.c_void_p
        L.octf_create.restype = P
        L.octf_create.argtypes = [C.POINTER(CtfConfig)]
        L.octf_destroy.argtypes = [P]
        L.octf_seed.argtypes = [P, C.c_uint64, C.c_uint64]
        L.octf_set_rng_state.argtypes = [P, P, P]
        L.octf_get_rng_state.argtypes = [P, P, P]
        L.octf_get_rng_counters.argtypes = [P, P]
        L.octf_set_rng_counters.argtypes = [P, P]
        L.octf_reset.argtypes = [P]
        L.octf_step.restype = C.c_uint32
        L.octf_step.argtypes = [P, P, P, P]
        L.octf_observe.argtypes = [P, P, P, C.c_uint32]
        L.octf_get_state.argtypes = [P, C.POINTER(CtfStateView)]
        L.octf_set_state.argtypes = [P, C.POINTER(CtfStateView)]
        L.octf_f64_to_f16.restype = C.c_uint16
        L.octf_f64_to_f16.argtypes = [C.c_double]
        L.octf_philox_actions.argtypes = [P, C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32]
        L.octf_run_batch.restype = C.c_uint64
        L.octf_run_batch.argtypes = [C.POINTER(CtfConfig), C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]
        L.octf_bench_digest.argtypes = [C.POINTER(CtfConfig), C.c_int32, P, C.c_uint32, C.c_int32, C.c_uint64, C.c_int32, C.c_uint64,
                                        C.c_int32, P, P, P, P, P, P]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """One environment of the C oracle, driven with numpy arrays."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.n, self.g, self.c = cfg.n_agents, cfg.grid_size, cfg.n_channels
        self.m = 2 * self.n + 6
        self._h = lib().octf_create(C.byref(cfg))
        if not self._h:
            raise MemoryError

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().octf_destroy(self._h)
            except TypeError:  # interpreter shutdown: the module's globals are already gone
                pass
            self._h = None

    def seed(self, py_seed, np_seed):
        lib().octf_seed(self._h, int(py_seed), int(np_seed))

    def set_rng_state(self, py_mt625=None, np_mt625=None):
        a = None if py_mt625 is None else np.ascontiguousarray(py_mt625, dtype=np.uint32)
        b = None if np_mt625 is None else np.ascontiguousarray(np_mt625, dtype=np.uint32)
        lib().octf_set_rng_state(self._h, None if a is None else _ptr(a), None if b is None else _ptr(b))

    def get_rng_state(self):
        a = np.zeros(625, np.uint32)
        b = np.zeros(625, np.uint32)
        lib().octf_get_rng_state(self._h, _ptr(a), _ptr(b))
        return a, b

    def get_rng_counters(self):
        """counter mode: (words consumed from the `random` tape, ... from the np.random tape)"""
        a = np.zeros(2, np.uint64)
        lib().octf_get_rng_counters(self._h, _ptr(a))
        return int(a[0]), int(a[1])

    def reset(self):
        lib().octf_reset(self._h)

    def step(self, actions):
        act = np.ascontiguousarray(actions, dtype=np.int8)
        assert act.shape == (self.n,)
        rewards = np.zeros(self.n, np.float64)
        done = np.zeros(1, np.uint8)
        status = lib().octf_step(self._h, _ptr(act), _ptr(rewards), _ptr(done))
        return rewards, bool(done[0]), int(status)

    def observe(self, reverse_mask=_abi.REVERSE_DEFAULT):
        obs = np.zeros((self.n, self.c, self.g, self.g), np.uint8)
        meta = np.zeros((self.n, self.m), np.uint16)
        lib().octf_observe(self._h, _ptr(obs), _ptr(meta), int(reverse_mask))
        return obs, meta.view(np.float16)

    def get_state(self):
        v = CtfStateView()
        lib().octf_get_state(self._h, C.byref(v))
        return v

    def set_state(self, view):
        lib().octf_set_state(self._h, C.byref(view))


def f64_to_f16_bits(x):
    return lib().octf_f64_to_f16(float(x))


def philox_actions(n_agents, seed, step, env_index):
    out = np.zeros(n_agents, np.int8)
    lib().octf_philox_actions(_ptr(out), n_agents, int(seed), int(step), int(env_index))
    return out


def run_batch(cfg, n_envs, n_steps, seed_base=0, action_seed=0, with_observe=True, n_threads=1):
    return lib().octf_run_batch(C.byref(cfg), n_envs, n_steps, int(seed_base), int(action_seed), int(with_observe), int(n_threads))


DIGEST_MULT = 0x9E3779B97F4A7C15


def digest_weights(n):
    """w(i) = (i + 1) * DIGEST_MULT mod 2**64 as uint64: a digest is sum(value_i * w(i)) mod 2**64 (octf_bench_digest)."""
    return (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(DIGEST_MULT)).astype(np.uint64)


def bench_digest(cfg, seeds, env_offset, period, stagger_seed, n_steps, action_seed, n_threads):
    """bench.py's protocol on the CPU for every env -> dict of per-step / per-env digests (see ctf_oracle.c)."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    E = int(seeds.shape[0])
    out = dict(obs=np.zeros((n_steps, E), np.uint64), meta=np.zeros((n_steps, E), np.uint64), rew=np.zeros((n_steps, E), np.uint64),
               done=np.zeros((n_steps, E), np.uint8), rng=np.zeros((E, 2), np.uint64), misc=np.zeros((E, 3), np.int32))
    lib().octf_bench_digest(C.byref(cfg), E, _ptr(seeds), int(env_offset), int(period), int(stagger_seed), int(n_steps), int(action_seed),
                            int(n_threads), _ptr(out["obs"]), _ptr(out["meta"]), _ptr(out["rew"]), _ptr(out["done"]), _ptr(out["rng"]),
                            _ptr(out["misc"]))
    return out
