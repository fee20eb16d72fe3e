"""Host-side mirror of the reference's ``GridworldCtf`` (reference gridworld_ctf.py:13) over the C ABI.

Two classes:

``VecGridworldCtf``  E independent envs resident in HBM, stepped and rendered by HIP kernels; all I/O is
                     torch CUDA tensors handed to the C ABI as raw device pointers on the caller's
                     current HIP stream.  This is the fast path.
``GridworldCtf``     the reference's single-env API (same constructor kwargs, methods, attributes and
                     exceptions) as a batch of one, so ``ppo.py`` / ``utils.duel`` / the ``0_..8_*.py``
                     scripts run against it unchanged (put this directory first on sys.path).

There is no CPU fallback: without the HIP library and a GPU, construction raises.
"""
import ctypes as C
import random as _py_random
from collections import defaultdict
from functools import partial

import numpy as np

try:
    from . import _abi, config as _config
except ImportError:  # pragma: no cover - this directory itself on sys.path (drop-in use)
    import _abi
    import config as _config

__all__ = ["GridworldCtf", "VecGridworldCtf"]


def _torch():
    import torch

    return torch


def _as_seed_array(seeds, n):
    if seeds is None:
        seeds = np.arange(n, dtype=np.uint64)
    arr = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(-1))
    if arr.shape[0] != n:
        raise ValueError(f"need {n} seeds, got {arr.shape[0]}")
    return arr


def expand_codes(codes, n_channels):
    """codes uint8 [..., G, G] (observe_codes) -> the one-hot planes uint8 [..., C, G, G] of standardise_state
    (torch tensor or numpy array)."""
    if isinstance(codes, np.ndarray):
        k = np.arange(n_channels, dtype=np.uint8).reshape((n_channels, 1, 1))
        low, c = (codes & 0x7F)[..., None, :, :], codes[..., None, :, :]
        return np.where(k == 0, c >> 7, (low == k) & (k > 0)).astype(np.uint8)
    torch = _torch()
    k = torch.arange(n_channels, dtype=torch.uint8, device=codes.device).reshape((n_channels, 1, 1))
    low, c = (codes & 0x7F).unsqueeze(-3), codes.unsqueeze(-3)
    return torch.where(k == 0, c >> 7, ((low == k) & (k > 0)).to(torch.uint8))


class VecGridworldCtf:
    """``n_envs`` GridworldCtf instances on one MI355X.

    Env ``e`` reproduces, bit for bit, a reference process that ran
    ``random.seed(py_seeds[e]); np.random.seed(np_seeds[e])`` before constructing its env and was fed
    the same actions.  Outputs live in tensors owned by this object and are overwritten by the next call.
    """

    def __init__(self, n_envs, device=None, py_seeds=None, np_seeds=None, log_metrics=True, tune_placement=None, _lib=None,
                 rng_mode="mt19937", placement_tries=None, placement_gib=None, **env_kwargs):
        """tune_placement: pick the observation buffer among a few candidate allocations by timing the render into each
        (default: on for batches whose observation block exceeds 256 MiB).  On MI355X about half of all large hipMalloc
        allocations stream 20 % slower than the others (6.5 vs 5.3 TB/s for a bare store stream into the very same
        virtual address range after a free / re-allocate: it is the physical backing, profiles/r02_alloc_probe*.txt)."""
        torch = _torch()
        self._lib = _lib or _abi.load_library()  # _lib: a side-by-side build, profiling only (tools/ab_inproc.py)
        if rng_mode not in ("mt19937", "counter"):
            raise ValueError("rng_mode must be 'mt19937' (the reference's generators, bit for bit) or 'counter'")
        # "counter": opt-in counter-based streams (include/ctf_env.h CTF_RNG_COUNTER) — word n of an env's stream is
        # Philox4x32-10(seed, n); the draws are made from those words by the reference's own rules
        self.rng_mode = rng_mode
        self.cfg, self.derived = _config.build_config(env_kwargs, log_metrics=log_metrics,
                                                      rng_mode=_abi.RNG_COUNTER if rng_mode == "counter" else _abi.RNG_MT19937)
        if not torch.cuda.is_available():
            raise _abi.CtfLibraryError("no HIP device visible: the GridworldCtf kernels need a GPU (there is no CPU fallback)")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.n_envs = int(n_envs)
        d = self.derived
        self.N_AGENTS, self.GRID_SIZE = d["n_agents"], d["grid_size"]
        self.N_CHANNELS, self.META_LEN = self.cfg.n_channels, 2 * d["n_agents"] + 6
        self.AGENT_TEAMS, self.AGENT_TYPES = d["agent_teams"], d["agent_types"]
        self.TILES_USED = d["tiles_used"]
        with torch.cuda.device(self.device):
            torch.cuda.current_stream()  # make sure torch has initialised HIP on this device first
            h = C.c_void_p()
            _abi.check(self._lib.ctf_create(C.byref(self.cfg), self.n_envs, self.device.index, C.byref(h)), self._lib)
        self._h = h
        E, N = self.n_envs, self.N_AGENTS
        self.rewards = torch.zeros((E, N), dtype=torch.float32, device=self.device)
        self.rewards64 = torch.zeros((E, N), dtype=torch.float64, device=self.device)
        self.done = torch.zeros((E,), dtype=torch.uint8, device=self.device)
        self._obs = None  # uint8 [E, N, C, G, G]: allocated (and placed) on first use — a codes-only caller never pays for it
        self._codes = None
        self._self_cells = None
        self.meta = torch.zeros((E, N, self.META_LEN), dtype=torch.float16, device=self.device)
        self.seed(py_seeds, np_seeds)
        if tune_placement is None:
            tune_placement = E * N * self.N_CHANNELS * self.GRID_SIZE ** 2 > (256 << 20)
        self._tune_placement = bool(tune_placement)
        import os

        self._placement_tries = int(placement_tries if placement_tries is not None else os.environ.get("CTF_PLACEMENT_TRIES", 256))
        self._placement_gib = float(placement_gib if placement_gib is not None else os.environ.get("CTF_PLACEMENT_GIB", 16))
        self._placement_seconds = float(os.environ.get("CTF_PLACEMENT_SECONDS", 3.0))
        self.placement_probe_ms = None
        self.placement_fill_ms = None
        self.placement = None  # what the placement search found: kind fast / intermediate / slow, render / fill ratio, ...

    @property
    def obs(self):
        if self._obs is None:
            torch = _torch()
            self._obs = torch.zeros((self.n_envs, self.N_AGENTS, self.N_CHANNELS, self.GRID_SIZE, self.GRID_SIZE),
                                    dtype=torch.uint8, device=self.device)
            if self._tune_placement:
                self._tune_obs_placement()
        return self._obs

    @obs.setter
    def obs(self, buf):
        self._obs = buf

    @property
    def codes(self):
        if self._codes is None:
            self._codes = _torch().zeros((self.n_envs, self.N_AGENTS, self.GRID_SIZE, self.GRID_SIZE), dtype=_torch().uint8,
                                         device=self.device)
        return self._codes

    def _tune_obs_placement(self, good_enough=0.95):
        """Keep the candidate allocation the render streams into fastest (see __init__).

        BOUNDED in what it HOLDS: one candidate beside the best one so far — two observation buffers, 3.3 GB for the arena batch —
        never more than ``placement_gib`` GiB (default 16; CTF_PLACEMENT_GIB; a batch whose two buffers exceed it is not searched)
        nor 45 % of the free device memory: a co-resident policy / learner is not starved while this searches.  A rejected candidate
        goes back to the driver at once (``torch.cuda.empty_cache``); the next allocation is of an independent kind even when nothing
        is held in between (tools/placement_probe2.py, profiles/r03_placement_search_strategies.txt: holding the rejected ones, as
        round 2 did, finds fast buffers no more often), so the search can afford ``placement_tries`` candidates (default 256;
        CTF_PLACEMENT_TRIES) at a few milliseconds each.  An allocation failure ends the search with what it has.

        The search stops at a candidate whose render takes at most ``good_enough`` x the time of a plain ``fill_`` of the same
        buffer (which does not depend on the buffer's kind; 0.95 is the very top of what any workload has shown, so in practice the
        time limit ends the search: stopping at 1.00 cost the 20 x 20 map 5-8 % when the first candidate under it was a 0.99), or — once it holds one of the fast kind (render / fill <= 1.08: with the
        render's nontemporal stores the kinds lie at 1.02-1.07 and 1.15-1.26 for the 15 x 15 arena, from 0.95 in a continuum for the
        20 x 20 one, the fill itself 0.234-0.245 ms from box to box; profiles/r05_render_nontemporal.md) — when
        ``placement_seconds`` of wall time are used up; ten seconds while it holds none.  ``self.placement`` says what was found."""
        import time

        torch = _torch()
        stream = torch.cuda.current_stream(self.device)

        def timed(fn, reps=3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
            b.synchronize()
            return a.elapsed_time(b) / reps

        def probe(buf):
            self.obs = buf
            return timed(lambda: self.observe(meta=False))

        t0 = time.perf_counter()
        best = self.obs
        nbytes = max(1, best.numel())
        free_bytes, _ = torch.cuda.mem_get_info(self.device)
        budget = min(int(self._placement_gib * (1 << 30)), int(0.45 * free_bytes) + nbytes)  # (the first buffer is already ours)
        tries = max(1, int(self._placement_tries)) if 2 * nbytes <= budget else 1
        best_ms = probe(best)
        times = [best_ms]
        fill_ms = timed(lambda: best.fill_(0))
        # Once a buffer of the good cluster is in hand the search goes on for its BEST members — the cluster itself spans 5 % of the
        # render's time (round 5, plain stores: one box's two runs kept 0.2614 and 0.2497 ms = 199 and 207 M env-steps/s) and a
        # candidate costs 1.5 ms — until one is good enough, the tries are used up, or `placement_seconds` (default 3.0;
        # CTF_PLACEMENT_SECONDS) have gone by (3.3 times that, i.e. ten seconds by default, while no buffer of the fast kind has turned up at all).  (Round 3's rule — eight more tries — found a second good one 4 times in 10.)
        for _ in range(tries - 1):
            if best_ms <= good_enough * fill_ms:
                break
            elapsed = time.perf_counter() - t0
            in_hand = best_ms <= 1.08 * fill_ms  # a buffer of the fast kind (the relative test of round 3 — 7 % under the slowest seen —
                                                 # misfires on one outlier: a slow-only box stopped at 1.19 after 3.7 s)
            if (in_hand and elapsed > self._placement_seconds) or elapsed > min(10.0, 3.3 * self._placement_seconds):
                break  # (the second bound: a box that hands out slow allocations only — one in six to thirteen fresh boxes; at 50 ms a
                       # candidate ten seconds are ~200 tries, enough where one allocation in fifty is fast)
            try:
                cand = torch.empty_like(best)
            except torch.cuda.OutOfMemoryError:
                break  # (fragmentation, another process on the GPU): the best one so far is kept
            ms = probe(cand)
            times.append(ms)
            if ms < best_ms:
                best, best_ms = cand, ms
            self.obs = best
            del cand
            torch.cuda.empty_cache()  # the loser goes back to the driver now, not into torch's cache
        self.obs = best
        fill_ms = timed(lambda: best.fill_(0))  # of the buffer that is kept
        ratio = best_ms / fill_ms
        self.placement_probe_ms = times
        self.placement_fill_ms = fill_ms
        self.placement = dict(kind="fast" if ratio <= 1.08 else ("intermediate" if ratio <= 1.14 else "slow"), render_over_fill=ratio,
                              render_ms=best_ms, fill_ms=fill_ms, candidates=len(times), slowest_candidate_render_ms=max(times),
                              peak_held_bytes=min(len(times), 2) * nbytes, searched_bytes=len(times) * nbytes,
                              search_ms=(time.perf_counter() - t0) * 1e3)

    # -- plumbing -----------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ctf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_dev(self, t, dtype, numel):
        torch = _torch()
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == self.device and t.dtype == dtype
                and t.is_contiguous() and t.numel() == numel):
            raise ValueError(f"expected a contiguous {dtype} tensor of {numel} elements on {self.device}")
        return C.c_void_p(t.data_ptr())

    # -- RNG ----------------------------------------------------------------------------------
    def seed(self, py_seeds=None, np_seeds=None):
        py = _as_seed_array(py_seeds, self.n_envs)
        npz = _as_seed_array(np_seeds if np_seeds is not None else py_seeds, self.n_envs)
        if self.rng_mode == "mt19937" and (npz >> np.uint64(32)).any():
            raise ValueError("Seed must be between 0 and 2**32 - 1")  # np.random.seed's own message
        _abi.check(self._lib.ctf_seed(self._h, py.ctypes.data_as(C.c_void_p), npz.ctypes.data_as(C.c_void_p), self._stream()), self._lib)
        _torch().cuda.current_stream(self.device).synchronize()  # host seed arrays may go away

    def set_rng_state(self, env_index, py_mt625=None, np_mt625=None):
        a = None if py_mt625 is None else np.ascontiguousarray(py_mt625, dtype=np.uint32)
        b = None if np_mt625 is None else np.ascontiguousarray(np_mt625, dtype=np.uint32)
        _abi.check(self._lib.ctf_set_rng_state(self._h, env_index, None if a is None else a.ctypes.data_as(C.c_void_p),
                                               None if b is None else b.ctypes.data_as(C.c_void_p)), self._lib)

    def get_rng_state(self, env_index):
        a, b = np.zeros(625, np.uint32), np.zeros(625, np.uint32)
        _abi.check(self._lib.ctf_get_rng_state(self._h, env_index, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)), self._lib)
        return a, b

    def set_rng_states(self, py_states=None, np_states=None):
        """Bulk, stream-ordered hand-over: uint32 CUDA tensors [E, 625] (624 words + position per env), either may be None."""
        torch = _torch()
        ptr = lambda t: None if t is None else self._check_dev(t, torch.int32 if t.dtype == torch.int32 else torch.uint32, self.n_envs * 625)
        _abi.check(self._lib.ctf_set_rng_states(self._h, ptr(py_states), ptr(np_states), self._stream()), self._lib)

    def get_rng_states(self, py=True, np_=True):
        """-> (py_states, np_states): int32 CUDA tensors [E, 625] holding the uint32 words of both generators of every env in
        the standard form (None for a generator not asked for).  Stream-ordered; no device synchronisation."""
        torch = _torch()
        a = torch.empty((self.n_envs, 625), dtype=torch.int32, device=self.device) if py else None
        b = torch.empty((self.n_envs, 625), dtype=torch.int32, device=self.device) if np_ else None
        _abi.check(self._lib.ctf_get_rng_states(self._h, None if a is None else C.c_void_p(a.data_ptr()),
                                                None if b is None else C.c_void_p(b.data_ptr()), self._stream()), self._lib)
        return a, b

    def get_rng_counters(self):
        """counter mode: int64 CUDA tensor [E, 2] = words consumed so far from the `random` / np.random stream of every env (the
        whole RNG state of an env there, besides its two seeds).  Stream-ordered."""
        out = _torch().empty((self.n_envs, 2), dtype=_torch().int64, device=self.device)
        _abi.check(self._lib.ctf_get_rng_counters(self._h, C.c_void_p(out.data_ptr()), self._stream()), self._lib)
        return out

    def set_rng_counters(self, counters):
        """counter mode: the way back (int64 CUDA tensor [E, 2]) — a checkpoint restore after ``seed``."""
        ptr = self._check_dev(counters, _torch().int64, self.n_envs * 2)
        _abi.check(self._lib.ctf_set_rng_counters(self._h, ptr, self._stream()), self._lib)

    # -- the hot path -------------------------------------------------------------------------
    def reset(self, mask=None):
        ptr = None if mask is None else self._check_dev(mask, _torch().uint8, self.n_envs)
        _abi.check(self._lib.ctf_reset(self._h, ptr, self._stream()), self._lib)

    def step(self, actions, auto_reset=False, want_f64=False):
        """actions: int8 CUDA tensor [E, N].  -> (rewards float32 [E, N], done uint8 [E]) (views of
        this object's buffers).  With want_f64 the float64 rewards are also written to ``rewards64``."""
        a = self._check_dev(actions, _torch().int8, self.n_envs * self.N_AGENTS)
        _abi.check(self._lib.ctf_step(self._h, a, C.c_void_p(self.rewards.data_ptr()),
                                      C.c_void_p(self.rewards64.data_ptr()) if want_f64 else None,
                                      C.c_void_p(self.done.data_ptr()), _abi.STEP_AUTO_RESET if auto_reset else 0,
                                      self._stream()), self._lib)
        return self.rewards, self.done

    def observe(self, reverse_mask=None, obs=True, meta=True):
        """-> (obs uint8 [E, N, C, G, G], meta float16 [E, N, 2N+6]).  ``reverse_mask`` bit i = reverse_grid
        for agent i (default: team(i) == 1, what every caller in the reference passes)."""
        rm = _abi.REVERSE_DEFAULT if reverse_mask is None else int(reverse_mask) & ((1 << self.N_AGENTS) - 1)
        _abi.check(self._lib.ctf_observe(self._h, C.c_void_p(self.obs.data_ptr()) if obs else None,
                                         C.c_void_p(self.meta.data_ptr()) if meta else None, rm, self._stream()), self._lib)
        return self.obs, self.meta

    def observe_kernel(self):
        """Which kernel ``observe`` launches for this object's buffer: "k_observe_tiles" or "k_observe" (the library's own rule)."""
        return "k_observe_tiles" if self._lib.ctf_observe_kernel(self._h, C.c_void_p(self.obs.data_ptr())) == 1 else "k_observe"

    def observe_stores(self):
        """"nontemporal" when ``observe`` streams this object's buffer past the caches (the tile kernel on a batch whose observations
        exceed 320 MB), else "plain"."""
        return "nontemporal" if self._lib.ctf_observe_stores_hinted(self._h, C.c_void_p(self.obs.data_ptr())) == 1 else "plain"

    def observe_codes(self, reverse_mask=None, codes=True, meta=True):
        """The observation in compact form -> (codes uint8 [E, N, G, G], meta float16 [E, N, 2N+6]): low 7 bits = the tile
        plane (1..C-1) that is 1 at the cell, 0 = none; bit 7 = plane 0 (own position).  ``expand_codes`` gives the planes.
        ``self.self_cells`` (uint16 [E, N]) receives the cell index of every agent's bit 7 in the same launch."""
        rm = _abi.REVERSE_DEFAULT if reverse_mask is None else int(reverse_mask) & ((1 << self.N_AGENTS) - 1)
        if self._self_cells is None:
            self._self_cells = _torch().zeros((self.n_envs, self.N_AGENTS), dtype=_torch().int16, device=self.device)
        _abi.check(self._lib.ctf_observe_codes(self._h, C.c_void_p(self.codes.data_ptr()) if codes else None,
                                               C.c_void_p(self.meta.data_ptr()) if meta else None,
                                               C.c_void_p(self._self_cells.data_ptr()) if codes else None, rm, self._stream()), self._lib)
        return self.codes, self.meta

    @property
    def self_cells(self):
        """int16 [E, N]: where bit 7 sits in each agent's row of the last ``observe_codes`` (None before the first)."""
        return self._self_cells

    def step_observe(self, actions, auto_reset=False, want_f64=False, reverse_mask=None):
        """step() then observe() in one call -> (rewards, done, obs, meta)."""
        a = self._check_dev(actions, _torch().int8, self.n_envs * self.N_AGENTS)
        rm = _abi.REVERSE_DEFAULT if reverse_mask is None else int(reverse_mask) & ((1 << self.N_AGENTS) - 1)
        _abi.check(self._lib.ctf_step_observe(self._h, a, C.c_void_p(self.rewards.data_ptr()),
                                              C.c_void_p(self.rewards64.data_ptr()) if want_f64 else None,
                                              C.c_void_p(self.done.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                              C.c_void_p(self.meta.data_ptr()), rm,
                                              _abi.STEP_AUTO_RESET if auto_reset else 0, self._stream()), self._lib)
        return self.rewards, self.done, self.obs, self.meta

    def host_step(self, actions=None, py_in=None, np_in=None, reverse_mask=None, rng_out=False, view=None, obs=None, meta=None):
        """ctf_host_step (a handle of ONE env): everything in host memory, one round trip to the device.  actions: int8 numpy [N]
        or None (no step); py_in / np_in: uint32 numpy [625] to install before the step; -> (rewards float64 [N], done, status
        bits, view, py_out, np_out) with obs (uint8 [N, C, G, G]) and meta (float16 [N, M]) filled in place when given."""
        n = self.N_AGENTS
        rm = _abi.REVERSE_DEFAULT if reverse_mask is None else int(reverse_mask) & ((1 << n) - 1)
        ptr = lambda a: None if a is None else a.__array_interface__["data"][0]  # (the plain address: a third of the cost of ctypes.data_as)
        rewards = np.zeros(n, np.float64)
        done, status = C.c_int32(0), C.c_uint32(0)
        view = view if view is not None else _abi.CtfStateView()
        py_out = np.empty(625, np.uint32) if rng_out else None
        np_out = np.empty(625, np.uint32) if rng_out else None
        _abi.check(self._lib.ctf_host_step(self._h, ptr(actions), ptr(py_in), ptr(np_in), rm, 0, ptr(rewards), C.byref(done),
                                           C.byref(status), C.byref(view), ptr(py_out), ptr(np_out), ptr(obs), ptr(meta),
                                           self._stream()), self._lib)
        return rewards, bool(done.value), status.value, view, py_out, np_out

    def random_actions(self, out, seed, step, env_offset=0):
        """Fill ``out`` (int8 [E, N]) with the synthetic Philox action stream of bench.py / the tests."""
        a = self._check_dev(out, _torch().int8, self.n_envs * self.N_AGENTS)
        _abi.check(self._lib.ctf_random_actions(self._h, a, int(seed), int(step), int(env_offset), self._stream()), self._lib)
        return out

    def counters(self):
        """-> (metrics int32 [E, 13, N] in _abi.METRIC_NAMES order, team_flag_captures int32 [E, 2], env_step_count int32 [E]):
        what utils.duel / MetricsLogger.harvest_metrics read from ``env.metrics``, for every env at once."""
        torch = _torch()
        E, N = self.n_envs, self.N_AGENTS
        met = torch.empty((E, _abi.N_METRICS, N), dtype=torch.int32, device=self.device)
        caps = torch.empty((E, 2), dtype=torch.int32, device=self.device)
        steps = torch.empty((E,), dtype=torch.int32, device=self.device)
        _abi.check(self._lib.ctf_export_counters(self._h, C.c_void_p(met.data_ptr()), C.c_void_p(caps.data_ptr()),
                                                 C.c_void_p(steps.data_ptr()), self._stream()), self._lib)
        return met, caps, steps

    def action_mask(self):
        """uint8 [N, 9]: 1 where the action is legal for the agent's type (agent_network.py:66-75)."""
        m = np.zeros((self.N_AGENTS, _abi.N_ACTIONS), np.uint8)
        _abi.check(self._lib.ctf_action_mask(self._h, m.ctypes.data_as(C.c_void_p)), self._lib)
        return m

    # -- host views ---------------------------------------------------------------------------
    def get_state(self, env_index):
        v = _abi.CtfStateView()
        _abi.check(self._lib.ctf_get_state(self._h, int(env_index), C.byref(v)), self._lib)
        return v

    def set_state(self, env_index, view):
        _abi.check(self._lib.ctf_set_state(self._h, int(env_index), C.byref(view)), self._lib)

    def status(self):
        bits = C.c_uint32(0)
        _abi.check(self._lib.ctf_status(self._h, C.byref(bits), self._stream()), self._lib)
        return bits.value


# ----------------------------------------------------------------------------------------------
# the reference's single-env API
# ----------------------------------------------------------------------------------------------
# REVERSED_ACTION_MAP (gridworld_ctf.py:147-196) written as permutations of 0..8:
#   None: both axes flipped, 0: rows flipped, 1: columns flipped, 2: anti-diagonal reflection
_REVERSED_ACTIONS = {
    None: (1, 0, 3, 2, 4, 6, 5, 8, 7),
    0: (1, 0, 2, 3, 4, 6, 5, 7, 8),
    1: (0, 1, 3, 2, 4, 5, 6, 8, 7),
    2: (2, 3, 0, 1, 4, 7, 8, 5, 6),
}
_UNIT_MOVES = ((-1, 0), (1, 0), (0, 1), (0, -1), (0, 0))


def _action_deltas():
    table = {}
    for typ, reach in ((0, 0), (1, 0), (2, 2), (3, 1)):  # scout, guardian, vaulter (jumps 2), miner (acts at 1)
        table[typ] = {a: _UNIT_MOVES[a] for a in range(5)}
        for a in range(5, 9):
            dr, dc = _UNIT_MOVES[a - 5]
            table[typ][a] = (dr * reach, dc * reach)
    return table


class GridworldCtf:
    """Drop-in for the reference class, backed by the HIP kernels (a batch of one env).

    ``rng="global"`` (default) keeps the reference's RNG contract exactly: every ``step`` consumes
    from the process-global ``random`` and ``np.random`` generators (their MT19937 states are handed
    to the device before the step and written back after it), so ``random.seed(s); np.random.seed(s)``
    in the caller has the effect it has on the reference.  ``rng="device"`` keeps private streams on
    the device (seeded with ``seed=``) and avoids the two 2.5 KB transfers per step.
    """

    def __init__(self, rng="global", seed=0, device=None, **kwargs):
        self._ctor = dict(rng=rng, seed=seed, device=device, kwargs=dict(kwargs))
        self._vec = VecGridworldCtf(1, device=device, py_seeds=[seed], np_seeds=[seed], **kwargs)
        self._rng_mode = rng
        d = self._vec.derived
        kw = d["kwargs"]
        # attributes the reference exposes (gridworld_ctf.py:58-290)
        self.AGENT_CONFIG, self.SCENARIO = kw["AGENT_CONFIG"], kw["SCENARIO"]
        self.GAME_STEPS = kw["GAME_STEPS"]
        self.ENABLE_OBSTACLES, self.DROP_FLAG_WHEN_NO_HP = kw["ENABLE_OBSTACLES"], kw["DROP_FLAG_WHEN_NO_HP"]
        self.HOME_FLAG_CAPTURE, self.USE_EASY_CAPTURE = kw["HOME_FLAG_CAPTURE"], kw["USE_EASY_CAPTURE"]
        self.USE_ADJUSTED_REWARDS, self.MAX_BLOCK_TILE_PCT = kw["USE_ADJUSTED_REWARDS"], kw["MAX_BLOCK_TILE_PCT"]
        self.LOG_METRICS, self.MAP_SYMMETRY_CHECK = kw["LOG_METRICS"], kw["MAP_SYMMETRY_CHECK"]
        self.N_AGENTS = d["n_agents"]
        self.ACTION_SPACE = 8
        self.ENV_DIMS = (1, 1, kw["GRID_SIZE"], kw["GRID_SIZE"])
        self.WIN_MARGIN_SCALAR, self.LOSS_MARGIN_SCALAR = 0.1, 0.00
        self.REWARD_CAPTURE, self.REWARD_STEP, self.REWARD_TAG = 1, 0, 0.0
        self.OPP_FLAG_CAPTURE_PUNISHMENT_SCALAR = 0.5
        self.WINNING_POINTS = np.inf
        self.DEFENSIVE_ZONE_DISTANCE = 3
        self.ACTION_DELTAS = _action_deltas()
        self.REVERSED_ACTION_MAP = {k: dict(enumerate(v)) for k, v in _REVERSED_ACTIONS.items()}
        self.AGENT_TEAMS, self.AGENT_TYPES = d["agent_teams"], d["agent_types"]
        self.AGENT_TYPE_HP, self.AGENT_HP_HEALING_PER_STEP = kw["AGENT_TYPE_HP"], kw["AGENT_HP_HEALING_PER_STEP"]
        self.AGENT_TYPE_DAMAGE = kw["AGENT_TYPE_DAMAGE"]
        self.AGENT_TYPE_ACTION_MASK = {0: 1, 1: 1, 2: 0, 3: 0}
        self.GUARDIAN_DEFENSE_DISTANCE, self.GUARDIAN_TAGGING_RANGE = 3, 1
        self.GUARDIAN_DAMAGE_MULTIPLIER, self.TAG_PROBABILITY = kw["GUARDIAN_DAMAGE_MULTIPLIER"], kw["TAG_PROBABILITY"]
        self.VAULT_HP_COST, self.VAULT_MIN_HP = kw["VAULT_HP_COST"], kw["VAULT_MIN_HP"]
        self.AGENT_FLAG_CAPTURE_TYPES = [0, 1, 2, 3]
        self.MAX_AGENT_BLOCKS = 1000
        self.OPEN_TILE, self.BLOCK_TILE, self.DESTRUCTIBLE_TILE1, self.DESTRUCTIBLE_TILE2 = 0, 1, 2, 3
        self.AGENT_TYPE_TILE_MAP = _config.AGENT_TYPE_TILE_MAP
        self.AGENT_TILE_MAP = d["agent_tile_map"]
        self.FLAG_TILE_MAP = dict(_config.FLAG_TILE_MAP)
        self.STD_OWN_FLAG_TILE, self.STD_OPP_FLAG_TILE = 12, 13
        s = kw["SCENARIO"]
        self.SCENARIO_NAME, self.FLIP_AXIS, self.GRID_SIZE = s["SCENARIO_NAME"], s["FLIP_AXIS"], s["GRID_SIZE"]
        self.FLAG_POSITIONS, self.CAPTURE_POSITIONS = s["FLAG_POSITIONS"], s["CAPTURE_POSITIONS"]
        self.SPAWN_POSITIONS, self.AGENT_STARTING_POSITIONS = s["SPAWN_POSITIONS"], s["AGENT_STARTING_POSITIONS"]
        self.OPPONENTS = d["opponents"]
        self.TILES_USED = d["tiles_used"]
        self.agent_teams_np = np.array([v for v in self.AGENT_TEAMS.values()], dtype=np.uint8)
        self.grid = np.zeros((self.GRID_SIZE, self.GRID_SIZE), dtype=np.uint8)  # live array, updated in place
        self.has_flag = np.zeros(self.N_AGENTS, dtype=np.uint8)
        self._obs_cache = None
        c, g, n = len(self.TILES_USED) + 1, self.GRID_SIZE, self.N_AGENTS
        self._obs_host = np.zeros((n, c, g, g), dtype=np.uint8)        # filled by every ctf_host_step round trip
        self._meta_host = np.zeros((n, 2 * n + 6), dtype=np.float16)
        self._rng_in = np.empty((2, 625), dtype=np.uint32)
        self._view_buf = _abi.CtfStateView()  # refilled by every round trip (the attributes below are copies taken from it)
        self._default_mask = self._default_reverse_mask()
        self._py_last = self._np_last = None  # the generator states the last step wrote back to random / np.random
        self.reset()

    # -- pickling / deepcopy (Ray hands the env to workers by value) ------------------------------
    def __getstate__(self):
        return dict(ctor=self._ctor, view=bytes(self._vec.get_state(0)), rng=self._vec.get_rng_state(0))

    def __setstate__(self, state):
        c = state["ctor"]
        self.__init__(rng=c["rng"], seed=c["seed"], device=c["device"], **c["kwargs"])
        view = _abi.CtfStateView.from_buffer_copy(state["view"])
        self._vec.set_state(0, view)
        self._vec.set_rng_state(0, *state["rng"])
        self._pull()

    def __deepcopy__(self, memo):
        new = object.__new__(type(self))
        new.__setstate__(self.__getstate__())
        return new

    # -- host mirror of the device state --------------------------------------------------------
    def _mirror(self, v):
        """The reference's attributes from a state view (the same round trip has refreshed _obs_host / _meta_host)."""
        n, g = self.N_AGENTS, self.GRID_SIZE
        self.grid[...] = np.frombuffer(v.grid, dtype=np.uint8, count=g * g).reshape(g, g)
        self.agent_positions = {i: (int(v.pos[i][0]), int(v.pos[i][1])) for i in range(n)}
        self.has_flag[...] = np.frombuffer(v.has_flag, dtype=np.uint8, count=n)
        self.agent_hp = {i: v.hp[i] for i in range(n)}
        self.block_inventory = {i: int(v.inventory[i]) for i in range(n)}
        self._arr = [int(v.perm[i]) for i in range(n)]
        self.env_step_count = int(v.step_count)
        self.done = bool(v.done)
        self._view = v
        self._obs_cache = None  # (renders under a non-default reversal; the default one is _obs_host / _meta_host, always current)

    def _pull(self):
        """State view + the default observation of the env as it is now, in one round trip (no step)."""
        _, _, _, v, _, _ = self._vec.host_step(None, view=self._view_buf, obs=self._obs_host, meta=self._meta_host)
        self._mirror(v)

    @property
    def metrics(self):
        """The reference's metrics dict (gridworld_ctf.py:425-470), rebuilt from the device counters:
        team_* and agent_type_* entries are sums of the agent-level counters, as in the reference."""
        v, n, g = self._view, self.N_AGENTS, self.GRID_SIZE
        out = {"team_wins": {0: 0, 1: 0}}
        for k, name in enumerate(_abi.METRIC_NAMES):
            team = {0: 0, 1: 0}
            by_type = defaultdict(partial(defaultdict, int))
            agent = defaultdict(int)
            for i in range(n):
                val = int(v.metrics[k][i])
                if val:
                    agent[i] = val
                    team[self.AGENT_TEAMS[i]] += val
                    by_type[self.AGENT_TEAMS[i]][self.AGENT_TYPES[i]] += val
            out["team_" + name], out["agent_type_" + name], out["agent_" + name] = team, by_type, agent
        out["team_flag_captures"] = {0: int(v.team_captures[0]), 1: int(v.team_captures[1])}
        vis = defaultdict(partial(np.zeros, (g, g), dtype=np.uint8))
        for i in range(n):
            vis[i] = np.frombuffer(v.visitation[i], dtype=np.uint8, count=g * g).reshape(g, g).copy()
        out["agent_visitation_maps"] = vis
        return out

    # -- reference API ----------------------------------------------------------------------------
    def reset(self):
        self._vec.reset()
        self._pull()
        if self.MAP_SYMMETRY_CHECK:
            assert np.all(self.standardise_state(0) == self.standardise_state(1, reverse_grid=True))

    def _global_rng_in(self):
        """random / np.random -> the two uint32 [625] arrays (624 words + position) ctf_host_step installs before the step.
        -> (np.random's state tuple, unchanged): True when both generators are exactly where the previous step of THIS env left
        them — the device streams are then already in place and nothing needs to go over."""
        st = np.random.get_state()
        pt = _py_random.getstate()[1]
        if self._py_last is not None and pt == self._py_last and st[2] == self._np_last[1] and np.array_equal(st[1], self._np_last[0]):
            return st, True
        both = self._rng_in
        both[0] = pt
        both[1, :624] = st[1]
        both[1, 624] = st[2]
        return st, False

    def step(self, actions):
        acts = [actions[i] for i in range(self.N_AGENTS)]
        for i, a in enumerate(acts):
            if not (isinstance(a, (int, np.integer)) and 0 <= int(a) <= 8):
                raise KeyError(a)  # ACTION_DELTAS[type][action] in the reference
        glob = self._rng_mode == "global"
        st, in_place = self._global_rng_in() if glob else (None, True)
        # ONE round trip (ctf_host_step): generator states in, step, render of all N agents, state view and generator states out
        r64, _, status, v, py, npw = self._vec.host_step(np.array(acts, dtype=np.int8), None if in_place else self._rng_in[0],
                                                         None if in_place else self._rng_in[1], rng_out=glob, view=self._view_buf,
                                                         obs=self._obs_host, meta=self._meta_host)
        if glob:
            self._py_last = tuple(py.tolist())
            self._np_last = (npw[:624], int(npw[624]))
            _py_random.setstate((3, self._py_last, None))
            np.random.set_state((st[0], npw[:624], int(npw[624]), st[3], st[4]))
        self._mirror(v)
        if status & _abi.ST_NO_RESPAWN:
            raise ValueError("high <= 0")  # np.random.randint(0) in the reference's respawn
        if status & _abi.ST_SPAWN_EDGE:
            raise IndexError("respawn window clipped at row/col 0: the reference misplaces the agent here")
        return self.grid, r64.tolist(), self.done

    def _observe(self, reverse_mask):
        if reverse_mask == self._default_mask:
            return reverse_mask, self._obs_host, self._meta_host
        if self._obs_cache is None or self._obs_cache[0] != reverse_mask:  # a non-default reversal: its own render
            obs, meta = self._vec.observe(reverse_mask)
            self._obs_cache = (reverse_mask, obs[0].cpu().numpy(), meta[0].cpu().numpy())
        return self._obs_cache

    def _default_reverse_mask(self):
        return sum(1 << i for i in range(self.N_AGENTS) if self.AGENT_TEAMS[i] == 1)

    def standardise_state(self, agent_idx, reverse_grid=False):
        i = int(agent_idx)
        mask = self._default_mask
        if bool(reverse_grid) != bool((mask >> i) & 1):
            mask ^= 1 << i
        return self._observe(mask)[1][i][None].copy()

    def get_env_metadata(self, agent_idx):
        for t in self.AGENT_TYPES.values():
            if t not in self.agent_hp:
                raise KeyError(t)  # the reference indexes agent_hp by type id (gridworld_ctf.py:1041)
        return self._meta_host[int(agent_idx)][None].copy()  # (the metadata rows do not depend on the reversal)

    def get_env_dims(self):
        c, g, n = len(self.TILES_USED) + 1, self.GRID_SIZE, self.N_AGENTS
        return (c, g, g), (c - 1, g, g), (n * 2 + 6,), (n * 6 + n * self.ACTION_SPACE + 3,)

    def get_reversed_action(self, action):
        return self.REVERSED_ACTION_MAP[self.FLIP_AXIS][action]

    def get_tiles_used(self):
        return list(self.TILES_USED)

    def max_dim_distance_to_xy(self, xy, target_xy):
        return max(abs(xy[0] - target_xy[0]), abs(xy[1] - target_xy[1]))

    def agent_distance_to_xy(self, agent_idx, object_xy):
        return self.max_dim_distance_to_xy(self.agent_positions[agent_idx], object_xy)

    def render(self, sleep_time=0.2, ego_state_agent=None):
        """Minimal matplotlib view of the grid (presentation is outside the accelerated path)."""
        import matplotlib.pyplot as plt

        plt.figure(num="env_render")
        plt.clf()
        plt.imshow(self.grid, vmin=0, vmax=13, cmap="tab20")
        plt.title(f"step {self.env_step_count}")
        plt.pause(sleep_time)

    @staticmethod
    def render_indices(grid, agent_positions, has_flag, agent_teams, flag_positions):
        """The sprite index of every cell as the reference's ``render_image`` chooses it (gridworld_ctf.py:1130-1154): the
        tile code; 112 / 113 at a team's home flag cell while the OTHER team carries that flag; + 100 for an agent that
        carries a flag.  Pure function of the host-side state (tested without a GPU)."""
        idx = np.array(grid, dtype=np.int32)
        teams_with_flag = {0: 0, 1: 0}
        for a, flag in enumerate(has_flag):
            if flag == 1:
                teams_with_flag[agent_teams[a]] = 1
        if teams_with_flag[1] == 1:
            idx[tuple(flag_positions[0])] = 112
        if teams_with_flag[0] == 1:
            idx[tuple(flag_positions[1])] = 113
        for a, pos in agent_positions.items():
            if has_flag[a] == 1:
                idx[tuple(pos)] += 100
        return idx

    # one colour per sprite index (the reference pastes PNG sprites from cwd/img; this build draws squares and letters, so
    # that ``utils.create_gif`` / ``duel(render=True)`` work without those files)
    _SPRITE_RGB = {0: (0.93, 0.93, 0.93), 1: (0.25, 0.25, 0.25), 2: (0.55, 0.45, 0.35), 3: (0.72, 0.62, 0.52),
                   12: (0.35, 0.55, 1.0), 13: (1.0, 0.4, 0.4), 112: (0.8, 0.85, 1.0), 113: (1.0, 0.85, 0.85)}
    _TYPE_LETTER = "SGVM"  # scout, guardian, vaulter, miner

    def render_image(self, frame_path=None, plot_image=False):
        """gridworld_ctf.py:1112-1163: one image of the current grid; saved to ``frame_path`` (dpi 300, then closed) and / or
        shown.  Same cell -> sprite choice as the reference (``render_indices``); drawn with matplotlib primitives."""
        import matplotlib.pyplot as plt

        idx = self.render_indices(self.grid, self.agent_positions, self.has_flag, self.AGENT_TEAMS, self.FLAG_POSITIONS)
        g = self.GRID_SIZE
        rgb = np.zeros((g, g, 3))
        fig, ax = plt.subplots(figsize=(5, 5))
        for i in range(g):
            for j in range(g):
                k = int(idx[i, j])
                base = k - 100 if (k >= 104 and k <= 111) else k
                if 4 <= base <= 11:  # an agent: team colour, type letter, a ring when it carries a flag
                    team = 0 if base < 8 else 1
                    rgb[i, j] = (0.2, 0.4, 0.9) if team == 0 else (0.9, 0.25, 0.25)
                    ax.text(j, i, self._TYPE_LETTER[(base - 4) % 4] + ("*" if k >= 100 else ""), ha="center", va="center",
                            color="white", fontsize=max(4, 110 // g), fontweight="bold")
                else:
                    rgb[i, j] = self._SPRITE_RGB.get(k, (1.0, 0.0, 1.0))
                    if k in (12, 13, 112, 113):
                        ax.text(j, i, "F" if k < 100 else "f", ha="center", va="center", color="black", fontsize=max(4, 110 // g))
        ax.imshow(rgb, interpolation="nearest")
        ax.set_xticks(np.arange(-0.5, g, 1))
        ax.set_yticks(np.arange(-0.5, g, 1))
        ax.set_xticklabels([])
        ax.set_yticklabels([])
        ax.grid(color="white", linewidth=1)
        ax.tick_params(length=0)
        if plot_image:
            plt.show()
        if frame_path is not None:
            fig.savefig(frame_path, dpi=300)
            plt.close(fig)

    def play(self, player=0, agents=None, use_ego_state=False, device="cpu", render_ego_state=False):
        """gridworld_ctf.py:1165-1262: step the env from the keyboard (w s d a x = actions 0..4, t g h f = 5..8, p = quit);
        the other agents act randomly through ``np.random.randint(8, size=N)`` as in the reference.  ``agents`` (the
        reference's path through ``choose_action`` / ``ut.add_noise``, which its own networks do not provide) is not supported."""
        if agents is not None:
            raise NotImplementedError("play(agents=...) relies on Agent.choose_action, which the reference's networks do not define")
        keys = {"w": 0, "s": 1, "d": 2, "a": 3, "x": 4, "t": 5, "g": 6, "h": 7, "f": 8}
        self.reset()
        self.render()
        move_counter, total_score = 0, 0
        while True:
            print(f"Move {move_counter}")
            raw = None
            while raw not in list(keys) + ["p"]:
                raw = input("Enter an action")
            if raw == "p":
                print("Game exited")
                break
            actions = np.random.randint(8, size=self.N_AGENTS)
            actions[player] = keys[raw]
            _, rewards, done = self.step([int(a) for a in actions])
            total_score += rewards[0]
            move_counter += 1
            self.render()
            if done:
                print(f"You win!, total score {total_score}")
                break
