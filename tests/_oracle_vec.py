"""TEST INFRASTRUCTURE: a stand-in for ``VecGridworldCtf`` of ONE env backed by the CPU oracle, so that the Python layer of the drop-in
facade (``marl-ctf-development_amd.GridworldCtf``: attribute mirror, global-RNG contract, observation cache, exceptions, pickling) can
be driven in the build container — which has the reference but no GPU — by the REFERENCE'S OWN callers (tests/test_reference_callers_cpu.py).
It is monkeypatched over the class in the test only; the product never imports it (nor oracle/)."""
import importlib

import numpy as np

import oracle

abi = importlib.import_module("marl-ctf-development_amd._abi")
cfgmod = importlib.import_module("marl-ctf-development_amd.config")


class OracleVec:
    def __init__(self, n_envs, device=None, py_seeds=None, np_seeds=None, log_metrics=True, **env_kwargs):
        assert n_envs == 1
        self.cfg, self.derived = cfgmod.build_config(env_kwargs, log_metrics=log_metrics)
        d = self.derived
        self.n_envs, self.device = 1, "cpu"
        self.N_AGENTS, self.GRID_SIZE = d["n_agents"], d["grid_size"]
        self.N_CHANNELS, self.META_LEN = self.cfg.n_channels, 2 * d["n_agents"] + 6
        self.AGENT_TEAMS, self.AGENT_TYPES, self.TILES_USED = d["agent_teams"], d["agent_types"], d["tiles_used"]
        self._env = oracle.OracleEnv(self.cfg)
        self._env.seed(int((py_seeds or [0])[0]), int((np_seeds or py_seeds or [0])[0]))
        self._status = 0

    # what the facade calls ------------------------------------------------------------------------------------------------------
    def host_step(self, actions=None, py_in=None, np_in=None, reverse_mask=None, rng_out=False, view=None, obs=None, meta=None):
        e = self._env
        if py_in is not None or np_in is not None:
            e.set_rng_state(py_in, np_in)
        rewards, status = np.zeros(self.N_AGENTS, np.float64), 0
        if actions is not None:
            rewards, _, status = e.step(actions)
        o, m = e.observe(abi.REVERSE_DEFAULT if reverse_mask is None else int(reverse_mask))
        if obs is not None:
            obs[...] = o
        if meta is not None:
            meta[...] = m
        py_out, np_out = e.get_rng_state() if rng_out else (None, None)
        v = e.get_state()
        return rewards, bool(v.done), status, v, py_out, np_out

    def reset(self, mask=None):
        self._env.reset()

    def observe(self, reverse_mask=None, obs=True, meta=True):
        import torch

        o, m = self._env.observe(abi.REVERSE_DEFAULT if reverse_mask is None else int(reverse_mask))
        return torch.from_numpy(o[None].copy()), torch.from_numpy(m[None].copy())

    def get_state(self, env_index):
        return self._env.get_state()

    def set_state(self, env_index, view):
        self._env.set_state(view)

    def get_rng_state(self, env_index):
        return self._env.get_rng_state()

    def set_rng_state(self, env_index, py_mt625=None, np_mt625=None):
        self._env.set_rng_state(py_mt625, np_mt625)

    def close(self):
        pass
