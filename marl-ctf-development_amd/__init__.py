"""Batched GridworldCtf step / reset / observation path on MI355X (gfx950).

A from-scratch, MI355X-native implementation of ONE hot path of g-nightingale/marl-ctf-development:
``GridworldCtf.step()/.reset()`` plus the per-agent observation / metadata / action-mask outputs, run
for thousands of independent envs at once by hand-written HIP kernels behind a C ABI
(include/ctf_env.h).  See DESIGN.md.

The directory name carries a hyphen, so import it with importlib::

    ctf = importlib.import_module("marl-ctf-development_amd")

or, for drop-in use with the reference's own scripts, put this directory first on sys.path so that
``from gridworld_ctf import GridworldCtf`` resolves here.
"""
from . import _abi, config, configs, sharding  # noqa: F401
from .gridworld_ctf import GridworldCtf, VecGridworldCtf, expand_codes  # noqa: F401
from .maps import CtfScenarios  # noqa: F401
from .rollout import BatchedRolloutCollector  # noqa: F401
from .duel import batched_duel  # noqa: F401


def __getattr__(name):  # the policy module needs torch.nn: import it only when asked for
    if name in ("policy", "policy_native", "learner"):
        import importlib

        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
