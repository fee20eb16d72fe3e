"""Golden rollouts: runs the reference's own ``PPOTrainer.get_single_rollout`` (ppo.py:31-131) on the reference env
with the deterministic stub policies of tests/_stub_policy.py and records every tensor it returns.

Build container only:  python tests/golden/make_golden_rollout.py
"""
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402
from _stub_policy import StubPolicy  # noqa: E402


def main():
    import torch

    Ref, scn = _refimport.import_reference()
    mods = _refimport.import_reference.modules
    saved = list(sys.path)
    sys.path.insert(0, _refimport.REFERENCE_DIR)
    sys.modules.update(mods)
    try:
        import ppo as ref_ppo
    finally:
        sys.path[:] = saved
        for m in ("gridworld_ctf", "scenarios", "utils", "ppo", "agent_network"):
            sys.modules.pop(m, None)

    edge_kw = dict(AGENT_CONFIG={0: {"team": 0, "type": 0}, 1: {"team": 1, "type": 0}, 2: {"team": 0, "type": 1}, 3: {"team": 1, "type": 2}},
                   GAME_STEPS=60, MAP_SYMMETRY_CHECK=False, TAG_PROBABILITY=1.0, AGENT_TYPE_HP={0: 1, 1: 2, 2: 1, 3: 1},
                   USE_ADJUSTED_REWARDS=True)
    jobs = [
        ("rollout_arena_team0", "arena_iii", mg.ARENA_KW, 0, 24, 71),
        ("rollout_arena_team1", "arena_iii", mg.ARENA_KW, 1, 24, 72),
        ("rollout_split_team1", "arrow", mg.SPLIT_KW, 1, 40, 73),
    ]
    # a small synthetic map where flags get captured: non-zero, agent-distinct rewards (incl. the terminal margin at step 60)
    for seed in range(80, 400):
        jobs.append(("rollout_edge_rewards", mg.SYN_EDGE, edge_kw, seed % 2, 60, seed))
    have_rewards = False
    for name, scen_key, kwargs, team, steps, seed in jobs:
        if name == "rollout_edge_rewards" and have_rewards:
            continue
        random.seed(seed)
        np.random.seed(seed)
        try:
            env = Ref(SCENARIO=getattr(scn, scen_key) if isinstance(scen_key, str) else scen_key, **kwargs)
        except ValueError:
            continue
        dims = env.get_env_dims()
        args = types.SimpleNamespace(num_steps=steps, device="cpu")
        tr = ref_ppo.PPOTrainer(args, dims[0], dims[2])
        # what train_ppo sets up before it collects rollouts (ppo.py:273-288)
        tr.device = "cpu"
        tr.team_to_train = team
        tr.reverse_grid = team == 1
        tr.num_agents_per_team = env.N_AGENTS // 2
        tr.num_steps = steps * tr.num_agents_per_team
        tr.max_rewards = -np.inf
        try:
            out = tr.get_single_rollout(env, StubPolicy(3 + seed), StubPolicy(5 + seed))
        except ValueError:  # the one-open-cell spawn of SYN_EDGE was occupied: try the next seed
            continue
        keys = ("grid_states", "metadata_states", "actions", "use_action_mask", "logprobs", "rewards", "dones", "values",
                "next_grid_state", "next_metadata_state", "next_done")
        arrays = {k: v.numpy() for k, v in zip(keys, out)}
        if name == "rollout_edge_rewards":
            if len(np.unique(arrays["rewards"])) < 3:
                continue
            have_rewards = True
        arrays["grid_states"] = np.packbits(arrays["grid_states"].astype(np.uint8).reshape(-1))
        arrays["next_grid_state"] = arrays["next_grid_state"].astype(np.uint8)
        meta = dict(name=name, scenario=mg.jsonable_scenario(scen_key), kwargs=mg.jsonable_kwargs(kwargs), seed=seed, team=team, steps=steps,
                    grid_shape=[tr.num_steps] + list(dims[0]))
        arrays["case_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(name, os.path.getsize(path) // 1024, "KiB", "reward sum", float(arrays["rewards"].sum()), "actions", arrays["actions"][:8])


if __name__ == "__main__":
    main()
