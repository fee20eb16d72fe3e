"""Golden duels: the reference's own ``utils.duel`` (utils.py:500-573) on the reference env with the deterministic
stub policies; records the result sign and the final ``env.metrics`` counters.

Build container only:  python tests/golden/make_golden_duel.py
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402
from _stub_policy import StubDuelPolicy  # noqa: E402


def main():
    Ref, scn = _refimport.import_reference()
    ref_utils = _refimport.import_reference.modules["utils"]
    edge_kw = dict(AGENT_CONFIG={0: {"team": 0, "type": 0}, 1: {"team": 1, "type": 0}, 2: {"team": 0, "type": 1}, 3: {"team": 1, "type": 2}},
                   GAME_STEPS=120, MAP_SYMMETRY_CHECK=False, TAG_PROBABILITY=1.0, AGENT_TYPE_HP={0: 1, 1: 2, 2: 1, 3: 1},
                   USE_ADJUSTED_REWARDS=True)
    jobs = [("duel_arena", "arena_iii", mg.ARENA_KW, 40, 91, 0), ("duel_split_full", "arrow", dict(mg.SPLIT_KW, GAME_STEPS=90), 256, 92, 0)]
    for seed in range(100, 400):
        jobs.append(("duel_edge_result", mg.SYN_EDGE, edge_kw, 256, seed, seed))
    got_edge = False
    for name, scen_key, kwargs, max_steps, seed, salt in jobs:
        if name == "duel_edge_result" and got_edge:
            continue
        random.seed(seed)
        np.random.seed(seed)
        try:
            env = Ref(SCENARIO=getattr(scn, scen_key) if isinstance(scen_key, str) else scen_key, **kwargs)
            a, b, result = ref_utils.duel(env, StubDuelPolicy(3 + salt), StubDuelPolicy(5 + salt), (7, 9), return_result=True, max_steps=max_steps)
        except ValueError:
            continue
        if name == "duel_edge_result":
            if result == 0:
                continue
            got_edge = True
        n = env.N_AGENTS
        metrics = np.zeros((len(mg.METRIC_NAMES), n), np.int32)
        for k, mname in enumerate(mg.METRIC_NAMES):
            for i in range(n):
                metrics[k, i] = env.metrics["agent_" + mname].get(i, 0)
        caps = [int(env.metrics["team_flag_captures"][0]), int(env.metrics["team_flag_captures"][1])]
        meta = dict(name=name, scenario=mg.jsonable_scenario(scen_key), kwargs=mg.jsonable_kwargs(kwargs), seed=seed, salt=salt,
                    max_steps=max_steps, result=int(result), idxs=[int(a), int(b)], captures=caps, steps=int(env.env_step_count))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), metrics=metrics,
                            case_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        print(name, "result", result, "captures", caps, "steps", env.env_step_count, "tags", metrics[0].sum())


if __name__ == "__main__":
    main()
