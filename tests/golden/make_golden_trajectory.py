"""Golden trajectory export: the reference's ``utils.duel_json`` (utils.py:728-815) with the stub policies.
Build container only:  python tests/golden/make_golden_trajectory.py"""
import json
import os
import random
import sys
import tempfile

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
    seed = 55
    kwargs = dict(mg.ARENA_KW, GAME_STEPS=80, TAG_PROBABILITY=1.0, AGENT_TYPE_HP={0: 2, 1: 2, 2: 4, 3: 1.5}, VAULT_HP_COST=0.5, VAULT_MIN_HP=1.0)
    random.seed(seed)
    np.random.seed(seed)
    env = Ref(SCENARIO=scn.arena_iii, **kwargs)
    with tempfile.TemporaryDirectory() as tmp:
        rec = ref_utils.duel_json(env, StubDuelPolicy(4), StubDuelPolicy(6), max_steps=256, fname=os.path.join(tmp, "o.json"))
        text = open(os.path.join(tmp, "o.json")).read()
    rec = json.loads(text)  # through the reference's own converter: plain ints
    meta = dict(name="trajectory_arena", scenario="arena_iii", kwargs=mg.jsonable_kwargs(kwargs), seed=seed, salts=[4, 6], max_steps=256)
    out = os.path.join(HERE, "trajectory_arena.json")
    import gzip

    with gzip.open(out + ".gz", "wt") as f:
        json.dump({"case": meta, "record": rec}, f, separators=(",", ":"))
    moved = sum(1 for st in rec["movement"] for m in st if m["x"] or m["z"])
    print("steps", len(rec["movement"]), "moves", moved, "final score", rec["scores"][-1], os.path.getsize(out + ".gz"), "bytes")


if __name__ == "__main__":
    main()
