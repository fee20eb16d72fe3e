"""Counter-based RNG mode (include/ctf_env.h CTF_RNG_COUNTER) pinned to the reference: tests/golden/counter_*.npz were recorded
from the reference env itself with random.shuffle / np.random.rand / np.random.randint patched to read Philox tapes
(tests/golden/make_golden_counter.py); the oracle in counter mode — the same three algorithms on the same tape —
must reproduce them step by step, words consumed included.  The -m gpu tests then hold the HIP path to the oracle."""
import glob
import json
import os

import numpy as np
import pytest

import oracle
from _cases import GOLDEN, abi, cfgmod, kwargs_from_json, view_arrays


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "counter_*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_counter_mode_replays_the_patched_reference(path):
    z = np.load(path)
    meta = json.loads(bytes(z["case_json"]).decode())
    kw = kwargs_from_json(meta)
    cfg, _ = cfgmod.build_config(kw, log_metrics=True, rng_mode=abi.RNG_COUNTER)
    env = oracle.OracleEnv(cfg)
    env.seed(meta["py_seed"], meta["np_seed"])
    n, g = meta["n"], meta["g"]
    for t in range(meta["T"]):
        if env.get_state().done:
            env.reset()
        rewards, done, status = env.step(z["actions"][t])
        assert status == 0
        ctx = f"{meta['name']} step {t}"
        assert np.array_equal(rewards, z["rewards"][t]), ctx
        assert int(done) == int(z["done"][t]), ctx
        s = view_arrays(env.get_state(), n, g)
        for k in ("grid", "pos", "hp", "has_flag", "inv", "perm"):
            assert np.array_equal(s[k], z[k][t]), f"{ctx}: {k}"
        assert env.get_rng_counters() == (int(z["py_n"][t]), int(z["np_n"][t])), ctx + ": words consumed"
