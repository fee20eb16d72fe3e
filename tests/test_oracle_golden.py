"""The CPU oracle against the golden vectors recorded from the reference itself (bit-exact)."""
import numpy as np
import pytest

import oracle
from _cases import Case, abi, case_names, cfgmod, spawn_edge_kwargs, view_arrays


@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_reference_trajectory(name):
    case = Case(name)
    z = case.z
    cfg, derived = case.config()
    assert derived["tiles_used"] == case.meta["tiles_used"]
    assert (cfg.n_agents, cfg.n_channels, cfg.grid_size, cfg.grid_size) == case.obs_shape
    env = oracle.OracleEnv(cfg)
    env.seed(case.meta["seed"], case.meta["seed"])
    py0, np0 = case.seeded_states()
    got_py, got_np = env.get_rng_state()
    assert np.array_equal(got_py, py0) and np.array_equal(got_np, np0), "seeding differs from random.seed / np.random.seed"

    obs, meta = env.observe()
    assert np.array_equal(obs, case.unpack_obs(z["obs0"]))
    assert np.array_equal(meta.view(np.uint16), z["meta0"])

    extra = {int(t): k for k, t in enumerate(z["extra_steps"])}
    n, g = case.n, case.g
    for t in range(case.T):
        if t in case.reset_at:
            env.reset()
        rewards, done, status = env.step(z["actions"][t])
        assert status == 0
        s = view_arrays(env.get_state(), n, g)
        ctx = f"{name} step {t}"
        assert np.array_equal(s["grid"], z["grid"][t]), ctx
        assert np.array_equal(s["pos"], z["pos"][t]), ctx
        assert np.array_equal(s["hp"], z["hp"][t]), ctx  # float64, bit-exact
        assert np.array_equal(s["has_flag"], z["has_flag"][t]), ctx
        assert np.array_equal(s["inv"], z["inv"][t]), ctx
        assert np.array_equal(s["perm"], z["perm"][t]), ctx
        assert np.array_equal(rewards, z["rewards"][t]), ctx
        assert int(done) == int(z["done"][t]), ctx
        py, npw = env.get_rng_state()
        assert (int(py[624]), int(npw[624])) == (int(z["py_pos"][t]), int(z["np_pos"][t])), ctx + " (draw counts)"
        obs, meta = env.observe()
        assert np.array_equal(obs, case.unpack_obs(z["obs"][t])), ctx
        assert np.array_equal(meta.view(np.uint16), z["meta"][t]), ctx
        if t in extra:
            o_unrev, _ = env.observe(reverse_mask=0)
            o_rev, _ = env.observe(reverse_mask=(1 << n) - 1)
            assert np.array_equal(o_unrev, case.unpack_obs(z["extra_obs_unrev"][extra[t]])), ctx
            assert np.array_equal(o_rev, case.unpack_obs(z["extra_obs_rev"][extra[t]])), ctx

    s = view_arrays(env.get_state(), n, g)
    assert np.array_equal(s["metrics"], z["metrics"])
    assert np.array_equal(s["visitation"], z["visitation"])
    assert s["team_captures"] == case.meta["team_captures"]
    py, npw = env.get_rng_state()
    assert np.array_equal(py, z["py_state"]) and np.array_equal(npw, z["np_state"])


def test_f64_to_f16_matches_numpy():
    rng = np.random.default_rng(0)
    xs = np.concatenate([
        rng.random(2000), rng.random(2000) * 1e-4, rng.random(2000) * 1e-7, rng.random(2000) * 70000,
        -rng.random(500) * 3, np.array([0.0, -0.0, 1.0, 65504.0, 65519.9, 65520.0, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001,
                                        5.9604644775390625e-08, 6.097555160522461e-05, np.inf, -np.inf, 1 / 3, 501.0, 1 / 501]),
        np.arange(0, 501) / 500.0,
    ])
    want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle.f64_to_f16_bits(x) for x in xs], np.uint16)
    assert np.array_equal(got, want)


def test_a_respawn_offset_that_goes_negative_raises_the_spawn_edge_bit():
    """The one place where this build deliberately does NOT follow the reference ("differences"): a spawn position
    on row 0 / column 0 whose respawn draws an offset of 0 makes the reference store the coordinate -1 (gridworld_ctf.py:775-785; its
    own "WARNING" at :773) and carry on with NumPy's negative-index wrap.  Here the step reports CTF_ST_SPAWN_EDGE (the facade raises
    IndexError) and the agent sits on the wrapped cell G - 1.  Respawns on such a map whose "- 1" stays >= 0 (the agent lands one cell
    up / left of the open cell drawn) DO follow the reference: tests/golden/fuzz_edge0_*.npz."""
    kw = spawn_edge_kwargs()
    cfg, _ = cfgmod.build_config(kw, log_metrics=True)
    env = oracle.OracleEnv(cfg)
    env.seed(5, 5)
    rewards, done, status = env.step(np.array([4, 4], np.int8))
    assert status == abi.ST_SPAWN_EDGE
    s = view_arrays(env.get_state(), 2, 7)
    respawned = 0 if s["pos"][0].tolist() != [3, 3] else 1
    assert s["metrics"][abi.METRIC_NAMES.index("respawn_tag_count")].sum() >= 1
    r, c = s["pos"][respawned]
    assert (r == 6 and respawned == 0) or (c == 6 and respawned == 1)  # the wrapped row / column
    assert s["grid"][r, c] == 4 + 4 * respawned  # the agent's tile is where NumPy's wrap would have written it
