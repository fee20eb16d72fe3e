"""Differential test on randomly generated configurations (map, team sizes, types, flip axis, rule switches, odd batch
sizes, the largest supported grid and agent count): HIP path vs the CPU oracle, bit-exact, a few dozen steps each."""
import numpy as np
import pytest

import oracle
from _cases import cfgmod, pkg, view_arrays

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def random_scenario(rng, g, n):
    grid = np.zeros((g, g), np.uint8)
    interior = [(r, c) for r in range(1, g - 1) for c in range(1, g - 1)]
    rng.shuffle(interior)
    spawns = [tuple(int(x) for x in interior[0]), tuple(int(x) for x in interior[1])]  # never clipped at row/col 0
    free = [(r, c) for r in range(g) for c in range(g) if (r, c) not in spawns]
    rng.shuffle(free)
    picks = [tuple(int(x) for x in p) for p in free]
    flags = [picks.pop(), picks.pop()]
    starts = [picks.pop() for _ in range(n)]
    taken = set(flags + starts)
    blocks, destr = [], []
    for p in picks[: max(2, g * g // 12)]:
        # keep every spawn window mostly open so that respawns always find a cell
        if any(max(abs(p[0] - s[0]), abs(p[1] - s[1])) <= 1 for s in spawns) or p in taken:
            continue
        (blocks if rng.random() < 0.5 else destr).append(p)
    return {
        "SCENARIO_NAME": "rand", "GRID_SIZE": g, "FLIP_AXIS": [None, 0, 1, 2][int(rng.integers(4))],
        "FLAG_POSITIONS": dict(enumerate(flags)), "CAPTURE_POSITIONS": dict(enumerate(flags)),
        "SPAWN_POSITIONS": dict(enumerate(spawns)), "AGENT_STARTING_POSITIONS": dict(enumerate(starts)),
        "BLOCK_TILE_SLICES": blocks, "DESTRUCTIBLE_TILE_SLICES": destr,
    }


CASES = [(4, 2, 37), (5, 4, 65), (7, 6, 33), (9, 8, 130), (16, 10, 20), (17, 12, 9), (23, 14, 5), (32, 16, 6), (32, 2, 3), (8, 16, 64)]


@pytest.mark.parametrize("g,n,n_envs", CASES)
def test_random_config_matches_oracle(g, n, n_envs):
    rng = np.random.default_rng(1000 * g + n)
    scen = random_scenario(rng, g, n)
    kw = dict(
        SCENARIO=scen, AGENT_CONFIG={i: {"team": i % 2, "type": int(rng.integers(4))} for i in range(n)},
        GAME_STEPS=int(rng.integers(20, 45)), MAP_SYMMETRY_CHECK=False, USE_ADJUSTED_REWARDS=bool(rng.integers(2)),
        HOME_FLAG_CAPTURE=bool(rng.integers(2)), DROP_FLAG_WHEN_NO_HP=bool(rng.integers(2)),
        TAG_PROBABILITY=float(rng.choice([0.5, 0.75, 1.0])), AGENT_TYPE_HP={0: 2, 1: 3, 2: 2.5, 3: 1.5},
        AGENT_TYPE_DAMAGE={0: 1, 1: 0.5, 2: 0.75, 3: 1}, VAULT_HP_COST=0.5, VAULT_MIN_HP=0.75,
        AGENT_HP_HEALING_PER_STEP=float(rng.choice([0.25, 0.1])),
    )
    if n == 2:  # get_env_metadata indexes agent_hp by TYPE id (gridworld_ctf.py:1041): types must be < N
        kw["AGENT_CONFIG"] = {0: {"team": 0, "type": 1}, 1: {"team": 1, "type": 0}}
    log_metrics = bool(g % 2)
    cfg, _ = cfgmod.build_config(kw, log_metrics=log_metrics)
    seeds = np.arange(n_envs, dtype=np.uint64) * 31 + 7
    vec = pkg.VecGridworldCtf(n_envs, device=0, py_seeds=seeds, np_seeds=seeds, log_metrics=log_metrics, **kw)
    refs = [oracle.OracleEnv(cfg) for _ in range(n_envs)]
    for e, r in enumerate(refs):
        r.seed(int(seeds[e]), int(seeds[e]))
    acts = torch.empty((n_envs, n), dtype=torch.int8, device=vec.device)
    alive = np.ones(n_envs, bool)
    steps = 70
    for t in range(steps):
        vec.random_actions(acts, seed=99, step=t)
        mask = None if t % 16 else (1 << n) - 1 if t % 32 else 0  # non-default reversal flags now and then
        rewards, done = vec.step(acts, auto_reset=True, want_f64=True)
        obs, meta = vec.observe(reverse_mask=mask)
        if t % 5 == 0:  # the compact observation carries the same planes (dword and byte store paths, every grid size)
            codes, meta_c = vec.observe_codes(reverse_mask=mask)
            assert torch.equal(pkg.expand_codes(codes, vec.N_CHANNELS), obs) and torch.equal(meta_c, meta), f"G={g} N={n} step {t}"
        a, r64, d = acts.cpu().numpy(), vec.rewards64.cpu().numpy(), done.cpu().numpy()
        o, m = obs.cpu().numpy(), meta.cpu().numpy().view(np.uint16)
        for e, r in enumerate(refs):
            if not alive[e]:
                continue
            if r.get_state().done:
                r.reset()
            rw, dn, status = r.step(a[e])
            if status:
                alive[e] = False
                continue
            ro, rm = r.observe() if mask is None else r.observe(reverse_mask=mask)
            ctx = f"G={g} N={n} env {e} step {t}"
            assert np.array_equal(r64[e], rw), ctx
            assert int(d[e]) == int(dn), ctx
            assert np.array_equal(o[e], ro), ctx
            assert np.array_equal(m[e], rm.view(np.uint16)), ctx
    assert alive.sum() >= max(1, n_envs // 4)  # crowded random maps can run out of respawn cells (the reference raises there)
    for e in range(n_envs):
        if alive[e]:
            a_, b_ = view_arrays(vec.get_state(e), n, g), view_arrays(refs[e].get_state(), n, g)
            for k in ("grid", "pos", "hp", "has_flag", "inv", "perm", "metrics", "visitation"):
                assert np.array_equal(a_[k], b_[k]), f"G={g} N={n} env {e} final {k}"
            break
    vec.close()


@pytest.mark.parametrize("block", range(8))
def test_wild_configs_match_the_oracle(block):
    """Configurations outside what the shipped maps and scripts use (tests/_cases.py wild_config: unequal teams down to 1 v N-1, capture
    cells away from the flags, types that deal no damage, TAG_PROBABILITY 0, healing past the cap, free vaults, 4x4 grids with eight
    agents) — the same generator the build-container test runs against the LIVE reference (tests/test_oracle_vs_live_reference.py):
    HIP path vs oracle, 48 envs each, every step rewards (f64) / done / all observations and metadata rows, at the end the full state
    views and both generators' states of every env that never ran out of respawn cells."""
    from _cases import abi, wild_config

    rng = np.random.default_rng(888_000 + block)
    done_cfgs = 0
    for trial in range(6):
        drawn = wild_config(rng)
        if drawn is None:
            continue
        scen, kw = drawn
        kw = dict(kw, SCENARIO=scen)
        n, g = len(kw["AGENT_CONFIG"]), scen["GRID_SIZE"]
        cfg, _ = cfgmod.build_config(kw, log_metrics=True)
        n_envs = 48
        seeds = np.arange(n_envs, dtype=np.uint64) * 13 + 1000 * block + trial
        vec = pkg.VecGridworldCtf(n_envs, device=0, py_seeds=seeds, np_seeds=seeds, log_metrics=True, **kw)
        refs = [oracle.OracleEnv(cfg) for _ in range(n_envs)]
        for e, r in enumerate(refs):
            r.seed(int(seeds[e]), int(seeds[e]))
        acts = torch.empty((n_envs, n), dtype=torch.int8, device=vec.device)
        alive = np.ones(n_envs, bool)
        for t in range(kw["GAME_STEPS"] + 6):
            vec.random_actions(acts, seed=4711, step=t)
            rewards, done, obs, meta = vec.step_observe(acts, auto_reset=True, want_f64=True)
            a, r64, d = acts.cpu().numpy(), vec.rewards64.cpu().numpy(), done.cpu().numpy()
            o, m = obs.cpu().numpy(), meta.cpu().numpy().view(np.uint16)
            for e, r in enumerate(refs):
                if not alive[e]:
                    continue
                if r.get_state().done:
                    r.reset()
                rw, dn, status = r.step(a[e])
                if status:
                    alive[e] = False
                    continue
                ro, rm = r.observe()
                ctx = f"block {block} trial {trial} (G={g} N={n}) env {e} step {t}"
                assert np.array_equal(r64[e], rw) and int(d[e]) == int(dn), ctx
                assert np.array_equal(o[e], ro) and np.array_equal(m[e], rm.view(np.uint16)), ctx
        for e in range(n_envs):
            if alive[e]:
                a_, b_ = view_arrays(vec.get_state(e), n, g), view_arrays(refs[e].get_state(), n, g)
                for k in ("grid", "pos", "hp", "has_flag", "inv", "perm", "metrics", "visitation", "step_count", "team_captures"):
                    assert np.array_equal(np.asarray(a_[k]), np.asarray(b_[k])), f"block {block} trial {trial} env {e} final {k}"
                py, npw = vec.get_rng_state(e)
                rpy, rnp = refs[e].get_rng_state()
                assert np.array_equal(py, rpy) and np.array_equal(npw, rnp), f"block {block} trial {trial} env {e} generators"
        assert (vec.status() & ~abi.ST_NO_RESPAWN) == 0
        vec.close()
        done_cfgs += 1
    assert done_cfgs >= 3
