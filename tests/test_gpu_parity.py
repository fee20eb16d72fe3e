"""Parity of the HIP path (through the C ABI) on a real MI355X: bit-exact against the golden vectors
recorded from the reference and against the CPU oracle on seeded synthetic action streams."""
import copy
import importlib
import pickle
import random

import numpy as np
import pytest

import oracle
from _cases import Case, abi, case_names, cfgmod, pkg, view_arrays

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return 0


def _state_equal(a, b, ctx):
    for k in ("grid", "pos", "hp", "has_flag", "inv", "perm", "metrics", "visitation"):
        assert np.array_equal(a[k], b[k]), f"{ctx}: {k}"
    for k in ("step_count", "done", "team_captures"):
        assert a[k] == b[k], f"{ctx}: {k}"


@pytest.mark.parametrize("name", case_names())
def test_golden_trajectory(name):
    """Env 0 of a 3-env batch replays the reference trajectory; envs 1,2 run other seeds beside it."""
    case = Case(name)
    z, n, g = case.z, case.n, case.g
    seed = case.meta["seed"]
    vec = pkg.VecGridworldCtf(3, device=_dev(), py_seeds=[seed, seed + 1, 0], np_seeds=[seed, seed + 1, 0], **case.kwargs)
    assert vec.TILES_USED == case.meta["tiles_used"]
    py0, np0 = case.seeded_states()
    got_py, got_np = vec.get_rng_state(0)
    assert np.array_equal(got_py, py0) and np.array_equal(got_np, np0), "device seeding != random.seed / np.random.seed"

    obs, meta = vec.observe()
    assert np.array_equal(obs[0].cpu().numpy(), case.unpack_obs(z["obs0"]))
    assert np.array_equal(meta[0].cpu().numpy().view(np.uint16), z["meta0"])

    extra = {int(t): k for k, t in enumerate(z["extra_steps"])}
    acts = torch.zeros((3, n), dtype=torch.int8, device=vec.device)
    mask0 = torch.tensor([1, 0, 0], dtype=torch.uint8, device=vec.device)
    for t in range(case.T):
        if t in case.reset_at:
            vec.reset(mask0)
        a = torch.from_numpy(z["actions"][t])
        acts[0] = a
        acts[1] = a
        acts[2] = torch.from_numpy(z["actions"][(t * 7 + 3) % case.T])
        _, done = vec.step(acts, want_f64=True)
        obs, meta = vec.observe()
        ctx = f"{name} step {t}"
        assert np.array_equal(vec.rewards64[0].cpu().numpy(), z["rewards"][t]), ctx
        assert np.array_equal(vec.rewards[0].cpu().numpy(), z["rewards"][t].astype(np.float32)), ctx
        assert int(done[0]) == int(z["done"][t]), ctx
        assert np.array_equal(obs[0].cpu().numpy(), case.unpack_obs(z["obs"][t])), ctx
        assert np.array_equal(meta[0].cpu().numpy().view(np.uint16), z["meta"][t]), ctx
        if t % 4 == 0 or t in extra:  # the compact form carries the same planes
            codes, meta_c = vec.observe_codes()
            assert np.array_equal(pkg.expand_codes(codes[0].cpu().numpy(), case.obs_shape[1]), case.unpack_obs(z["obs"][t])), ctx
            assert np.array_equal(meta_c[0].cpu().numpy().view(np.uint16), z["meta"][t]), ctx
        if t % 9 == 0 or t in extra or t >= case.T - 3:
            s = view_arrays(vec.get_state(0), n, g)
            assert np.array_equal(s["grid"], z["grid"][t]), ctx
            assert np.array_equal(s["pos"], z["pos"][t]), ctx
            assert np.array_equal(s["hp"], z["hp"][t]), ctx
            assert np.array_equal(s["has_flag"], z["has_flag"][t]), ctx
            assert np.array_equal(s["inv"], z["inv"][t]), ctx
            assert np.array_equal(s["perm"], z["perm"][t]), ctx
            py, npw = vec.get_rng_state(0)
            assert (int(py[624]), int(npw[624])) == (int(z["py_pos"][t]), int(z["np_pos"][t])), ctx + " (draw counts)"
        if t in extra:
            o_unrev = vec.observe(reverse_mask=0)[0][0].cpu().numpy()
            assert np.array_equal(o_unrev, case.unpack_obs(z["extra_obs_unrev"][extra[t]])), ctx
            o_rev = vec.observe(reverse_mask=(1 << n) - 1)[0][0].cpu().numpy()
            assert np.array_equal(o_rev, case.unpack_obs(z["extra_obs_rev"][extra[t]])), ctx
            c_unrev = vec.observe_codes(reverse_mask=0, meta=False)[0][0].cpu().numpy()
            assert np.array_equal(pkg.expand_codes(c_unrev, case.obs_shape[1]), o_unrev), ctx
            c_rev = vec.observe_codes(reverse_mask=(1 << n) - 1, meta=False)[0][0].cpu().numpy()
            assert np.array_equal(pkg.expand_codes(c_rev, case.obs_shape[1]), o_rev), ctx

    s = view_arrays(vec.get_state(0), n, g)
    assert np.array_equal(s["metrics"], z["metrics"])
    assert np.array_equal(s["visitation"], z["visitation"])
    assert s["team_captures"] == case.meta["team_captures"]
    py, npw = vec.get_rng_state(0)
    assert np.array_equal(py, z["py_state"]) and np.array_equal(npw, z["np_state"])
    # envs 1 and 2 run other seeds; on the one-open-spawn-cell map they may legitimately find no respawn cell
    # (the random small maps too; with a spawn on row / column 0 they also meet the negative offset this build flags)
    allowed = abi.ST_NO_RESPAWN if name == "syn_edge_k1" or name.startswith("fuzz_") else 0
    allowed |= abi.ST_SPAWN_EDGE if name.startswith("fuzz_edge0") else 0
    assert vec.status() & ~allowed == 0
    vec.close()


@pytest.mark.parametrize("name,n_envs,steps,log_metrics", [
    ("arena_random", 200, 160, True),     # partial last wave (200 = 3*64 + 8)
    ("arena_random", 1300, 40, True),     # three tile groups of 512 envs, the last one ragged; 1 182 blocks padded to 1 184 for the XCD remap
    ("split_random", 4096, 64, True),     # BASELINE configs[1] size
    ("arena_stress", 130, 320, False),    # metrics compiled out, auto-reset crossing the episode end
    ("syn_axis1_drop", 96, 200, True),
    ("fence_axis0", 70, 120, True),       # obs block not a multiple of 16 bytes (4-byte aligned stores)
    ("syn_edge_k1", 64, 150, True),
    ("donut_1v1", 77, 110, True),         # obs block only 2-byte aligned (byte-store path)
])
@pytest.mark.parametrize("tiles", [0, 1])
def test_batch_matches_oracle(name, n_envs, steps, log_metrics, tiles, monkeypatch):
    """``tiles``: the render as the wave-per-env kernel (k_observe) or as one-shot 8 KiB tiles (k_observe_tiles; taken
    whenever an env's block is 16-byte aligned and >= 8 KiB, the wave-per-env kernel otherwise)."""
    monkeypatch.setenv("CTF_OBS_TILES", str(tiles))
    case = Case(name)
    auto_reset = name == "arena_stress"
    seeds = np.arange(n_envs, dtype=np.uint64) * 977 + 5
    vec = pkg.VecGridworldCtf(n_envs, device=_dev(), py_seeds=seeds, np_seeds=seeds, log_metrics=log_metrics, **case.kwargs)
    cfg, _ = case.config(log_metrics=log_metrics)
    refs = [oracle.OracleEnv(cfg) for _ in range(n_envs)]
    for e, r in enumerate(refs):
        r.seed(int(seeds[e]), int(seeds[e]))
    acts = torch.empty((n_envs, case.n), dtype=torch.int8, device=vec.device)
    alive = np.ones(n_envs, bool)  # envs whose oracle hit an error state (e.g. no respawn cell) are dropped
    for t in range(steps):
        vec.random_actions(acts, seed=0xC0FFEE, step=t, env_offset=11)
        rewards, done, obs, meta = vec.step_observe(acts, auto_reset=auto_reset, want_f64=True)
        if t % 8 == 0:  # compact observation == the one-hot block, every env (dword and byte store paths)
            codes, _ = vec.observe_codes(meta=False)
            assert torch.equal(pkg.expand_codes(codes, vec.N_CHANNELS), obs), f"{name} step {t}: codes"
        a = acts.cpu().numpy()
        r64 = vec.rewards64.cpu().numpy()
        d = done.cpu().numpy()
        o = obs.cpu().numpy()
        m = meta.cpu().numpy().view(np.uint16)
        for e, r in enumerate(refs):
            if not alive[e]:
                continue
            assert np.array_equal(a[e], oracle.philox_actions(case.n, 0xC0FFEE, t, 11 + e))
            if auto_reset and r.get_state().done:
                r.reset()
            rw, dn, status = r.step(a[e])
            if status:
                alive[e] = False
                continue
            ro, rm = r.observe()
            ctx = f"{name} env {e} step {t}"
            assert np.array_equal(r64[e], rw), ctx
            assert int(d[e]) == int(dn), ctx
            assert np.array_equal(o[e], ro), ctx
            assert np.array_equal(m[e], rm.view(np.uint16)), ctx
    assert alive.sum() >= (n_envs // 8 if name == "syn_edge_k1" else n_envs // 2)  # that map starves respawns by design
    for e in range(0, n_envs, max(1, n_envs // 16)):
        if alive[e]:
            _state_equal(view_arrays(vec.get_state(e), case.n, case.g), view_arrays(refs[e].get_state(), case.n, case.g), f"{name} env {e} final")
            py, npw = vec.get_rng_state(e)
            rpy, rnp = refs[e].get_rng_state()
            assert np.array_equal(py, rpy) and np.array_equal(npw, rnp)
    vec.close()


def test_hinted_render_stores_write_the_same_bytes(monkeypatch):
    """The tile render stores with the nontemporal hint when a batch's observations exceed 320 MB (ctf_derive.h: obs_store_nt) — the
    full-size digest tests run that variant, everything smaller the plain one.  Here both at one size, forced by CTF_OBS_NT: the
    same observation and metadata bytes every step (and the plain variant is compared with the oracle all over this file)."""
    case = Case("arena_random")
    n_envs = 1300
    seeds = np.arange(n_envs, dtype=np.uint64) * 13 + 1
    vecs = []
    for nt in ("0", "1"):
        monkeypatch.setenv("CTF_OBS_NT", nt)
        vecs.append(pkg.VecGridworldCtf(n_envs, device=_dev(), py_seeds=seeds, np_seeds=seeds, tune_placement=False, **case.kwargs))
    assert all(v.observe_kernel() == "k_observe_tiles" for v in vecs)
    acts = torch.empty((n_envs, case.n), dtype=torch.int8, device=vecs[0].device)
    for t in range(24):
        vecs[0].random_actions(acts, seed=77, step=t)
        outs = [v.step_observe(acts, auto_reset=True) for v in vecs]
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b), f"step {t}"
        if t % 6 == 0:  # a non-default reversal mask now and then
            o = [v.observe(reverse_mask=0b10100110)[0].clone() for v in vecs]
            assert torch.equal(o[0], o[1]) and o[0].any()
    for v in vecs:
        v.close()


def test_store_hint_follows_what_all_live_handles_on_the_device_write(monkeypatch):
    """The hint is for observations the caches cannot absorb — all of them: two shards of 8 192 arena envs (206 MB each) are stored
    plain one at a time and hinted while both are alive (ctf_abi.hip: store_hint); CTF_OBS_NT forces either way."""
    import gc

    monkeypatch.delenv("CTF_OBS_NT", raising=False)
    gc.collect()  # handles earlier tests dropped without close()
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    a = pkg.VecGridworldCtf(8192, device=_dev(), py_seeds=np.arange(8192, dtype=np.uint64), np_seeds=np.arange(8192, dtype=np.uint64),
                            tune_placement=False, **kw)
    assert a.observe_kernel() == "k_observe_tiles" and a.observe_stores() == "plain"
    b = pkg.VecGridworldCtf(8192, device=_dev(), py_seeds=np.arange(8192, dtype=np.uint64), np_seeds=np.arange(8192, dtype=np.uint64),
                            tune_placement=False, **kw)
    assert a.observe_stores() == "nontemporal" and b.observe_stores() == "nontemporal"
    oa, ob = a.observe()[0], b.observe()[0]
    assert torch.equal(oa, ob)  # (same seeds, no step yet: the same bytes through the hinted kernel as ...
    b.close()
    assert a.observe_stores() == "plain"
    assert torch.equal(a.observe()[0], ob)  # ... through the plain one)
    monkeypatch.setenv("CTF_OBS_NT", "1")
    c = pkg.VecGridworldCtf(64, device=_dev(), py_seeds=np.arange(64, dtype=np.uint64), np_seeds=np.arange(64, dtype=np.uint64), **kw)
    assert c.observe_stores() == "nontemporal"
    c.close()
    a.close()


@pytest.mark.parametrize("name", ["arena_stress", "split_random", "syn_edge_k1"])
@pytest.mark.parametrize("lanes", [1, 2, 4, 8])
def test_every_step_lane_width_matches_the_oracle(name, lanes, monkeypatch):
    """k_step gives every env a group of W lanes (W = the power of two covering the larger opponents list); CTF_STEP_W forces
    another width, with more tagging passes per turn (W < opponents) or idle lanes (W > opponents).  The slot -> lane rotation of
    the hit bits, the randint nibbles fetched from their owner lane and the split of the production all depend on W: every width
    must give the oracle's trajectory, MT19937 states included."""
    monkeypatch.setenv("CTF_STEP_W", str(lanes))
    case = Case(name)
    n_envs, steps = 150, 140
    seeds = np.arange(n_envs, dtype=np.uint64) * 131 + 3
    vec = pkg.VecGridworldCtf(n_envs, device=_dev(), py_seeds=seeds, np_seeds=seeds, **case.kwargs)
    cfg, _ = case.config()
    refs = [oracle.OracleEnv(cfg) for _ in range(n_envs)]
    for e, r in enumerate(refs):
        r.seed(int(seeds[e]), int(seeds[e]))
    acts = torch.empty((n_envs, case.n), dtype=torch.int8, device=vec.device)
    alive = np.ones(n_envs, bool)
    for t in range(steps):
        vec.random_actions(acts, seed=0xBEEF, step=t, env_offset=5)
        vec.step(acts, auto_reset=True, want_f64=True)
        a, r64, d = acts.cpu().numpy(), vec.rewards64.cpu().numpy(), vec.done.cpu().numpy()
        for e, r in enumerate(refs):
            if not alive[e]:
                continue
            if r.get_state().done:
                r.reset()
            rw, dn, status = r.step(a[e])
            if status:
                alive[e] = False
                continue
            assert np.array_equal(r64[e], rw) and int(d[e]) == int(dn), f"{name} W={lanes} env {e} step {t}"
        if t % 20 == 19:
            for e in range(0, n_envs, 7):
                if alive[e]:
                    _state_equal(view_arrays(vec.get_state(e), case.n, case.g), view_arrays(refs[e].get_state(), case.n, case.g),
                                 f"{name} W={lanes} env {e} step {t}")
                    py, npw = vec.get_rng_state(e)
                    rpy, rnp = refs[e].get_rng_state()
                    assert np.array_equal(py, rpy) and np.array_equal(npw, rnp), f"{name} W={lanes} env {e} step {t}: MT19937 states"
    assert alive.sum() >= (n_envs // 8 if name == "syn_edge_k1" else n_envs // 2)
    vec.close()


def _run_against_oracle(vec, cfg, case, seeds, steps, ctx, np_seeds=None, counters=False):
    n_envs = vec.n_envs
    refs = [oracle.OracleEnv(cfg) for _ in range(n_envs)]
    for e, r in enumerate(refs):
        r.seed(int(seeds[e]), int((np_seeds if np_seeds is not None else seeds)[e]))
    acts = torch.empty((n_envs, case.n), dtype=torch.int8, device=vec.device)
    alive = np.ones(n_envs, bool)
    for t in range(steps):
        vec.random_actions(acts, seed=0xFEED, step=t, env_offset=3)
        vec.step(acts, auto_reset=True, want_f64=True)
        a, r64, d = acts.cpu().numpy(), vec.rewards64.cpu().numpy(), vec.done.cpu().numpy()
        for e, r in enumerate(refs):
            if not alive[e]:
                continue
            if r.get_state().done:
                r.reset()
            rw, dn, status = r.step(a[e])
            if status:
                alive[e] = False
                continue
            assert np.array_equal(r64[e], rw) and int(d[e]) == int(dn), f"{ctx} env {e} step {t}"
        if t % 25 == 24 or t == steps - 1:
            ctr = vec.get_rng_counters().cpu().numpy() if counters else None
            for e in range(0, n_envs, 5):
                if alive[e]:
                    _state_equal(view_arrays(vec.get_state(e), case.n, case.g), view_arrays(refs[e].get_state(), case.n, case.g), f"{ctx} env {e} step {t}")
                    if counters:
                        assert tuple(int(x) for x in ctr[e]) == refs[e].get_rng_counters(), f"{ctx} env {e} step {t}: words consumed"
                    else:
                        py, npw = vec.get_rng_state(e)
                        rpy, rnp = refs[e].get_rng_state()
                        assert np.array_equal(py, rpy) and np.array_equal(npw, rnp), f"{ctx} env {e} step {t}: MT19937 states"
    return alive


@pytest.mark.parametrize("every", [0, 1])
def test_ring_regeneration_by_the_tail_blocks_and_by_the_safety_net(every, monkeypatch):
    """The MT19937 blocks are regenerated one launch behind their consumers, by the tail blocks of the next k_step launch (1, the
    default: whole rings, one wave each); a step that needs a ring that is not there regenerates it itself, one word at a time
    (0: no tail blocks at all).  Two implementations of the same digests: same trajectory, same generator states, across
    several block boundaries of both generators (624 words: ~10 arena steps of np.random)."""
    monkeypatch.setenv("CTF_RNG_REFILL_EVERY", str(every))
    case = Case("arena_stress")
    n_envs = 130
    seeds = np.arange(n_envs, dtype=np.uint64) * 59 + 11
    vec = pkg.VecGridworldCtf(n_envs, device=_dev(), py_seeds=seeds, np_seeds=seeds, **case.kwargs)
    alive = _run_against_oracle(vec, case.config()[0], case, seeds, 150, f"tail blocks {every}")
    assert alive.sum() >= n_envs // 2 and vec.status() == 0
    vec.close()


@pytest.mark.parametrize("name", ["arena_random", "split_random", "arena_stress"])
def test_counter_rng_mode_matches_the_oracle_reading_the_same_tape(name):
    """rng_mode="counter": word n of an env's stream is Philox4x32-10(seed, n) and random.shuffle / np.random.rand /
    np.random.randint draw from those words by their own rules — the oracle's three functions read the same tape;
    tests/golden/counter_*.npz pins the same against the reference itself."""
    case = Case(name)
    n_envs = 140
    seeds = np.arange(n_envs, dtype=np.uint64) * 0x9E3779B97F4A7C15 + 77  # any 64-bit values
    np_seeds = seeds ^ np.uint64(0xABCDEF0123456789)
    vec = pkg.VecGridworldCtf(n_envs, device=_dev(), py_seeds=seeds, np_seeds=np_seeds, rng_mode="counter", **case.kwargs)
    cfg, _ = cfgmod.build_config(case.kwargs, log_metrics=True, rng_mode=abi.RNG_COUNTER)
    alive = _run_against_oracle(vec, cfg, case, seeds, 200, f"counter {name}", np_seeds=np_seeds, counters=True)
    assert alive.sum() >= n_envs // 2 and vec.status() == 0
    with pytest.raises(abi.CtfLibraryError):
        vec.get_rng_state(0)  # there is no MT19937 state to hand over in this mode
    # a checkpoint of the RNG = the counters: restoring them reproduces the continuation
    ctr = vec.get_rng_counters().clone()
    snap = [vec.get_state(e) for e in range(n_envs)]
    acts = torch.empty((n_envs, case.n), dtype=torch.int8, device=vec.device)
    outs = []
    for rep in range(2):
        if rep:
            vec.seed(seeds, np_seeds)
            vec.set_rng_counters(ctr)
            for e in range(n_envs):
                vec.set_state(e, snap[e])
        rows = []
        for t in range(30):
            vec.random_actions(acts, seed=0xABC, step=t)
            vec.step(acts, auto_reset=True, want_f64=True)
            rows.append(vec.rewards64.clone())
        outs.append((torch.stack(rows), vec.get_rng_counters().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    vec.close()


def test_full_size_arena_properties_and_sample():
    """BASELINE configs[2] size (65 536 arena envs): size-independent properties on everything plus a
    bit-exact oracle check on a 64-env sample."""
    case = Case("arena_random")
    E, steps = 65536, 12
    seeds = np.arange(E, dtype=np.uint64) + 1_000_003
    vec = pkg.VecGridworldCtf(E, device=_dev(), py_seeds=seeds, np_seeds=seeds, log_metrics=False, **case.kwargs)
    cfg, _ = case.config(log_metrics=False)
    sample = np.linspace(0, E - 1, 64).astype(int)
    refs = {int(e): oracle.OracleEnv(cfg) for e in sample}
    for e, r in refs.items():
        r.seed(int(seeds[e]), int(seeds[e]))
    acts = torch.empty((E, case.n), dtype=torch.int8, device=vec.device)
    for t in range(steps):
        vec.random_actions(acts, seed=7, step=t)
        rewards, done, obs, meta = vec.step_observe(acts, want_f64=True)
        # plane 0 of every agent is one-hot; every cell is hot in at most 2 planes (own position + a tile plane)
        assert bool((obs[:, :, 0].sum(dim=(2, 3), dtype=torch.int32) == 1).all())
        assert int(obs.max()) == 1
        # team 0 sees exactly 4 own-team agent cells and 4 opponents (channels of tiles 4..7 / 8..11)
        tiles = vec.TILES_USED
        own = [k + 1 for k, tl in enumerate(tiles) if 4 <= tl <= 7]
        opp = [k + 1 for k, tl in enumerate(tiles) if 8 <= tl <= 11]
        assert bool((obs[:, :, own].sum(dim=(2, 3, 4), dtype=torch.int32) == 4).all())
        assert bool((obs[:, :, opp].sum(dim=(2, 3, 4), dtype=torch.int32) == 4).all())
        # a team-1 agent's reversed view of the walls equals a team-0 agent's view rotated by 180 degrees
        wall = tiles.index(1) + 1
        assert bool((obs[:, 0, wall] == torch.flip(obs[:, 1, wall], dims=(1, 2))).all())
        if t % 4 == 3:  # the compact observation of all 65 536 envs expands to the same 1.65 GB block
            codes, _ = vec.observe_codes(meta=False)
            for lo in range(0, E, 8192):
                assert torch.equal(pkg.expand_codes(codes[lo:lo + 8192], vec.N_CHANNELS), obs[lo:lo + 8192]), (t, lo)
        a = acts[torch.from_numpy(sample).to(vec.device)].cpu().numpy()
        o = obs[torch.from_numpy(sample).to(vec.device)].cpu().numpy()
        m = meta[torch.from_numpy(sample).to(vec.device)].cpu().numpy().view(np.uint16)
        r64 = vec.rewards64[torch.from_numpy(sample).to(vec.device)].cpu().numpy()
        for k, e in enumerate(sample):
            rw, dn, _ = refs[int(e)].step(a[k])
            ro, rm = refs[int(e)].observe()
            assert np.array_equal(r64[k], rw) and np.array_equal(o[k], ro) and np.array_equal(m[k], rm.view(np.uint16)), (t, e)
    assert vec.status() == 0
    vec.close()


def _digest_rows(t2d, w, mask=None, chunk=2048):
    """sum(value_i * w(i)) mod 2**64 of every row of an integer tensor [E, K] (oracle.digest_weights), in wrapping int64"""
    out = torch.empty(t2d.shape[0], dtype=torch.int64, device=t2d.device)
    for lo in range(0, t2d.shape[0], chunk):
        v = t2d[lo:lo + chunk].to(torch.int64)
        if mask is not None:
            v = v & mask
        out[lo:lo + chunk] = (v * w).sum(1)
    return out


@pytest.mark.parametrize("workload,E,lo", [("arena", 65536, 0), ("arena20", 65536, 0), ("arena", 32768, 3 * 32768)])
def test_every_env_of_the_benchs_state_matches_the_oracle(workload, E, lo):
    """What bench.py times, checked on ALL 65 536 envs: its own protocol (bench.stagger_phases: a discarded first episode, masked
    resets that leave the envs at staggered episode phases, ~131 envs auto-reset per step) and then 32 step_observe(auto_reset)
    steps, against the OpenMP oracle running the same protocol env by env (oracle.bench_digest).  Compared per env and per
    step: the float64 rewards' bit patterns, done, the whole observation block and the metadata rows (as 64-bit weighted
    sums), and at the end both MT19937 states, env_step_count and the capture counters.  The render runs through
    ctf_step_observe, i.e. with the ring regeneration on the side stream beside it.

    The third case is BASELINE.json configs[3]'s per-GPU shard — 262 144 envs global over 8 GPUs = 32 768 envs — created exactly as
    bench.py's rank 3 of 8 creates it (and as `secondary.arena_32768` of the N=1 line does): seeds and action streams are functions
    of the GLOBAL env index, envs [98 304, 131 072); the staggering phase is the local index, as in bench.stagger_phases."""
    import bench

    steps, period = 32, 500
    kw = _workload(workload)
    seeds = pkg.sharding.env_seeds(1, lo, lo + E)
    assert (lo, lo + E) == ((0, E) if lo == 0 else pkg.sharding.shard_range(262144, 3, 8))
    vec = pkg.VecGridworldCtf(E, device=_dev(), py_seeds=seeds, np_seeds=seeds, log_metrics=True, tune_placement=False, **kw)
    cfg, _ = cfgmod.build_config(kw, log_metrics=True)
    import os, time

    t0 = time.perf_counter()
    want = oracle.bench_digest(cfg, seeds, lo, period, 0x5747, steps, 0xC7F, min(len(os.sched_getaffinity(0)), 16))
    t_cpu = time.perf_counter() - t0
    vec.observe()
    bench.stagger_phases(vec, torch, lo, period)
    N, dev = vec.N_AGENTS, vec.device
    wt = lambda n: torch.from_numpy(oracle.digest_weights(n).view(np.int64)).to(dev)
    w_obs, w_meta, w_rew, w_rng = wt(vec.obs[0].numel()), wt(N * vec.META_LEN), wt(N), wt(625)
    acts = torch.empty((E, N), dtype=torch.int8, device=dev)
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    for t in range(steps):
        vec.random_actions(acts, seed=0xC7F, step=t, env_offset=lo)
        rewards, done, obs, meta = vec.step_observe(acts, auto_reset=True, want_f64=True)
        bad = (_digest_rows(obs.view(E, -1), w_obs) != i64(want["obs"][t])).nonzero()
        assert bad.numel() == 0, f"{workload} step {t}: observation of envs {bad[:8].flatten().tolist()} ({bad.numel()} in all)"
        assert torch.equal(_digest_rows(meta.view(torch.int16).view(E, -1), w_meta, mask=0xFFFF), i64(want["meta"][t])), f"{workload} step {t}: metadata"
        assert torch.equal(_digest_rows(vec.rewards64.view(torch.int64), w_rew), i64(want["rew"][t])), f"{workload} step {t}: rewards"
        assert torch.equal(done.cpu(), torch.from_numpy(want["done"][t])), f"{workload} step {t}: done"
    py, npw = vec.get_rng_states()
    assert torch.equal(_digest_rows(py, w_rng, mask=0xFFFFFFFF), i64(want["rng"][:, 0])), "random states"
    assert torch.equal(_digest_rows(npw, w_rng, mask=0xFFFFFFFF), i64(want["rng"][:, 1])), "np.random states"
    _, caps, nsteps = vec.counters()
    assert np.array_equal(nsteps.cpu().numpy(), want["misc"][:, 0]) and np.array_equal(caps.cpu().numpy(), want["misc"][:, 1:])
    assert vec.status() == 0
    print(f"\n{workload}: all {E} envs (global [{lo}, {lo + E})) x {steps} steps equal the oracle; oracle {t_cpu:.1f} s on the host cores")
    vec.close()


def _workload(name):
    if name == "arena20":
        return dict(pkg.configs.ARENA20_KWARGS, SCENARIO=pkg.configs.arena20_scenario())
    return dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)


@pytest.mark.parametrize("workload,auto_reset,tiles", [("arena", True, 0), ("arena", False, 1), ("arena20", True, 1), ("arena", True, 1)])
def test_bench_variant_at_bench_size_over_an_episode_end_and_a_visitation_fold(workload, auto_reset, tiles, monkeypatch):
    """The kernel variant bench.py times — metrics counters + visitation log ON, 65 536 envs — for 520 steps: with
    auto_reset the episode ends at step 500 and every env is reset inside the step launch; without it the envs run past
    GAME_STEPS and cross the 511-step in-kernel fold of the visitation log (gridworld_ctf.py:849-918, :479-486).  A 64-env
    sample is stepped through the oracle beside the GPU and compared after EVERY step (rewards f64, done, all N
    observations, all N metadata rows); at the end: every counter of ``counters()`` for the sample, the full state views
    (visitation maps included) and both MT19937 states.  Size-independent properties hold on all 65 536 envs throughout."""
    monkeypatch.setenv("CTF_OBS_TILES", str(tiles))
    kw = _workload(workload)
    E, steps = 65536, 520
    seeds = np.arange(E, dtype=np.uint64) + 2_000_006
    vec = pkg.VecGridworldCtf(E, device=_dev(), py_seeds=seeds, np_seeds=seeds, log_metrics=True, **kw)
    n, g = vec.N_AGENTS, vec.GRID_SIZE
    cfg, _ = pkg.config.build_config(kw, log_metrics=True)
    sample = np.unique(np.concatenate([np.linspace(0, E - 1, 61).astype(int), [1, 63, 64]]))
    sidx = torch.from_numpy(sample).to(vec.device)
    refs = [oracle.OracleEnv(cfg) for _ in sample]
    for e, r in zip(sample, refs):
        r.seed(int(seeds[e]), int(seeds[e]))
    acts = torch.empty((E, n), dtype=torch.int8, device=vec.device)
    tiles = vec.TILES_USED
    own = [k + 1 for k, tl in enumerate(tiles) if 4 <= tl <= 7]
    opp = [k + 1 for k, tl in enumerate(tiles) if 8 <= tl <= 11]
    for t in range(steps):
        vec.random_actions(acts, seed=0xBE7C, step=t)
        rewards, done, obs, meta = vec.step_observe(acts, auto_reset=auto_reset, want_f64=True)
        if t % 40 == 0 or t in (499, 500, 501, 510, 511, 512, steps - 1):  # properties of every env (1.65 / 2.9 GB reductions)
            assert bool((obs[:, :, 0].sum(dim=(2, 3), dtype=torch.int32) == 1).all()) and int(obs.max()) == 1
            assert bool((obs[:, :, own].sum(dim=(2, 3, 4), dtype=torch.int32) == 4).all())
            assert bool((obs[:, :, opp].sum(dim=(2, 3, 4), dtype=torch.int32) == 4).all())
            want_done = (t + 1 >= 500) if not auto_reset else (t + 1 == 500)
            assert bool((done == int(want_done)).all()), t
            codes, _ = vec.observe_codes(meta=False)
            for lo in range(0, E, 16384):
                assert torch.equal(pkg.expand_codes(codes[lo:lo + 16384], vec.N_CHANNELS), obs[lo:lo + 16384]), (t, lo)
        a = acts[sidx].cpu().numpy()
        o = obs[sidx].cpu().numpy()
        m = meta[sidx].cpu().numpy().view(np.uint16)
        r64 = vec.rewards64[sidx].cpu().numpy()
        r32 = rewards[sidx].cpu().numpy()
        d = done[sidx].cpu().numpy()
        for k, r in enumerate(refs):
            if auto_reset and r.get_state().done:
                r.reset()
            rw, dn, status = r.step(a[k])
            assert status == 0
            ro, rm = r.observe()
            ctx = f"{workload} env {sample[k]} step {t}"
            assert np.array_equal(r64[k], rw) and np.array_equal(r32[k], rw.astype(np.float32)), ctx
            assert int(d[k]) == int(dn), ctx
            assert np.array_equal(o[k], ro), ctx
            assert np.array_equal(m[k], rm.view(np.uint16)), ctx
    met, caps, nsteps = vec.counters()
    met, caps, nsteps = met[sidx].cpu().numpy(), caps[sidx].cpu().numpy(), nsteps[sidx].cpu().numpy()
    for k, r in enumerate(refs):
        want = view_arrays(r.get_state(), n, g)
        assert np.array_equal(met[k], want["metrics"]), f"{workload} env {sample[k]}: counters()"
        assert list(caps[k]) == want["team_captures"] and int(nsteps[k]) == want["step_count"]
        assert want["step_count"] == (steps if not auto_reset else steps - 500)
        _state_equal(view_arrays(vec.get_state(int(sample[k])), n, g), want, f"{workload} env {sample[k]} final state")
        py, npw = vec.get_rng_state(int(sample[k]))
        rpy, rnp = r.get_rng_state()
        assert np.array_equal(py, rpy) and np.array_equal(npw, rnp)
    assert vec.status() == 0
    vec.close()


@pytest.mark.parametrize("workload,tiles", [("arena", 0), ("arena", 1), ("arena20", 1), ("split", 0)])
def test_full_size_renders_are_run_to_run_identical(workload, tiles, monkeypatch):
    """The render's next-env state arrives through hand-placed loads with counted waits (ctf_kernels.hip): a wait that is
    one store short would show up as rare, timing-dependent differences at full size only.  Both renders, four runs each
    into fresh buffers on a busy device, must be identical — and equal to each other through expand_codes."""
    monkeypatch.setenv("CTF_OBS_TILES", str(tiles))
    kw = dict(pkg.configs.SPLIT_KWARGS, SCENARIO=pkg.CtfScenarios.arrow) if workload == "split" else _workload(workload)
    E = 65536
    vec = pkg.VecGridworldCtf(E, device=_dev(), py_seeds=np.arange(E) + 3, np_seeds=np.arange(E) + 3, **kw)
    acts = torch.empty((E, vec.N_AGENTS), dtype=torch.int8, device=vec.device)
    for t in range(25):
        vec.random_actions(acts, seed=77, step=t)
        vec.step(acts)
    first_obs, first_meta = (x.clone() for x in vec.observe())
    first_codes = vec.observe_codes()[0].clone()
    for run in range(4):
        vec.obs = torch.empty_like(first_obs)   # another placement each time
        vec.meta.zero_()
        obs, meta = vec.observe()
        assert torch.equal(obs, first_obs) and torch.equal(meta, first_meta), run
        vec.codes.zero_()
        codes, meta = vec.observe_codes()
        assert torch.equal(codes, first_codes) and torch.equal(meta, first_meta), run
    for lo in range(0, E, 16384):
        assert torch.equal(pkg.expand_codes(first_codes[lo:lo + 16384], vec.N_CHANNELS), first_obs[lo:lo + 16384])
    vec.close()


def test_reset_mask_and_state_roundtrip():
    case = Case("arena_random")
    vec = pkg.VecGridworldCtf(8, device=_dev(), **case.kwargs)
    acts = torch.empty((8, case.n), dtype=torch.int8, device=vec.device)
    for t in range(30):
        vec.random_actions(acts, seed=3, step=t)
        vec.step(acts)
    before = [view_arrays(vec.get_state(e), case.n, case.g) for e in range(8)]
    mask = torch.tensor([0, 1, 0, 0, 1, 0, 0, 0], dtype=torch.uint8, device=vec.device)
    vec.reset(mask)
    fresh = view_arrays(pkg.VecGridworldCtf(1, device=_dev(), **case.kwargs).get_state(0), case.n, case.g)
    for e in range(8):
        now = view_arrays(vec.get_state(e), case.n, case.g)
        if mask[e]:
            for k in ("grid", "pos", "hp", "has_flag", "inv", "metrics", "visitation"):
                assert np.array_equal(now[k], fresh[k]), k
            assert now["step_count"] == 0 and now["done"] == 0
            assert np.array_equal(now["perm"], before[e]["perm"])  # _arr survives reset (gridworld_ctf.py:244)
        else:
            _state_equal(now, before[e], f"env {e} untouched")
    # set_state(get_state) is the identity, also across envs
    v = vec.get_state(3)
    vec.set_state(5, v)
    _state_equal(view_arrays(vec.get_state(5), case.n, case.g), view_arrays(v, case.n, case.g), "roundtrip")
    py, npw = vec.get_rng_state(3)
    vec.set_rng_state(5, py, npw)
    vec.random_actions(acts, seed=9, step=0)
    acts[5] = acts[3]
    vec.step(acts, want_f64=True)
    _state_equal(view_arrays(vec.get_state(5), case.n, case.g), view_arrays(vec.get_state(3), case.n, case.g), "twin envs")
    assert np.array_equal(vec.rewards64[5].cpu().numpy(), vec.rewards64[3].cpu().numpy())
    vec.close()


def test_bulk_rng_hand_over_equals_the_per_env_one():
    """ctf_get_rng_states / ctf_set_rng_states ([E][625], stream-ordered) against the per-env synchronous calls: states
    mid-block (lazily regenerated on the device, finished by the export kernel) and freshly seeded ones; a state set in
    bulk continues exactly like the env it came from."""
    case = Case("arena_random")
    E = 300
    seeds = np.arange(E, dtype=np.uint64) * 31 + 7
    vec = pkg.VecGridworldCtf(E, device=_dev(), py_seeds=seeds, np_seeds=seeds, **case.kwargs)
    acts = torch.empty((E, case.n), dtype=torch.int8, device=vec.device)
    for t in range(23):  # ~64 np words and ~14 py words per step: several block boundaries are crossed, at different points per env
        if t in (0, 7, 22):
            py, npw = (x.cpu().numpy().view(np.uint32) for x in vec.get_rng_states())
            for e in (0, 1, 63, 64, 150, E - 1):
                one_py, one_np = vec.get_rng_state(e)
                assert np.array_equal(py[e], one_py) and np.array_equal(npw[e], one_np), (t, e)
        vec.random_actions(acts, seed=11, step=t)
        vec.step(acts)
    # twin batch: states handed over in bulk, same actions from here on -> identical trajectories
    twin = pkg.VecGridworldCtf(E, device=_dev(), **case.kwargs)
    py_d, np_d = vec.get_rng_states()
    twin.set_rng_states(py_d, np_d)
    for e in (0, 64, E - 1):
        twin.set_state(e, vec.get_state(e))
    for t in range(23, 40):
        vec.random_actions(acts, seed=11, step=t)
        vec.step(acts, want_f64=True)
        twin.step(acts, want_f64=True)
        for e in (0, 64, E - 1):
            assert torch.equal(vec.rewards64[e], twin.rewards64[e]), (t, e)
    for e in (0, 64, E - 1):
        _state_equal(view_arrays(twin.get_state(e), case.n, case.g), view_arrays(vec.get_state(e), case.n, case.g), f"twin env {e}")
        a, b = vec.get_rng_state(e), twin.get_rng_state(e)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    only_np = twin.get_rng_states(py=False)
    assert only_np[0] is None and only_np[1].shape == (E, 625)
    vec.close()
    twin.close()


def test_status_bits_for_bad_action_and_action_mask():
    case = Case("split_random")
    vec = pkg.VecGridworldCtf(4, device=_dev(), **case.kwargs)
    acts = torch.zeros((4, case.n), dtype=torch.int8, device=vec.device)
    acts[2, 1] = 9
    vec.step(acts)
    assert vec.status() & abi.ST_BAD_ACTION
    assert vec.status() == 0  # cleared by the read
    mask = vec.action_mask()
    want = np.array([[1] * 5 + ([0] * 4 if vec.AGENT_TYPES[i] in (0, 1) else [1] * 4) for i in range(case.n)], np.uint8)
    assert np.array_equal(mask, want)
    vec.close()


@pytest.mark.parametrize("name,steps", [("arena_stress", 560), ("split_random", 300), ("fuzz_13", 120), ("donut_1v1", 150)])
def test_host_step_round_trip_matches_the_oracle(name, steps):
    """ctf_host_step (the batch-of-one entry behind the reference's class API): every step hands both generator states in, and
    compares everything the ONE round trip brings back — float64 rewards, done, status, the whole state view (counters and
    visitation maps included: 560 steps without a reset cross the 511-step fold of the visitation log), all N observations and
    metadata rows, both generator states — with the oracle stepped beside it."""
    case = Case(name)
    kw = dict(case.kwargs)
    cfg, _ = cfgmod.build_config(kw, log_metrics=True)
    n, g, c = case.n, case.g, case.c
    vec = pkg.VecGridworldCtf(1, device=_dev(), py_seeds=[0], np_seeds=[0], **kw)
    ref = oracle.OracleEnv(cfg)
    ref.seed(case.meta["seed"], case.meta["seed"])
    py, npw = ref.get_rng_state()
    obs, meta = np.zeros((n, c, g, g), np.uint8), np.zeros((n, 2 * n + 6), np.float16)
    # no step: the state after the first reset, its observation; the device's own generators (seed 0) are left alone
    _, done, status, view, _, _ = vec.host_step(None, obs=obs, meta=meta)
    ro, rm = ref.observe()
    assert not done and status == 0 and np.array_equal(obs, ro) and np.array_equal(meta.view(np.uint16), rm.view(np.uint16))
    _state_equal(view_arrays(view, n, g), view_arrays(ref.get_state(), n, g), f"{name} initial")
    rng = np.random.default_rng(17)
    for t in range(steps):
        a = rng.integers(0, 9, n).astype(np.int8)
        # two steps in three leave the device's generators where the previous step left them (what the facade does while the caller's
        # random / np.random are untouched): the streams then run on across block ends by the step kernel's own ring regeneration
        keep = t % 3 != 0
        r64, done, status, view, py, npw = vec.host_step(a, None if keep else py, None if keep else npw, rng_out=True, obs=obs, meta=meta)
        rw, dn, st = ref.step(a)
        ctx = f"{name} step {t}"
        assert status == st and done == dn and np.array_equal(r64, rw), ctx
        _state_equal(view_arrays(view, n, g), view_arrays(ref.get_state(), n, g), ctx)
        ro, rm = ref.observe()
        assert np.array_equal(obs, ro) and np.array_equal(meta.view(np.uint16), rm.view(np.uint16)), ctx
        rpy, rnp = ref.get_rng_state()
        assert np.array_equal(py, rpy) and np.array_equal(npw, rnp), ctx + ": generator states"
    assert view.step_count == steps
    vec.close()
    big = pkg.VecGridworldCtf(2, device=_dev(), **kw)
    with pytest.raises(abi.CtfLibraryError, match="ONE env"):
        big.host_step(None)
    big.close()


def test_facade_with_private_device_streams_leaves_the_global_generators_alone():
    """GridworldCtf(rng="device", seed=s): the env draws from its own device streams (seeded like random.seed(s); np.random.seed(s)),
    the process-global generators are neither read nor advanced, and the trajectory is the reference's for that seed."""
    case = Case("arena_random")
    z = case.z
    random.seed(999)
    np.random.seed(999)
    before = (random.getstate(), np.random.get_state())
    env = pkg.GridworldCtf(rng="device", seed=case.meta["seed"], **case.kwargs)
    for t in range(40):
        _, rewards, done = env.step([int(a) for a in z["actions"][t]])
        assert np.array_equal(env.grid, z["grid"][t]) and rewards == [float(r) for r in z["rewards"][t]]
        assert np.array_equal(env.standardise_state(3, reverse_grid=True), case.unpack_obs(z["obs"][t])[3][None])
    after = (random.getstate(), np.random.get_state())
    assert after[0] == before[0] and np.array_equal(after[1][1], before[1][1]) and after[1][2] == before[1][2]


def test_spawn_edge_status_bit_and_the_facades_index_error():
    """The documented difference from the reference: a respawn offset that goes negative (spawn on row / column 0,
    gridworld_ctf.py:773-785) sets CTF_ST_SPAWN_EDGE — same state as the oracle — and the drop-in facade raises IndexError."""
    from _cases import spawn_edge_kwargs

    kw = spawn_edge_kwargs()
    cfg, _ = cfgmod.build_config(kw, log_metrics=True)
    seeds = np.arange(8, dtype=np.uint64) + 5
    vec = pkg.VecGridworldCtf(8, device=_dev(), py_seeds=seeds, np_seeds=seeds, **kw)
    acts = torch.full((8, 2), 4, dtype=torch.int8, device=vec.device)
    vec.step(acts)
    assert vec.status() == abi.ST_SPAWN_EDGE
    for e in range(8):
        ref = oracle.OracleEnv(cfg)
        ref.seed(int(seeds[e]), int(seeds[e]))
        _, _, st = ref.step(np.array([4, 4], np.int8))
        assert st == abi.ST_SPAWN_EDGE
        _state_equal(view_arrays(vec.get_state(e), 2, 7), view_arrays(ref.get_state(), 2, 7), f"env {e}")
    vec.close()
    random.seed(5)
    np.random.seed(5)
    env = pkg.GridworldCtf(**kw)
    with pytest.raises(IndexError):
        env.step([4, 4])


def test_facade_is_a_drop_in_for_the_reference_api():
    """The single-env facade driven exactly as ppo.py / utils.duel drive the reference, global RNG contract included."""
    case = Case("script_8_arena")
    z, n = case.z, case.n
    random.seed(case.meta["seed"])
    np.random.seed(case.meta["seed"])
    env = pkg.GridworldCtf(**case.kwargs)
    assert env.get_env_dims()[0] == tuple(case.obs_shape[1:]) and env.get_env_dims()[2] == (case.meta["meta_len"],)
    assert env.N_AGENTS == n and env.GRID_SIZE == case.g
    for t in range(60):
        for i in np.arange(env.N_AGENTS):  # np.int64 indices, as the reference's loops use
            rev = env.AGENT_TEAMS[i] != 0
            s = env.standardise_state(i, reverse_grid=rev)
            m = env.get_env_metadata(i)
            want_s = case.unpack_obs(z["obs"][t - 1] if t else z["obs0"])[i][None]
            want_m = (z["meta"][t - 1] if t else z["meta0"])[i][None]
            assert s.shape == want_s.shape and s.dtype == np.uint8 and np.array_equal(s, want_s)
            assert m.dtype == np.float16 and np.array_equal(m.view(np.uint16), want_m)
            assert env.AGENT_TYPE_ACTION_MASK[env.AGENT_TYPES[i]] in (0, 1)
        grid, rewards, done = env.step([int(a) for a in z["actions"][t]])
        assert grid is env.grid and np.array_equal(grid, z["grid"][t])
        assert isinstance(rewards, list) and rewards == [float(r) for r in z["rewards"][t]]
        assert done == bool(z["done"][t])
        assert env.agent_positions == {i: tuple(int(x) for x in z["pos"][t][i]) for i in range(n)}
        assert np.array_equal(env.has_flag, z["has_flag"][t])
        assert env._arr == [int(x) for x in z["perm"][t]]
        assert random.getstate()[1][624] == int(z["py_pos"][t]) and np.random.get_state()[2] == int(z["np_pos"][t])
    assert env.get_reversed_action(0) == 1 and env.get_reversed_action(7) == 8  # FLIP_AXIS None
    with pytest.raises(KeyError):
        env.step([9] * n)
    # travels by value like the reference env does under Ray (deepcopy / pickle), continuing identically
    twin = copy.deepcopy(env)
    blob = pickle.loads(pickle.dumps(env))
    import cloudpickle  # what Ray moves its task arguments with (the reference's env only survives cloudpickle)

    cblob = cloudpickle.loads(cloudpickle.dumps(env))
    st_py, st_np = random.getstate(), np.random.get_state()
    out_a = env.step([int(a) for a in z["actions"][60]])
    for other in (twin, blob, cblob):
        random.setstate(st_py)
        np.random.set_state(st_np)
        out_b = other.step([int(a) for a in z["actions"][60]])
        assert np.array_equal(out_a[0], out_b[0]) and out_a[1] == out_b[1] and out_a[2] == out_b[2]
    assert np.array_equal(out_a[0], z["grid"][60])
    met = env.metrics
    assert met["team_flag_captures"] == {0: 0, 1: 0} and set(met["agent_visitation_maps"]) == set(range(n))


def test_symmetry_assertion_and_no_respawn_cell_errors_match_the_reference():
    maps = pkg.CtfScenarios
    bad = dict(pkg.configs.SPLIT_KWARGS, SCENARIO=maps.arrow, MAP_SYMMETRY_CHECK=True)  # asymmetric types -> AssertionError
    with pytest.raises(AssertionError):
        pkg.GridworldCtf(**bad)
    # a spawn whose 3x3 window has no open cell: np.random.randint(0) -> ValueError in the reference
    scen = copy.deepcopy(maps.arrow)
    scen["SPAWN_POSITIONS"] = {0: (6, 1), 1: (9, 9)}
    scen["BLOCK_TILE_SLICES"] = list(scen["BLOCK_TILE_SLICES"]) + [(8, 8), (8, 9), (8, 10), (9, 8), (9, 10), (10, 8), (10, 9), (10, 10)]
    scen["AGENT_STARTING_POSITIONS"] = {0: (5, 2), 1: (5, 3), 2: (5, 1), 3: (9, 5)}
    kw = dict(pkg.configs.SPLIT_KWARGS, SCENARIO=scen, TAG_PROBABILITY=1.0, AGENT_TYPE_HP={0: 1, 1: 8, 2: 8, 3: 7})
    env = pkg.GridworldCtf(rng="device", seed=1, **kw)
    with pytest.raises(ValueError):
        for _ in range(20):
            env.step([4, 4, 4, 4])  # agent 0 (guardian, team 0) stands next to agent 1 (team 1, hp 1)


def test_one_hip_runtime_in_the_process():
    maps_txt = open("/proc/self/maps").read()
    libs = {line.split()[-1] for line in maps_txt.splitlines() if "libamdhip64" in line}
    assert len(libs) == 1, libs
    assert any("libctf_hip.so" in line for line in maps_txt.splitlines())


@pytest.mark.gpu
def test_placement_search_is_bounded_and_says_what_it_found():
    """VecGridworldCtf looks for an observation buffer of the fast kind on first use: it holds two buffers at most, tries no
    more than it was told to, leaves the render's output identical, and reports kind / ratio / candidates / bytes / time."""
    torch = pytest.importorskip("torch")
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    E = 16384  # 413 MB of observations: above the 256 MiB threshold of the default
    ref = pkg.VecGridworldCtf(E, device=_dev(), tune_placement=False, **kw)
    want, want_meta = ref.observe()
    want = want.clone()
    vec = pkg.VecGridworldCtf(E, device=_dev(), placement_tries=5, placement_gib=2, **kw)
    assert vec.placement is None
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    got, got_meta = vec.observe()
    pl = vec.placement
    nbytes = got.numel()
    assert set(pl) >= {"kind", "render_over_fill", "render_ms", "fill_ms", "candidates", "slowest_candidate_render_ms", "peak_held_bytes",
                       "searched_bytes", "search_ms"}
    assert 1 <= pl["candidates"] <= 5 and pl["kind"] in ("fast", "intermediate", "slow") and len(vec.placement_probe_ms) == pl["candidates"]
    assert pl["peak_held_bytes"] <= 2 * nbytes and pl["searched_bytes"] == pl["candidates"] * nbytes
    assert torch.cuda.max_memory_allocated() - base <= 2 * nbytes + (64 << 20)  # the candidate and the best so far, never more
    assert pl["render_ms"] == min(vec.placement_probe_ms) and pl["slowest_candidate_render_ms"] == max(vec.placement_probe_ms)
    assert torch.equal(got, want) and torch.equal(got_meta, want_meta)
    # a cap below two buffers: no search at all
    small = pkg.VecGridworldCtf(E, device=_dev(), placement_gib=0.5, **kw)
    small.observe()
    assert small.placement["candidates"] == 1
    for v in (ref, vec, small):
        v.close()
