"""Host-side translation of the reference's constructor kwargs + scenario dict into the flat
``ctf_config`` the C ABI takes.

What stays in Python on purpose: the numpy-slice painting of the map
(reference gridworld_ctf.py:352-381), the ``TILES_USED`` channel order, which is whatever CPython's
``list(set(...))`` yields (gridworld_ctf.py:488-499), and the truncated ``OPPONENTS`` lists
(gridworld_ctf.py:391-395).
"""
import numpy as np

try:  # imported as a package module or, for drop-in use, with this directory on sys.path
    from . import _abi
except ImportError:  # pragma: no cover
    import _abi

# constructor defaults, gridworld_ctf.py:19-52
DEFAULT_KWARGS = dict(
    AGENT_CONFIG={0: {"team": 0, "type": 0}, 1: {"team": 1, "type": 0}},
    SCENARIO=None,
    GAME_STEPS=256,
    GRID_SIZE=10,
    ENABLE_OBSTACLES=False,
    DROP_FLAG_WHEN_NO_HP=False,
    HOME_FLAG_CAPTURE=False,
    USE_EASY_CAPTURE=True,
    USE_ADJUSTED_REWARDS=False,
    MAX_BLOCK_TILE_PCT=0.2,
    LOG_METRICS=True,
    MAP_SYMMETRY_CHECK=True,
    AGENT_TYPE_HP={0: 8, 1: 6, 2: 4, 3: 4},
    AGENT_HP_HEALING_PER_STEP=0.25,
    AGENT_TYPE_DAMAGE={0: 1, 1: 0.5, 2: 1, 3: 1},
    TAG_PROBABILITY=0.75,
    GUARDIAN_DAMAGE_MULTIPLIER=5.0,
    VAULT_HP_COST=0.5,
    VAULT_MIN_HP=2.5,
)

OPEN_TILE, BLOCK_TILE, DESTRUCTIBLE_TILE1, DESTRUCTIBLE_TILE2 = 0, 1, 2, 3
FLAG_TILE_MAP = {0: 12, 1: 13}
# tile of an agent: 4 + type for team 0, 8 + type for team 1 (gridworld_ctf.py:256-264)
AGENT_TYPE_TILE_MAP = {t: {0: 4 + t, 1: 8 + t} for t in range(4)}


def paint_grid(scenario, n_agents, agent_tile_map):
    """The map at reset: later writes win (blocks, destructibles, flags, agents)."""
    g = int(scenario["GRID_SIZE"])
    grid = np.zeros((g, g), dtype=np.uint8)
    for slc in scenario["BLOCK_TILE_SLICES"]:
        grid[slc] = BLOCK_TILE
    for slc in scenario["DESTRUCTIBLE_TILE_SLICES"]:
        grid[slc] = DESTRUCTIBLE_TILE1
    for team in (0, 1):
        grid[scenario["FLAG_POSITIONS"][team]] = FLAG_TILE_MAP[team]
    for i in range(n_agents):
        grid[scenario["AGENT_STARTING_POSITIONS"][i]] = agent_tile_map[i]
    return grid


def tiles_used(grid, agent_types):
    """Channel k+1 of an observation shows tile TILES_USED[k].  The order is CPython's set iteration
    order over the same elements inserted in the same order as the reference does."""
    tiles = [x for x in np.unique(grid) if x != 0]
    if DESTRUCTIBLE_TILE1 in tiles and 3 in agent_types.values():
        tiles.append(DESTRUCTIBLE_TILE2)
    tiles += [8 + t for t in agent_types.values()]  # opponent-coloured tile of every agent type present
    return list(set(tiles))


def opponents_of(agent_teams):
    """OPPONENTS[t] = agents of the other team in dict order, cut to N // 2."""
    half = len(agent_teams) // 2
    return {
        0: [k for k, v in agent_teams.items() if v == 1][:half],
        1: [k for k, v in agent_teams.items() if v == 0][:half],
    }


def build_config(kwargs, log_metrics=True, rng_mode=0):
    """-> (CtfConfig, derived) where ``derived`` holds the host-side attributes of the reference env.
    ``rng_mode``: _abi.RNG_MT19937 (the reference's two generators, default) or _abi.RNG_COUNTER (include/ctf_env.h)."""
    kw = dict(DEFAULT_KWARGS)
    unknown = set(kwargs) - set(kw)
    if unknown:
        raise TypeError(f"GridworldCtf.__init__() got unexpected keyword argument(s) {sorted(unknown)}")
    kw.update(kwargs)
    scenario = kw["SCENARIO"]
    if scenario is None:
        # the reference's generate_map path reads FLAG_POSITIONS before it exists (gridworld_ctf.py:513)
        raise AttributeError("'GridworldCtf' object has no attribute 'FLAG_POSITIONS' (SCENARIO=None is dead in the reference)")

    agent_config = kw["AGENT_CONFIG"]
    n = len(agent_config)
    if sorted(agent_config.keys()) != list(range(n)):
        raise KeyError("AGENT_CONFIG keys must be 0..N-1")
    if n > _abi.MAX_AGENTS:
        raise ValueError(f"at most {_abi.MAX_AGENTS} agents")
    agent_teams = {k: agent_config[k]["team"] for k in agent_config.keys()}
    agent_types = {k: agent_config[k]["type"] for k in agent_config.keys()}
    agent_tile_map = {k: AGENT_TYPE_TILE_MAP[agent_types[k]][agent_teams[k]] for k in agent_config.keys()}
    g = int(scenario["GRID_SIZE"])
    if not (4 <= g <= _abi.MAX_GRID):
        raise ValueError(f"GRID_SIZE must be in 4..{_abi.MAX_GRID}")

    grid = paint_grid(scenario, n, agent_tile_map)
    tiles = tiles_used(grid, agent_types)
    if len(tiles) + 1 > _abi.MAX_CHANNELS:
        raise ValueError("too many observation channels")
    opponents = opponents_of(agent_teams)

    c = _abi.CtfConfig()
    c.abi_version = _abi.ABI_VERSION
    c.n_agents = n
    c.grid_size = g
    c.n_channels = len(tiles) + 1
    c.game_steps = int(kw["GAME_STEPS"])
    flip = scenario["FLIP_AXIS"]
    if flip not in (None, 0, 1, 2):
        raise KeyError(flip)  # REVERSED_ACTION_MAP has no such key
    c.flip_axis = -1 if flip is None else int(flip)
    c.home_flag_capture = int(bool(kw["HOME_FLAG_CAPTURE"]))
    c.use_adjusted_rewards = int(bool(kw["USE_ADJUSTED_REWARDS"]))
    c.drop_flag_when_no_hp = int(bool(kw["DROP_FLAG_WHEN_NO_HP"]))
    c.log_metrics = int(bool(log_metrics))
    c.rng_mode = int(rng_mode)
    for t in (0, 1):
        c.n_opponents[t] = len(opponents[t])
        for k, a in enumerate(opponents[t]):
            c.opponents[t][k] = a
    c.heal_per_step = float(kw["AGENT_HP_HEALING_PER_STEP"])
    c.tag_probability = float(kw["TAG_PROBABILITY"])
    c.guardian_damage_multiplier = float(kw["GUARDIAN_DAMAGE_MULTIPLIER"])
    c.vault_hp_cost = float(kw["VAULT_HP_COST"])
    c.vault_min_hp = float(kw["VAULT_MIN_HP"])
    # constants the reference hard-codes in its constructor (gridworld_ctf.py:75-80)
    c.reward_capture = 1.0
    c.reward_step = 0.0
    c.reward_tag = 0.0
    c.win_margin_scalar = 0.1
    c.loss_margin_scalar = 0.0
    c.opp_capture_punishment = 0.5
    for t in range(4):
        # types absent from the dicts are never indexed by the reference either; 1.0 keeps divisions finite
        c.type_hp[t] = float(kw["AGENT_TYPE_HP"][t]) if t in kw["AGENT_TYPE_HP"] else 1.0
        c.type_damage[t] = float(kw["AGENT_TYPE_DAMAGE"][t]) if t in kw["AGENT_TYPE_DAMAGE"] else 0.0
    for i in range(n):
        if agent_types[i] not in kw["AGENT_TYPE_HP"]:
            raise KeyError(agent_types[i])
        c.agent_team[i] = agent_teams[i]
        c.agent_type[i] = agent_types[i]
        c.start_pos[i][0], c.start_pos[i][1] = scenario["AGENT_STARTING_POSITIONS"][i]
    for t in (0, 1):
        c.flag_pos[t][0], c.flag_pos[t][1] = scenario["FLAG_POSITIONS"][t]
        c.capture_pos[t][0], c.capture_pos[t][1] = scenario["CAPTURE_POSITIONS"][t]
        c.spawn_pos[t][0], c.spawn_pos[t][1] = scenario["SPAWN_POSITIONS"][t]
    for k, tile in enumerate(tiles):
        c.tile_of_channel[k + 1] = int(tile)
    flat = grid.reshape(-1)
    for k in range(g * g):
        c.init_grid[k] = int(flat[k])

    derived = dict(
        kwargs=kw,
        n_agents=n,
        grid_size=g,
        agent_teams=agent_teams,
        agent_types=agent_types,
        agent_tile_map=agent_tile_map,
        tiles_used=tiles,
        opponents=opponents,
        flip_axis=flip,
        init_grid=grid,
    )
    return c, derived
