"""Environment kwargs of the two BASELINE configurations, as the reference's experiment scripts pass
them (0_the_split.py:33-61 and 8_arena.py:33-63), minus SCENARIO (take it from ``maps.CtfScenarios`` or
from the reference's own ``scenarios.CtfScenarios``)."""

SPLIT_KWARGS = {  # 0_the_split: arrow map, 2v2
    "GRID_SIZE": 11,
    "AGENT_CONFIG": {
        0: {"team": 0, "type": 1},
        1: {"team": 1, "type": 0},
        2: {"team": 0, "type": 0},
        3: {"team": 1, "type": 0},
    },
    "GAME_STEPS": 500,
    "USE_ADJUSTED_REWARDS": True,
    "MAP_SYMMETRY_CHECK": False,
    "AGENT_TYPE_HP": {0: 10, 1: 8, 2: 8, 3: 7},
    "AGENT_TYPE_DAMAGE": {0: 1, 1: 0.5, 2: 0.5, 3: 1},
    "GUARDIAN_DAMAGE_MULTIPLIER": 5.0,
    "VAULT_HP_COST": 1.25,
}

ARENA_KWARGS = {  # 8_arena: arena_iii map, 4v4 heterogeneous
    "GRID_SIZE": 15,
    "AGENT_CONFIG": {
        0: {"team": 0, "type": 1},
        1: {"team": 1, "type": 1},
        2: {"team": 0, "type": 2},
        3: {"team": 1, "type": 2},
        4: {"team": 0, "type": 3},
        5: {"team": 1, "type": 3},
        6: {"team": 0, "type": 0},
        7: {"team": 1, "type": 0},
    },
    "GAME_STEPS": 500,
    "USE_ADJUSTED_REWARDS": True,
    "MAP_SYMMETRY_CHECK": True,
    "AGENT_TYPE_HP": {0: 10, 1: 8, 2: 8, 3: 7},
    "AGENT_TYPE_DAMAGE": {0: 1, 1: 0.5, 2: 0.5, 3: 1},
    "GUARDIAN_DAMAGE_MULTIPLIER": 5.0,
    "VAULT_HP_COST": 1.25,
}


# BASELINE.json words the target as "8_arena (4v4, 20x20 grid)"; the reference's arena maps are 15x15.  For that wording:
# the 8_arena agent table and rules on a SYNTHETIC point-symmetric 20x20 map of this build's own (the same map the
# syn_arena20 golden trajectory was recorded on from the reference env).  bench.py --workload arena20.
ARENA20_ROWS = (
    "....................",
    "....................",
    "....................",
    "##......+#.#+.......",
    "....................",
    "....#.....+.........",
    "......+......+......",
    "...............#....",
    "........+......#....",
    "++.......#..........",
    "..........#.......++",
    "....#......+........",
    "....#...............",
    "......+......+......",
    ".........+.....#....",
    "....................",
    ".......+#.#+......##",
    "....................",
    "....................",
    "....................",
)


def arena20_scenario():
    try:
        from .maps import _scenario
    except ImportError:  # pragma: no cover
        from maps import _scenario
    return _scenario("SynArena20", None, ARENA20_ROWS, flags=((3, 10), (16, 9)), captures=((3, 10), (16, 9)),
                     spawns=((7, 17), (12, 2)),
                     starts=((7, 16), (12, 3), (8, 17), (11, 2), (6, 17), (13, 2), (7, 18), (12, 1)))


ARENA20_KWARGS = dict(ARENA_KWARGS, GRID_SIZE=20)
