"""Batched duel: the counterpart of ``utils.duel`` (reference utils.py:500-573) for E envs at once.

Per env it reproduces the reference loop: ``reset()``; every step each agent observes (team 0 raw, team 1 flipped),
team-0 agents are driven by ``agent`` and team-1 agents by ``opponent`` whose actions are mapped back through
``REVERSED_ACTION_MAP`` (utils.py:535-551); the loop ends when the env reports ``done`` or after ``max_steps + 1``
steps (``step_count > max_steps``, utils.py:559).  All envs of a batch share GAME_STEPS and start together, so they
all stop at the same step.  Returns what the reference returns per env: the result sign (utils.py:562-569) and the
counters ``env.metrics`` holds (as tensors; ``VecGridworldCtf.counters``).
"""
try:
    from .rollout import BatchedRolloutCollector
except ImportError:  # pragma: no cover
    from rollout import BatchedRolloutCollector


def batched_duel(vec, agent, opponent, max_steps=256):
    """-> dict(result int8 [E] (+1 team 0 wins, 0 draw, -1 team 1 wins), team_flag_captures int32 [E, 2],
    metrics int32 [E, 13, N], steps int)."""
    import torch

    col = BatchedRolloutCollector(vec, 1, 0)  # reuses the policy plumbing; its rollout buffers hold one step
    use_codes = col.use_codes(agent, opponent)  # policies with a compact-observation path (policy_native.py) get the code bytes
    game_steps = int(vec.cfg.game_steps)
    n_steps = min(game_steps, int(max_steps) + 1)
    vec.reset()
    with torch.no_grad():
        for _ in range(n_steps):
            vec.step(col.joint_actions(agent, opponent, use_codes)[1])
    metrics, caps, _ = vec.counters()
    result = torch.sign(caps[:, 0] - caps[:, 1]).to(torch.int8)
    return dict(result=result, team_flag_captures=caps, metrics=metrics, steps=n_steps)


def duel_trajectory(vec, agent, opponent, env_index=0, max_steps=256):
    """The three.js viewer record of ``utils.duel_json`` (reference utils.py:728-815) for ONE env of a batched duel:
    static map, per-step position deltas + ``has_flag``, destructible tiles and scores — the same dict, key for key
    (the reference also dumps it to a file; pass the result to ``json.dump``)."""
    import numpy as np
    import torch

    col = BatchedRolloutCollector(vec, 1, 0)
    use_codes = col.use_codes(agent, opponent)
    n, g = vec.N_AGENTS, vec.GRID_SIZE
    scen = vec.derived["kwargs"]["SCENARIO"]
    vec.reset()

    def snapshot():
        v = vec.get_state(env_index)
        grid = np.frombuffer(v.grid, dtype=np.uint8, count=g * g).reshape(g, g)
        pos = [(int(v.pos[i][0]), int(v.pos[i][1])) for i in range(n)]
        return grid, pos, [int(v.has_flag[i]) for i in range(n)], (int(v.team_captures[0]), int(v.team_captures[1]))

    def tiles_of(grid):
        return [{"x": int(x), "z": int(z), "type": 0} for z, x in zip(*np.where(grid == 2))] + \
               [{"x": int(x), "z": int(z), "type": 1} for z, x in zip(*np.where(grid == 3))]

    grid, pos, _, _ = snapshot()
    out = {
        "grid_size": g,
        "flag_pos": {f"{k}": {"x": v[1], "z": v[0]} for k, v in scen["FLAG_POSITIONS"].items()},
        "spawn_pos": {f"{k}": {"x": v[1], "z": v[0]} for k, v in scen["SPAWN_POSITIONS"].items()},
        "agent_config": [{"team": vec.AGENT_TEAMS[i], "type": vec.AGENT_TYPES[i], "start_x": scen["AGENT_STARTING_POSITIONS"][i][1],
                          "start_z": scen["AGENT_STARTING_POSITIONS"][i][0]} for i in range(n)],
        "block_tiles": [{"x": int(x), "z": int(z)} for z, x in zip(*np.where(grid == 1))],
        "destructible_tiles": tiles_of(grid),
    }
    movement, tiles, scores = [], [], []
    n_steps = min(int(vec.cfg.game_steps), int(max_steps) + 1)
    with torch.no_grad():
        for _ in range(n_steps):
            vec.step(col.joint_actions(agent, opponent, use_codes)[1])
            grid, new_pos, has_flag, caps = snapshot()
            movement.append([{"x": new_pos[i][1] - pos[i][1], "z": new_pos[i][0] - pos[i][0], "has_flag": has_flag[i]} for i in range(n)])
            tiles.append(tiles_of(grid))
            scores.append([{"t0": caps[0], "t1": caps[1]}])
            pos = new_pos
    out["movement"], out["tiles"], out["scores"] = movement, tiles, scores
    return out
