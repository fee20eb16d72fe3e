"""The N>1 path of bench.py on CPU: world_size-2 gloo processes exercise the shard layout, the per-step
rollout all-gather (double-buffered, async) and the max-over-ranks timing rule."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sh = importlib.import_module("marl-ctf-development_amd.sharding")


def test_shard_ranges_partition_the_envs():
    for n, w in [(65536, 8), (262144, 8), (10, 3), (7, 8), (4096, 1)]:
        ranges = [sh.shard_range(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1


def test_env_seeds_are_global_and_unique():
    a = sh.env_seeds(1, 0, 65536)
    b = sh.env_seeds(1, 65536, 131072)
    assert a[0] == 1_000_003 and b[0] == 1_000_003 + 65536
    assert len(np.unique(np.concatenate([a, b]))) == 131072
    assert int(sh.env_seeds(5000, 0, 1)[0]) == (1_000_003 * 5000) % 2 ** 32


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    assert sh.world_from_env() == (rank, rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        E, N = 6, 4
        lo, hi = sh.shard_range(world * E, rank, world)
        rewards = torch.zeros((E, N), dtype=torch.float32)
        done = torch.zeros((E,), dtype=torch.uint8)
        gather = sh.RolloutGather(rewards, done, world)
        out = []
        for t in range(5):  # two gathers may be in flight: the buffers alternate
            rewards = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, N) * 10 + t
            done = ((torch.arange(lo, hi) + t) % 3 == 0).to(torch.uint8)
            slot = gather.start(rewards, done)
            gr, gd = gather.result(slot)
            out.append((gr.clone(), gd.clone()))
        for t, (gr, gd) in enumerate(out):
            want_r = torch.arange(0, world * E, dtype=torch.float32)[:, None].repeat(1, N) * 10 + t
            want_d = ((torch.arange(0, world * E) + t) % 3 == 0).to(torch.uint8)
            assert torch.equal(gr, want_r) and torch.equal(gd, want_d)
        # the chunked variant bench.py uses: steps write straight into the chunk buffer, one gather per 4 steps, a partial last chunk
        cg = sh.ChunkedRolloutGather(E, N, torch.device("cpu"), world, chunk=4)
        T = 10
        got = {}
        for t in range(T):
            r, d = cg.views(t)
            r.copy_(torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, N) * 10 + t)
            d.copy_(((torch.arange(lo, hi) + t) % 3 == 0).to(torch.uint8))
            cg.step_done(t)
            if t % 4 == 3:
                gr, gd = cg.result(t // 4)
                got[t // 4] = (gr.clone(), gd.clone())
        cg.flush(T)
        gr, gd = cg.result(T // 4)
        got[T // 4] = (gr.clone(), gd.clone())
        for t in range(T):
            gr, gd = got[t // 4]
            assert gr.shape == (world, 4, E, N)
            want_r = torch.arange(0, world * E, dtype=torch.float32)[:, None].repeat(1, N) * 10 + t
            want_d = ((torch.arange(0, world * E) + t) % 3 == 0).to(torch.uint8)
            assert torch.equal(gr[:, t % 4].reshape(world * E, N), want_r) and torch.equal(gd[:, t % 4].reshape(world * E), want_d)
        # the collector's hand-off of the FULL compact set (reward, log-prob, value, action, mask + codes / metadata),
        # launched in chunks of slots as a rollout fills, returned in global env order
        S, G, M = 12, 3, 5
        env_ids = torch.arange(lo, hi, dtype=torch.float32)

        def slot_major(scale, dtype=torch.float32):  # value of slot s, global env g = scale * (100 * s + g)
            return (scale * (100.0 * torch.arange(S, dtype=torch.float32)[:, None] + env_ids[None, :])).to(dtype)

        local = dict(rewards=slot_major(1.0), logprobs=slot_major(-0.5), values=slot_major(0.25),
                     actions=(slot_major(1.0) % 9), use_action_mask=(slot_major(1.0) % 2),
                     grid_codes=(slot_major(1.0) % 251).to(torch.uint8)[:, :, None, None].repeat(1, 1, G, G),
                     metadata_states=slot_major(0.5)[:, :, None].repeat(1, 1, M))
        for with_obs in (False, True):
            ho = sh.RolloutHandoff(world, with_observations=with_obs)
            for a, b in ((0, 4), (4, 8), (8, 12)):
                ho.launch(a, b, local)
            got = ho.result()
            all_ids = torch.arange(0, world * E, dtype=torch.float32)
            full = 100.0 * torch.arange(S, dtype=torch.float32)[:, None] + all_ids[None, :]
            assert torch.equal(got["rewards"], full) and torch.equal(got["logprobs"], -0.5 * full) and torch.equal(got["values"], 0.25 * full)
            assert torch.equal(got["actions"], full % 9) and torch.equal(got["use_action_mask"], full % 2)
            assert got["actions"].dtype == torch.float32 and got["rewards"].shape == (S, world * E)
            if with_obs:
                assert torch.equal(got["grid_codes"], (full % 251).to(torch.uint8)[:, :, None, None].repeat(1, 1, G, G))
                assert torch.equal(got["metadata_states"], (0.5 * full)[:, :, None].repeat(1, 1, M))
            else:
                assert "grid_codes" not in got
            nd = ho.gather_once((env_ids % 2).clone())
            assert torch.equal(nd, all_ids % 2)
        # a sentinel decision / action outside 0..255 survives the trip (the pack is int32), and a hand-off that launched nothing
        # returns empty tensors of the right trailing shape instead of raising
        ho = sh.RolloutHandoff(world)
        ho.launch(0, 4, dict(local, use_action_mask=local["use_action_mask"] - 1.0, actions=local["actions"] + 300.0))
        got = ho.result()
        assert torch.equal(got["use_action_mask"], (full % 2)[:4] - 1.0) and torch.equal(got["actions"], (full % 9)[:4] + 300.0)
        empty = sh.RolloutHandoff(world)
        empty.launch(0, 0, local)
        got = empty.result()
        assert got["rewards"].shape == (0, world * E) and got["actions"].shape == (0, world * E)
        slow = sh.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
        assert slow == float(world)
        # bench.py's self-check of a multi-GPU line: every rank of the job reported one positive time
        seen, times = sh.gather_rank_times(rank, 1.0 + rank, world)
        assert sh.ranks_complete(seen, times, world) and sorted(seen) == list(range(world))
        assert not sh.ranks_complete(seen[:-1], times[:-1], world) and not sh.ranks_complete(seen, [0.0] + times[1:], world)
        assert not sh.ranks_complete([0] * world, times, world)
        q.put((rank, "ok"))
    except Exception as exc:  # pragma: no cover
        q.put((rank, repr(exc)))
    finally:
        dist.destroy_process_group()


def test_rollout_gather_and_timing_rule_world2_gloo():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "ok"), (1, "ok")], res
