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
        slow = sh.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
        assert slow == float(world)
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
