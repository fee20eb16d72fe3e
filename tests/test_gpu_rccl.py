"""The rollout exchange over real RCCL (backend "nccl"), as far as one GPU allows: a one-rank process group, the collective
forced on, both gather variants — the launch, stream ordering and buffer rotation of the N > 1 path of bench.py."""
import importlib
import socket

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sh = importlib.import_module("marl-ctf-development_amd.sharding")


def test_rollout_gathers_over_rccl_with_one_rank():
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        E, N, C = 4096, 8, 4
        cg = sh.ChunkedRolloutGather(E, N, dev, 1, chunk=C, force_collective=True)
        base = torch.arange(E * N, dtype=torch.float32, device=dev).reshape(E, N)
        for t in range(2 * C + 1):
            r, d = cg.views(t)
            r.copy_(base + t)            # what the step kernel would write
            d.fill_(t % 2)
            cg.step_done(t)
            if t % C == C - 1:
                gr, gd = cg.result(t // C)
                assert gr.shape == (1, C, E, N)
                for i in range(C):
                    assert torch.equal(gr[0, i], base + (t - C + 1 + i)) and int(gd[0, i, 0]) == (t - C + 1 + i) % 2
        cg.flush(2 * C + 1)
        gr, gd = cg.result(2)
        assert torch.equal(gr[0, 0], base + 2 * C)
        # the per-step variant
        rewards = torch.zeros((E, N), dtype=torch.float32, device=dev)
        done = torch.zeros((E,), dtype=torch.uint8, device=dev)
        g = sh.RolloutGather(rewards, done, 1, force_collective=True)
        for t in range(5):
            rewards.copy_(base * 2 + t)
            done.fill_(t % 2)
            slot = g.start(rewards, done)
            rewards.zero_()              # the env overwrites its buffers while the gather may still be reading its snapshot
            gr, gd = g.result(slot)
            assert torch.equal(gr, base * 2 + t) and int(gd[0]) == t % 2
        assert sh.max_over_ranks(1.5, dev, 2) == 1.5  # the all-reduce of the timing rule (forced through the collective)
    finally:
        dist.destroy_process_group()
