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


def test_collector_hands_the_full_compact_rollout_over_rccl():
    """BatchedRolloutCollector(handoff=RolloutHandoff): the N > 1 path of BASELINE configs[3]/[4] as far as one GPU allows — a
    one-rank RCCL group with the collectives forced on.  The gathered ("global") tensors must equal the local rollout, chunk
    boundaries and the ragged last chunk included, with and without the observations."""
    import importlib
    import sys, os

    import numpy as np
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _stub_policy import StubCodesPolicy

    pkg = importlib.import_module("marl-ctf-development_amd")
    rollout = importlib.import_module("marl-ctf-development_amd.rollout")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
        E, T = 96, 11
        vec = pkg.VecGridworldCtf(E, device=0, py_seeds=np.arange(E) + 9, np_seeds=np.arange(E) + 9, **kw)
        col = rollout.BatchedRolloutCollector(vec, T, 0)
        a, b = StubCodesPolicy(4, vec.N_CHANNELS, pkg.expand_codes), StubCodesPolicy(6, vec.N_CHANNELS, pkg.expand_codes)
        for with_obs in (False, True):
            ho = sh.RolloutHandoff(1, with_observations=with_obs, force_collective=True)
            out = col.collect(a, b, handoff=ho, handoff_chunk=4)  # chunks of 4, 4 and 3 env steps
            g = out["global"]
            for key in ("rewards", "logprobs", "values", "actions", "use_action_mask", "dones"):
                assert g[key].shape == out[key].shape == (T * 4, E) and torch.equal(g[key], out[key]), key
            assert torch.equal(g["next_done"], out["next_done"])
            if with_obs:
                assert torch.equal(g["grid_codes"], out["grid_codes"]) and torch.equal(g["metadata_states"], out["metadata_states"])
                assert torch.equal(g["next_grid_codes"], out["next_grid_codes"]) and torch.equal(g["next_metadata_state"], out["next_metadata_state"])
            else:
                assert "grid_codes" not in g
        # without a hand-off nothing changes
        assert "global" not in col.collect(a, b)
        vec.close()
    finally:
        dist.destroy_process_group()


def test_data_parallel_ppo_update_over_rccl_with_one_rank():
    """learner.PPOLearner's N-rank path (global permutation from a broadcast seed, advantage statistics and the flat gradient through
    all-reduces) over real RCCL in a one-rank group.  With one rank the global minibatches ARE the local ones, so the result must equal
    the plain single-process update with the same order source:
    * on the float32 stock network (deterministic kernels) to float32 round-off — the normalisation's statistics come from float64
      sums on the N-rank path, nothing else differs;
    * on the native network in deterministic mode (fixed-order gradient reductions) likewise, and two runs are bit-identical;
    * on the native network in its default mode (bf16 MFMA kernels, float atomics in the weight gradients: not run-to-run stable, and
      Adam turns a gradient near zero into a full step of either sign) no further than two single-process runs are from each other."""
    import copy
    import os, sys

    import numpy as np
    import torch.distributed as dist

    learner = importlib.import_module("marl-ctf-development_amd.learner")
    pn = importlib.import_module("marl-ctf-development_amd.policy_native")
    pol = importlib.import_module("marl-ctf-development_amd.policy")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _policy_weights import fill_

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        c, g, m, S, E = 14, 15, 22, 8, 96
        gen = torch.Generator().manual_seed(5)
        r = lambda *shape: torch.rand(*shape, generator=gen)
        codes = torch.randint(0, c, (S, E, g, g), generator=gen).to(torch.uint8)
        rollout = dict(grid_codes=codes, metadata_states=r(S, E, m).half().float(), actions=torch.randint(0, 9, (S, E), generator=gen).float(),
                       use_action_mask=torch.randint(0, 2, (S, E), generator=gen).float(), logprobs=-2.2 + 0.3 * r(S, E), rewards=r(S, E) - 0.4,
                       dones=torch.zeros(S, E), values=0.3 * r(S, E), next_grid_codes=codes[0].clone(), next_metadata_state=r(E, m).half().float(),
                       next_done=torch.zeros(E))
        rollout = {k: v.to(dev) for k, v in rollout.items()}
        args = dict(update_epochs=2, num_minibatches=4)
        params = lambda n: torch.cat([p.detach().reshape(-1) for p in n.parameters()])

        def updated(base, **kw):
            net = copy.deepcopy(base)
            np.random.seed(11)
            losses = learner.PPOLearner(net, c, **kw, **args).update(rollout, micro_batch=100)
            return params(net), losses

        base = fill_(pol.CtfPolicy(9, c, g, m, compute_dtype=torch.float32)).to(dev)
        a, l_dp = updated(base, world=1, rank=0, force_collective=True)
        b, l_sp = updated(base, order="device")
        assert float((a - params(base)).abs().max()) > 1e-4
        assert float((a - b).abs().max()) < 2e-6, float((a - b).abs().max())
        assert np.allclose(l_dp, l_sp, rtol=1e-5, atol=1e-6), (l_dp, l_sp)

        # deterministic mode (fixed-order gradient reductions, tests/test_gpu_deterministic.py): run-to-run identical, and the single
        # process takes its advantage statistics from the same float64 sums as the N-rank path — a one-rank job IS the single process
        base = fill_(pn.CtfPolicyNative(9, c, g, m)).to(dev)
        a, l_dp = updated(base, world=1, rank=0, force_collective=True, deterministic=True)
        a2, l_dp2 = updated(base, world=1, rank=0, force_collective=True, deterministic=True)
        b, l_sp = updated(base, order="device", deterministic=True)
        assert torch.equal(a, a2) and l_dp == l_dp2
        print("deterministic dp vs sp: max", float((a - b).abs().max()), "mean", float((a - b).abs().mean()))
        assert float((a - b).abs().max()) < 2e-6, float((a - b).abs().max())

        a, l_dp = updated(base, world=1, rank=0, force_collective=True)
        b, l_sp = updated(base, order="device")
        b2, l_sp2 = updated(base, order="device")
        spread = float((b - b2).abs().max())
        steps = args["update_epochs"] * args["num_minibatches"] * 2.5e-4  # no parameter can move further than this
        assert float((a - b).abs().max()) <= max(2 * spread, 0.25 * steps), (float((a - b).abs().max()), spread)
        assert float((a - b).abs().mean()) < 2e-5
        # (the last minibatch's losses are taken after seven optimiser steps of slightly different parameters: a few 1e-3 apart between
        # two runs of the very same single-process update)
        run_spread = np.abs(np.array(l_sp) - np.array(l_sp2))
        assert np.all(np.abs(np.array(l_dp) - np.array(l_sp)) <= np.maximum(4 * run_spread, 5e-2 * np.abs(np.array(l_sp)) + 1e-3)), (l_dp, l_sp, l_sp2)
    finally:
        dist.destroy_process_group()
