#!/usr/bin/env python3
"""End-to-end self-play throughput (BASELINE.json configs[4], "ppo.py self-play on 8_arena ... end-to-end steps/sec"): the batched env
+ two policy networks of the reference's architecture (random weights) collecting a rollout, then the reference's learner — GAE + PPO
update (ppo.py:133-242 semantics, marl-ctf-development_amd/learner.py) — consuming the compact rollout.  Secondary benchmark: the
graded one is bench.py, which carries this one's 1-GPU result in its `secondary` block.

    python bench_rollout.py --envs 16384 --steps 16
    python bench_rollout.py --gpus 8 --envs 65536 --steps 16      # starts its own 8 ranks (bench.world_or_launch); --envs is PER GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 bench_rollout.py --gpus 8 ...

N ranks (one process per GPU; the reference: one Ray task per env, ppo.py:264-266,349-376, all rollouts concatenated into one update,
:359-376): rank r collects the rollout of global envs [r * E, (r + 1) * E) with NO exchange — the networks' parameters are identical on
every rank — and the update is data-parallel over the global minibatches: one flat gradient all-reduce (RCCL) per optimiser step
(learner.PPOLearner(world=N)).  Timing: barrier + synchronize on both sides of the iteration, MAX over ranks; `value` = all ranks'
env-steps / that time.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _sync(torch, dist, device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize(device)


def run(envs=16384, steps=16, policy="native", dtype="bf16", update=True, update_epochs=4, num_minibatches=4, device=0, micro_batch=1 << 20,
        rank=0, world=1, dist=None, run_seed=1, order=None, force_dp=False, deterministic=None):
    """-> dict: rollout / update / total env-steps per second of one PPO iteration (rollout of `steps` env steps of `envs`
    envs PER RANK, then `update_epochs` x `num_minibatches` minibatch updates over the world * envs * steps * 4 samples)."""
    import torch

    pkg = importlib.import_module("marl-ctf-development_amd")
    learner = importlib.import_module("marl-ctf-development_amd.learner")
    sh = pkg.sharding
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    lo = rank * envs
    seeds = sh.env_seeds(run_seed, lo, lo + envs)  # functions of the GLOBAL env index
    vec = pkg.VecGridworldCtf(envs, device=device, py_seeds=seeds, np_seeds=seeds, **kw)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    dev = torch.device("cuda", device)
    if policy == "native":
        nets = [pkg.policy_native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN).to(dev) for _ in range(2)]
    else:
        nets = [pkg.policy.CtfPolicy(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN, compute_dtype=dt).to(dev) for _ in range(2)]
    if world > 1:  # the same two networks on every rank; every rank samples its actions from its own stream
        for k, net in enumerate(nets):
            learner.broadcast_module(net)
            if hasattr(net, "reseed"):
                net.reseed((0x5EED0000 + 2 * rank + k) * 0x9E3779B97F4A7C15)
    if policy == "native":
        for net in nets:
            net.prepare()
    log = lambda msg: print(f"[bench_rollout {time.strftime('%H:%M:%S')} rank {rank}] {msg}", file=sys.stderr, flush=True)
    col = pkg.BatchedRolloutCollector(vec, steps, 0)
    if os.environ.get("CTF_ROLLOUT_OVERLAP") == "1":  # A/B only: the opponent's conv front on a second stream (measured slower)
        col.overlap_teams = True
    warm = pkg.BatchedRolloutCollector(vec, min(steps, 4), 0) if steps > 16 else col
    # Warm-up: kernel selection, buffer placement — and the clocks.  A process that times its first or second rollout reads 15-21 M
    # env-steps/s where the ninth reads 31 M (round 5: three of four cold standalone runs; inside bench.py, after 20 s of env work, the
    # same call reads 31 M every time).  So: warm-up rollouts until two in a row take the same time within 3 % (at most twelve).
    prev, n_warm = None, 0
    max_warm = int(os.environ.get("CTF_ROLLOUT_WARMUPS", 12))
    while n_warm < max_warm:
        torch.cuda.synchronize(dev)
        tw = time.perf_counter()
        warm.collect(*nets)
        torch.cuda.synchronize(dev)
        tw = time.perf_counter() - tw
        n_warm += 1
        if prev is not None and abs(tw - prev) <= 0.03 * prev:
            break
        prev = tw
    del warm
    col.preallocate(*nets)  # the rollout buffer exists before the timed rollout, as in any loop that collects more than once (a
    torch.cuda.synchronize(dev)  # hipMalloc of 29 GB inside the timed region read anything from 1.05 to 2.6 s for the 500-step rollout)
    log(f"warm-up done ({n_warm} rollouts)")
    _sync(torch, dist, dev)
    t0 = time.perf_counter()
    out = col.collect(*nets)
    _sync(torch, dist, dev)
    rollout_s = sh.max_over_ranks(time.perf_counter() - t0, dev, world)
    # env-only time for the same number of steps
    acts = torch.zeros((envs, vec.N_AGENTS), dtype=torch.int8, device=dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        vec.observe_codes() if policy == "native" else vec.observe()
        vec.step(acts)
    torch.cuda.synchronize()
    env_s = time.perf_counter() - t1
    res = {
        "metric": "end-to-end self-play env-steps/sec (env + 2 policy networks" + (" + GAE + PPO update)" if update else ")"),
        "unit": "env-steps/s", "n_gpus": world, "envs": envs, "envs_per_gpu": envs, "global_envs": world * envs, "steps": steps, "policy": policy,
        "policy_dtype": "bf16" if policy == "native" else dtype,
        "rollout_env_steps_per_s": world * envs * steps / rollout_s, "rollout_s": rollout_s,
        "policy_samples_per_sec": world * envs * steps * vec.N_AGENTS / rollout_s, "env_share_of_rollout_time": env_s / rollout_s,
        "rollout_bytes_per_gpu": int(sum(t.numel() * t.element_size() for t in out.values() if hasattr(t, "numel"))),
        "ranks_share_one_device": bool(os.environ.get("CTF_BENCH_ONE_DEVICE")),
    }
    if update:
        import copy

        samples = envs * steps * (vec.N_AGENTS // 2)  # per rank
        micro = min(micro_batch, samples // num_minibatches)
        # A minibatch is evaluated in pieces (learner.optimise(micro_batch=): the same update, gradients accumulated).  The warm-up runs
        # one such piece untimed (hipBLASLt / MIOpen pick their kernels on first use of a shape).  In the 65 536 envs x 16 steps iteration
        # (minibatches of 1 048 576 samples): 16.8 / 17.7 / 18.4 M sample-passes/s in pieces of 262 144 / 524 288 / the whole minibatch
        log(f"rollout {rollout_s:.3f} s; warm-up piece of {micro} samples ...")
        throwaway = learner.PPOLearner(copy.deepcopy(nets[0]), vec.N_CHANNELS, update_epochs=1, num_minibatches=1)  # local: no collective
        adv, ret = throwaway.advantages(out)
        flat = lambda t: t.reshape((-1,) + tuple(t.shape[2:]))[:micro]
        grids = out["grid_codes"] if "grid_codes" in out else out["grid_states"]
        throwaway.optimise(flat(grids), flat(out["metadata_states"]), flat(out["logprobs"]), flat(out["actions"]), flat(out["use_action_mask"]),
                           flat(adv), flat(ret), flat(out["values"]))
        del throwaway, adv, ret
        log("warm-up done; timed update ...")
        lrn = learner.PPOLearner(nets[0], vec.N_CHANNELS, world=world, rank=rank, order=order, update_epochs=update_epochs,
                                 num_minibatches=num_minibatches, force_collective=force_dp, deterministic=deterministic)
        _sync(torch, dist, dev)
        t2 = time.perf_counter()
        losses = lrn.update(out, micro_batch=micro)  # (no progress callback: it would read the losses back after every minibatch)
        _sync(torch, dist, dev)
        update_s = sh.max_over_ranks(time.perf_counter() - t2, dev, world)
        res.update({
            "update_s": update_s, "update_samples": world * samples, "update_epochs": update_epochs, "num_minibatches": num_minibatches,
            "micro_batch": micro, "minibatch_order": lrn.order, "deterministic": lrn.deterministic,
            "update_sample_passes_per_s": world * samples * update_epochs / update_s,
            "value": world * envs * steps / (rollout_s + update_s), "learner_share_of_time": update_s / (rollout_s + update_s),
            "losses_v_pg_entropy": [float(x) for x in losses],
            "peak_device_memory_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
            "note": "the learner is the reference's PPO update (ppo.py:174-242) on the compact rollout: the native conv front as the forward "
                    "(ctf_policy_features_train), its whole backward native (ctf_policy_front_backward), the small dense layers' weight / bias "
                    "gradients native (ctf_policy_linear_wgrad), hipBLASLt GEMMs for fc1 and the dense layers' forward / data gradients" +
                    ("; data-parallel over the global minibatches, one flat gradient all-reduce per optimiser step" if lrn.dp else ""),
        })
    else:
        res["value"] = res["rollout_env_steps_per_s"]
    vec.close()
    return res


def dryrun(rank, world, dist, steps):
    """CTF_BENCH_DRYRUN: the N-rank iteration with no kernel behind it — a synthetic compact rollout per rank and the data-parallel
    learner on the CPU policy over gloo: launcher, rendezvous, parameter broadcast, the global-minibatch update and the line."""
    import torch

    learner = importlib.import_module("marl-ctf-development_amd.learner")
    policy = importlib.import_module("marl-ctf-development_amd.policy")
    c, g, m, e = 6, 7, 10, 6
    gen = torch.Generator().manual_seed(50 + rank)
    s = steps * 2
    r = lambda *shape: torch.rand(*shape, generator=gen)
    codes = torch.randint(0, c, (s, e, g, g), generator=gen).to(torch.uint8)
    out = dict(grid_codes=codes, metadata_states=r(s, e, m), actions=torch.randint(0, 9, (s, e), generator=gen).float(),
               use_action_mask=torch.zeros(s, e), logprobs=-2.2 + 0.1 * r(s, e), rewards=r(s, e), dones=torch.zeros(s, e), values=r(s, e),
               next_grid_codes=codes[0].clone(), next_metadata_state=r(e, m), next_done=torch.zeros(e))
    torch.manual_seed(rank)
    net = policy.CtfPolicy(9, c, g, m)
    lrn = learner.PPOLearner(net, c, world=world, rank=rank, update_epochs=2, num_minibatches=2)
    losses = lrn.update(out)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    digest = float(flat.double().sum())
    got = [None] * world
    if world > 1:
        dist.all_gather_object(got, (rank, digest))
    else:
        got = [(rank, digest)]
    return {"metric": "DRY RUN (CTF_BENCH_DRYRUN): no kernel ran", "n_gpus": world, "ranks_seen": sorted(x[0] for x in got),
            "parameters_identical_on_all_ranks": len({x[1] for x in got}) == 1, "losses_v_pg_entropy": [float(x) for x in losses],
            "update_samples": world * s * e, "value": 0.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--envs", type=int, default=16384, help="envs PER GPU")
    ap.add_argument("--steps", type=int, default=16, help="env steps per rollout (the reference's num_steps: 500 = one whole episode, 8_arena.py:74)")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--policy", choices=["native", "torch"], default="native",
                    help="native: compact observation + MFMA conv front (policy_native.py); torch: one-hot planes + stock PyTorch modules")
    ap.add_argument("--no-update", action="store_true", help="rollout only (round 1's figure)")
    ap.add_argument("--update-epochs", type=int, default=4)
    ap.add_argument("--num-minibatches", type=int, default=4)
    ap.add_argument("--micro-batch", type=int, default=1 << 20, help="samples per forward / backward piece of a minibatch")
    ap.add_argument("--order", choices=["numpy", "device"], default=None, help="minibatch order: np.random.shuffle on the host (the reference's) or "
                    "torch.randperm on the device (default with several ranks)")
    ap.add_argument("--deterministic", action="store_true", help="fixed-order gradient reductions (PPOLearner(deterministic=True); also CTF_DETERMINISTIC=1)")
    args = ap.parse_args()

    import bench  # the launch rule of --gpus N lives there (stdlib only: the launcher never touches HIP)

    rc = bench.world_or_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    sys.stdout.flush()
    json_fd = os.dup(1)  # stdout carries exactly one JSON line (RCCL's banner goes to fd 1 as well)
    os.dup2(2, 1)

    import torch

    sh = importlib.import_module("marl-ctf-development_amd.sharding")
    rank, local_rank, world = sh.world_from_env()
    is_dry = bool(os.environ.get("CTF_BENCH_DRYRUN"))
    dist = None
    if world > 1 or os.environ.get("CTF_FORCE_DIST"):
        import torch.distributed as dist

        bench.single_rank_rendezvous()
        if os.environ.get("CTF_BENCH_ONE_DEVICE"):  # rehearsal of the N-rank iteration on a one-GPU box: every rank on device 0, gloo
            local_rank = 0
        if is_dry or os.environ.get("CTF_BENCH_ONE_DEVICE"):
            if not is_dry:
                torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            if local_rank >= torch.cuda.device_count():
                raise SystemExit(f"rank {rank}: local rank {local_rank} has no device ({torch.cuda.device_count()} visible)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if is_dry:
        res = dryrun(rank, world, dist, args.steps)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench_rollout.py needs a GPU: there is no CPU fallback")
        res = run(args.envs, args.steps, args.policy, args.dtype, not args.no_update, args.update_epochs, args.num_minibatches,
                  device=local_rank, micro_batch=args.micro_batch, rank=rank, world=world, dist=dist, order=args.order,
                  force_dp=dist is not None and world == 1, deterministic=True if args.deterministic else None)  # CTF_FORCE_DIST: the N-rank update's collectives over RCCL with one rank
    if rank == 0:
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
