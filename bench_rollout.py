#!/usr/bin/env python3
"""End-to-end self-play throughput on one GPU (BASELINE.json configs[4], "ppo.py self-play on 8_arena ... end-to-end
steps/sec"): the batched env + two policy networks of the reference's architecture (random weights) collecting a
rollout, then the reference's learner — GAE + PPO update (ppo.py:133-242 semantics, marl-ctf-development_amd/learner.py,
stock PyTorch) — consuming the compact rollout.  Secondary benchmark: the graded one is bench.py, which carries this
one's result in its `secondary` block.

    python bench_rollout.py --envs 16384 --steps 16
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def run(envs=16384, steps=16, policy="native", dtype="bf16", update=True, update_epochs=4, num_minibatches=4, device=0, micro_batch=1 << 20):
    """-> dict: rollout / update / total env-steps per second of one PPO iteration (rollout of `steps` env steps of `envs`
    envs, then `update_epochs` x `num_minibatches` minibatch updates over its envs * steps * 4 samples)."""
    import torch

    pkg = importlib.import_module("marl-ctf-development_amd")
    learner = importlib.import_module("marl-ctf-development_amd.learner")
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    vec = pkg.VecGridworldCtf(envs, device=device, **kw)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    dev = torch.device("cuda", device)
    if policy == "native":
        nets = [pkg.policy_native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN).to(dev).prepare() for _ in range(2)]
    else:
        nets = [pkg.policy.CtfPolicy(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN, compute_dtype=dt).to(dev) for _ in range(2)]
    log = lambda msg: print(f"[bench_rollout {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)
    col = pkg.BatchedRolloutCollector(vec, steps, 0)
    if os.environ.get("CTF_ROLLOUT_OVERLAP") == "1":  # A/B only: the opponent's conv front on a second stream (measured slower)
        col.overlap_teams = True
    col.collect(*nets)  # warm-up (MIOpen kernel selection, buffer placement)
    log("warm-up rollout done")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = col.collect(*nets)
    torch.cuda.synchronize()
    rollout_s = time.perf_counter() - t0
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
        "unit": "env-steps/s", "n_gpus": 1, "envs": envs, "steps": steps, "policy": policy,
        "policy_dtype": "bf16" if policy == "native" else dtype,
        "rollout_env_steps_per_s": envs * steps / rollout_s, "rollout_s": rollout_s,
        "policy_samples_per_sec": envs * steps * vec.N_AGENTS / rollout_s, "env_share_of_rollout_time": env_s / rollout_s,
    }
    if update:
        import copy

        samples = envs * steps * (vec.N_AGENTS // 2)
        micro = min(micro_batch, samples // num_minibatches)
        # A minibatch is evaluated in pieces (learner.optimise(micro_batch=): the same update, gradients accumulated).  MIOpen
        # compiles its convolution kernels on first use of every (batch, C, H, W) shape (up to a minute on a fresh box), so the
        # warm-up runs one such piece untimed.  In this iteration (65 536 envs x 16 steps: minibatches of 1 048 576 samples): 16.8 / 17.7 /
        # 18.4 M sample-passes/s in pieces of 262 144 / 524 288 / the whole minibatch (36 GB of activations)
        log(f"rollout {rollout_s:.3f} s; warm-up piece of {micro} samples (MIOpen compiles its kernels) ...")
        throwaway = learner.PPOLearner(copy.deepcopy(nets[0]), vec.N_CHANNELS, update_epochs=1, num_minibatches=1)
        adv, ret = throwaway.advantages(out)
        flat = lambda t: t.reshape((-1,) + tuple(t.shape[2:]))[:micro]
        grids = out["grid_codes"] if "grid_codes" in out else out["grid_states"]
        throwaway.optimise(flat(grids), flat(out["metadata_states"]), flat(out["logprobs"]), flat(out["actions"]), flat(out["use_action_mask"]),
                           flat(adv), flat(ret), flat(out["values"]))
        del throwaway
        log("warm-up done; timed update ...")
        lrn = learner.PPOLearner(nets[0], vec.N_CHANNELS, update_epochs=update_epochs, num_minibatches=num_minibatches)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        losses = lrn.update(out, micro_batch=micro)  # (no progress callback: it would read the losses back after every minibatch)
        torch.cuda.synchronize()
        update_s = time.perf_counter() - t2
        res.update({
            "update_s": update_s, "update_samples": samples, "update_epochs": update_epochs, "num_minibatches": num_minibatches, "micro_batch": micro,
            "update_sample_passes_per_s": samples * update_epochs / update_s,
            "value": envs * steps / (rollout_s + update_s), "learner_share_of_time": update_s / (rollout_s + update_s),
            "losses_v_pg_entropy": [float(x) for x in losses],
            "note": "the learner is the reference's PPO update (ppo.py:174-242) on the compact rollout: the native conv front as the forward "
                    "(ctf_policy_features_train), native data- and weight-gradient kernels (ctf_policy_front_dgrad / _wgrad), hipBLASLt GEMMs "
                    "for the dense layers (policy_native._NativeFront); it still dominates the iteration",
        })
    else:
        res["value"] = res["rollout_env_steps_per_s"]
    vec.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--policy", choices=["native", "torch"], default="native",
                    help="native: compact observation + MFMA conv front (policy_native.py); torch: one-hot planes + stock PyTorch modules")
    ap.add_argument("--no-update", action="store_true", help="rollout only (round 1's figure)")
    ap.add_argument("--update-epochs", type=int, default=4)
    ap.add_argument("--num-minibatches", type=int, default=4)
    ap.add_argument("--micro-batch", type=int, default=1 << 20, help="samples per forward / backward piece of a minibatch")
    args = ap.parse_args()
    print(json.dumps(run(args.envs, args.steps, args.policy, args.dtype, not args.no_update, args.update_epochs, args.num_minibatches, micro_batch=args.micro_batch)))


if __name__ == "__main__":
    main()
