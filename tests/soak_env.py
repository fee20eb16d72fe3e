#!/usr/bin/env python3
"""Soak test of the env path at full size (not part of the suite): thousands of steps with Philox actions and auto-reset;
every few steps the one-hot render (k_observe) is compared with the expansion of the compact render (k_observe_codes) —
two independent kernels — over all envs, and a 32-env sample is stepped through the CPU oracle beside it."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")  # repo root (this script lives in tests/: it uses the oracle)
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (test infrastructure: this tool is a checker, not the product)

pkg = importlib.import_module("marl-ctf-development_amd")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng_mode = sys.argv[3] if len(sys.argv) > 3 else "mt19937"  # or "counter"
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
seeds = np.arange(E, dtype=np.uint64) + 77
vec = pkg.VecGridworldCtf(E, device=0, py_seeds=seeds, np_seeds=seeds, rng_mode=rng_mode, **kw)
sample = np.linspace(0, E - 1, 32).astype(int)
refs = {int(e): oracle.OracleEnv(vec.cfg) for e in sample}
for e, r in refs.items():
    r.seed(int(seeds[e]), int(seeds[e]))
sidx = torch.from_numpy(sample).cuda()
acts = torch.empty((E, 8), dtype=torch.int8, device="cuda")
bad = 0
t0 = time.time()
for t in range(steps):
    vec.random_actions(acts, seed=4242, step=t)
    rewards, done, obs, meta = vec.step_observe(acts, auto_reset=True, want_f64=True)
    a = acts[sidx].cpu().numpy()
    r64 = vec.rewards64[sidx].cpu().numpy()
    o = obs[sidx].cpu().numpy() if t % 10 == 0 else None
    for k, e in enumerate(sample):
        r = refs[int(e)]
        if r.get_state().done:
            r.reset()
        rw, dn, status = r.step(a[k])
        if status or not np.array_equal(r64[k], rw) or (o is not None and not np.array_equal(o[k], r.observe()[0])):
            bad += 1
    if t % 25 == 0:
        codes, _ = vec.observe_codes(meta=False)
        for lo in range(0, E, 16384):
            if not torch.equal(pkg.expand_codes(codes[lo:lo + 16384], vec.N_CHANNELS), obs[lo:lo + 16384]):
                bad += 1
    if t % 500 == 0:
        print(f"step {t} status {vec.status()} mismatches {bad} elapsed {time.time() - t0:.1f}s", flush=True)
# where the generators stand after all those ring switches: the standard-form MT19937 states (or the words consumed of the Philox tapes)
ctr = vec.get_rng_counters().cpu().numpy() if rng_mode == "counter" else None
for e in sample:
    if rng_mode == "counter":
        bad += tuple(int(x) for x in ctr[int(e)]) != tuple(int(x) for x in refs[int(e)].get_rng_counters())
    else:
        (py, npw), (opy, onp) = vec.get_rng_state(int(e)), refs[int(e)].get_rng_state()
        bad += not (np.array_equal(py, opy) and np.array_equal(npw, onp))
print("done", steps, "steps; status", vec.status(), "mismatches", bad, flush=True)
sys.exit(1 if bad or vec.status() else 0)
