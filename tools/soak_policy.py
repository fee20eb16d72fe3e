#!/usr/bin/env python3
"""Soak test of the native policy path (not part of the suite): many env steps at full size with two networks in the loop;
every few steps the front kernels are re-run and compared (per-agent kernel against itself, shared-view kernel against the
per-agent rows), the env's status bits are read, and outputs are checked for finiteness."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("marl-ctf-development_amd")
native = pkg.policy_native
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(E, device=0, **kw)
nets = [native.CtfPolicyNative(9, vec.N_CHANNELS, 15, vec.META_LEN, seed=s).cuda() for s in (1, 2)]
col = pkg.BatchedRolloutCollector(vec, 1, 0)
real = torch.from_numpy(native.act_column_order(15, vec.META_LEN) >= 0).cuda()
teams = [[i for i in range(8) if vec.AGENT_TEAMS[i] == t] for t in (0, 1)]
t0 = time.time()
bad = 0
with torch.no_grad():
    for t in range(steps):
        (a_act, a_lp, a_val, _, _, _), env_act = col.joint_actions(nets[0], nets[1], True)
        vec.step(env_act, auto_reset=True)
        if t % 50 == 0:
            codes, meta = vec.observe_codes()
            for tm in (0, 1):
                f = nets[tm].features_from_codes(codes, meta, teams[tm])
                g = nets[tm].features_from_codes(codes, meta, teams[tm])
                h = nets[tm].features_from_codes(codes, meta, teams[tm], shared_view=True, self_cells=vec.self_cells)
                ok = torch.equal(f, g) and torch.equal(f[:, real], h[:, real]) and bool(torch.isfinite(a_val).all()) and bool(torch.isfinite(a_lp).all())
                bad += 0 if ok else 1
            if t % 500 == 0:
                print(f"step {t} status {vec.status()} mismatches {bad} elapsed {time.time() - t0:.1f}s", flush=True)
print("done", steps, "steps; status", vec.status(), "mismatches", bad, flush=True)
sys.exit(1 if bad or vec.status() else 0)
