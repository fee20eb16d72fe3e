"""Distribution of render time / fill time over fresh observation-buffer allocations (the search of VecGridworldCtf._tune_obs_placement
run to the end): where do the kinds lie?  Usage: python tools/placement_hist.py [envs] [arena|arena20] [tries]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("marl-ctf-development_amd")

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "arena"
TRIES = int(sys.argv[3]) if len(sys.argv) > 3 else 120
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii) if WORKLOAD == "arena" else \
    dict(pkg.configs.ARENA20_KWARGS, SCENARIO=pkg.configs.arena20_scenario())
seeds = np.arange(E, dtype=np.uint64) + 11
vec = pkg.VecGridworldCtf(E, device=0, py_seeds=seeds, np_seeds=seeds, tune_placement=False, placement_tries=TRIES, **kw)
vec._placement_seconds = 60.0
vec._tune_obs_placement(good_enough=0.0)
r = np.sort(np.array(vec.placement_probe_ms)) / vec.placement_fill_ms
print(f"{WORKLOAD} {E} envs, CTF_OBS_NT={os.environ.get('CTF_OBS_NT', '1')}: fill {vec.placement_fill_ms:.4f} ms, {len(r)} candidates, render / fill:")
print("  sorted:", " ".join(f"{x:.3f}" for x in r))
print("  " + "  ".join(f"<={b:.2f}: {int((r <= b).sum())}" for b in (0.96, 0.98, 1.0, 1.02, 1.04, 1.06, 1.08, 1.10, 1.12, 1.16, 1.20, 1.30)))
