"""Does a nontemporal hint on the render's stores keep the step kernel's working set (88 MB of grids, records and digest windows per
launch at 65 536 arena envs) in the memory-side cache across the 1.65 GB the render writes?  k_step alone, back to back, takes
0.057 ms; between renders 0.068.  Two handles in one process on the SAME observation buffer, CTF_OBS_NT=0 / 1, interleaved.
Usage: python tools/render_nt_ab.py [envs] [arena|arena20] [tuned|untuned]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("marl-ctf-development_amd")

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "arena"
TUNE = (sys.argv[3] != "untuned") if len(sys.argv) > 3 else True
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii) if WORKLOAD == "arena" else \
    dict(pkg.configs.ARENA20_KWARGS, SCENARIO=pkg.configs.arena20_scenario())
seeds = np.arange(E, dtype=np.uint64) + 11
vecs = {}
for nt in (0, 1, "auto"):
    if nt == "auto":
        os.environ.pop("CTF_OBS_NT", None)
    else:
        os.environ["CTF_OBS_NT"] = str(nt)
    vecs[nt] = pkg.VecGridworldCtf(E, device=0, py_seeds=seeds, np_seeds=seeds, log_metrics=True, tune_placement=(TUNE and nt == 0), **kw)
_ = vecs[0].obs  # the placement search runs here
vecs[1].obs = vecs[0].obs
vecs["auto"].obs = vecs[0].obs
print(WORKLOAD, E, "envs; placement", vecs[0].placement, flush=True)
dev = vecs[0].device
table = torch.empty((8, E, 8), dtype=torch.int8, device=dev)
for k in range(8):
    vecs[0].random_actions(table[k], seed=7, step=k)


def window(vec, reps=100):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * reps + 1)]
    ev[0].record()
    for k in range(reps):
        vec.step(table[k % 8], auto_reset=True)
        ev[3 * k + 1].record()
        vec.observe()
        ev[3 * k + 2].record()
        ev[3 * k + 3].record()
    torch.cuda.synchronize(dev)
    st = np.mean([ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(reps)])
    ob = np.mean([ev[3 * k + 1].elapsed_time(ev[3 * k + 2]) for k in range(reps)])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(reps):
        vec.step_observe(table[k % 8], auto_reset=True)
    b.record()
    torch.cuda.synchronize(dev)
    return st, ob, a.elapsed_time(b) / reps


for nt in (0, 1, "auto"):
    for _ in range(2):
        window(vecs[nt], 50)
for rnd in range(4):
    for nt in (0, 1, "auto"):
        st, ob, whole = window(vecs[nt])
        print(f"round {rnd} CTF_OBS_NT={nt}: k_step {st:.4f} ms  render {ob:.4f} ms (raw events)  step_observe {whole:.4f} ms = {E / whole / 1e3:.1f} M env-steps/s", flush=True)
