"""k_step alone at every lane width (CTF_STEP_W) over a range of batch sizes: where does the automatic rule (ctf_kernels.hip:
step_lanes) leave time on the table?  Usage: python tools/step_lanes_sweep.py [arena|split]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("marl-ctf-development_amd")


def step_ms(kw, n_envs, w, metrics=True, reps=200):
    if w:
        os.environ["CTF_STEP_W"] = str(w)
    else:
        os.environ.pop("CTF_STEP_W", None)
    seeds = np.arange(n_envs, dtype=np.uint64) + 11
    vec = pkg.VecGridworldCtf(n_envs, device=0, py_seeds=seeds, np_seeds=seeds, log_metrics=metrics, tune_placement=False, **kw)
    dev = vec.device
    table = torch.empty((8, n_envs, vec.N_AGENTS), dtype=torch.int8, device=dev)
    for k in range(8):
        vec.random_actions(table[k], seed=7, step=k)
    for k in range(64):  # spread the episode phases a little and warm up
        vec.step(table[k % 8], auto_reset=True)
    torch.cuda.synchronize(dev)
    out = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(reps):
            vec.step(table[k % 8], auto_reset=True)
        b.record()
        torch.cuda.synchronize(dev)
        out.append(a.elapsed_time(b) / reps)
    vec.close()
    return sorted(out)[2]


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "arena"
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii) if name == "arena" else dict(pkg.configs.SPLIT_KWARGS, SCENARIO=pkg.CtfScenarios.arrow)
    print(f"{name}: k_step ms per launch (back-to-back launches, median of 5 x 200), by lanes per env", flush=True)
    for n_envs in (4096, 8192, 16384, 32768, 49152, 65536, 131072):
        row = [step_ms(kw, n_envs, w) for w in (0, 1, 2, 4, 8)]
        print(f"{n_envs:7d} envs: auto {row[0]:.4f} | W=1 {row[1]:.4f}  W=2 {row[2]:.4f}  W=4 {row[3]:.4f}  W=8 {row[4]:.4f}", flush=True)
