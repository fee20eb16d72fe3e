"""Does k_step of one env chunk hide under the render of another?  Splits the bench's 65 536 arena envs into P handles on
S HIP streams and times whole steps (step + observe of every chunk) against the one-handle, one-stream schedule.

    python tools/overlap_probe.py [steps]
"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

pkg = importlib.import_module("marl-ctf-development_amd")


def make(E, lo, kwargs):
    seeds = pkg.sharding.env_seeds(0, lo, lo + E)
    vec = pkg.VecGridworldCtf(E, device=0, py_seeds=seeds, np_seeds=seeds, log_metrics=True, **kwargs)
    acts = torch.empty((64, E, vec.N_AGENTS), dtype=torch.int8, device="cuda")
    for t in range(64):
        vec.random_actions(acts[t], seed=0xC7F, step=t, env_offset=lo)
    vec.observe()
    bench.stagger_phases(vec, torch, lo, kwargs["GAME_STEPS"])
    return vec, acts


def run(parts, streams, K, priority=False):
    E = 65536 // parts
    kwargs = bench.WORKLOADS["arena"][1](pkg)
    vecs = [make(E, i * E, kwargs) for i in range(parts)]
    if priority:
        ss = [torch.cuda.Stream(priority=-1 if i % 2 else 0) for i in range(streams)]
    else:
        ss = [torch.cuda.Stream() for _ in range(streams)]

    def step(t):
        for i, (vec, acts) in enumerate(vecs):
            with torch.cuda.stream(ss[i % streams]):
                vec.step(acts[t % 64], auto_reset=True)
                vec.observe()

    for t in range(10):
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        step(10 + t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    del vecs
    torch.cuda.empty_cache()
    return ms


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for rep in range(2):
        for parts, streams, prio in [(1, 1, False), (2, 1, False), (2, 2, False), (4, 2, False), (4, 4, False), (8, 2, False), (2, 2, True), (4, 2, True)]:
            ms = run(parts, streams, K, prio)
            print(f"parts {parts} streams {streams} prio {int(prio)}: {ms:.4f} ms/step  {65536 / ms / 1e3:.1f} M env-steps/s", flush=True)
