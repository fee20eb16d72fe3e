"""Plain calls against hipGraph replays of the same calls (tests/test_gpu_hipgraph.py holds the parity side): ms per
ctf_step_observe for the bench workloads, K steps per graph.  Usage: python tools/graph_replay_bench.py [K]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("marl-ctf-development_amd")


def run(name, kw, n_envs, k_steps, reps=40):
    seeds = np.arange(n_envs, dtype=np.uint64) + 11
    vec = pkg.VecGridworldCtf(n_envs, device=0, py_seeds=seeds, np_seeds=seeds, log_metrics=False, **kw)
    dev = vec.device
    table = torch.empty((k_steps, n_envs, vec.N_AGENTS), dtype=torch.int8, device=dev)
    for k in range(k_steps):
        vec.random_actions(table[k], seed=7, step=k)

    def plain():
        for k in range(k_steps):
            vec.step_observe(table[k], auto_reset=True)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
        best = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize(dev)
            best.append(a.elapsed_time(b) / (reps * k_steps))
        return sorted(best)[2]

    t_plain = timed(plain)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream(device=dev)):
        plain()
    torch.cuda.synchronize(dev)
    t_graph = timed(graph.replay)
    t_plain2 = timed(plain)
    print(f"{name:8s} {n_envs:6d} envs, {k_steps} steps per graph: plain {t_plain:.4f} / {t_plain2:.4f} ms per step, replayed {t_graph:.4f} ms "
          f"({n_envs / t_graph / 1e3:.1f} M env-steps/s against {n_envs / min(t_plain, t_plain2) / 1e3:.1f} M)", flush=True)
    vec.close()


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    arena = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    split = dict(pkg.configs.SPLIT_KWARGS, SCENARIO=pkg.CtfScenarios.arrow)
    run("split", split, 4096, k)
    run("split", split, 1024, k)
    run("arena", arena, 4096, k)
    run("arena", arena, 65536, k)
