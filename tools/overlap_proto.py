"""Prototype: does running step(chunk c+1) beside observe(chunk c) on two streams hide the step kernel?
K independent handles of E/K envs each stand in for env ranges of one handle."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("marl-ctf-development_amd")
E = 65536
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)

def run(K, steps=150, warm=30):
    vecs = [pkg.VecGridworldCtf(E // K, device=0, **kw) for _ in range(K)]
    acts = [torch.zeros((E // K, 8), dtype=torch.int8, device="cuda") for _ in range(K)]
    for k in range(K):
        vecs[k].random_actions(acts[k], seed=k, step=0)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(K)]
    def one():
        # step(c) on stream A in order; observe(c) on stream B once step(c) is done
        for k in range(K):
            with torch.cuda.stream(sA):
                vecs[k].step(acts[k], auto_reset=True)
                evs[k].record(sA)
            with torch.cuda.stream(sB):
                sB.wait_event(evs[k])
                vecs[k].observe()
        # next global step's step(0) must wait for observe(K-1)... (policy would sit in between): join
        sA.wait_stream(sB)
    for _ in range(warm):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for v in vecs:
        v.close()
    return dt * 1e3

for K in (1, 2, 4, 8, 1, 4):
    print(f"K={K}: {run(K):.4f} ms per global step -> {E / run(K) / 1e3:.1f} M env-steps/s", flush=True)
