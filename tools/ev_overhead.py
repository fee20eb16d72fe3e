"""What do the HIP events that bench.py records around its launches cost?  The arena step with 3 / 2 / 1 / 0 event records per step."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
pkg = importlib.import_module("marl-ctf-development_amd")
kw = bench.WORKLOADS["arena"][1](pkg)
E = 65536
vec = pkg.VecGridworldCtf(E, device=0, **kw)
acts = torch.empty((64, E, vec.N_AGENTS), dtype=torch.int8, device="cuda")
for t in range(64):
    vec.random_actions(acts[t], seed=0xC7F, step=t)
vec.observe()
bench.stagger_phases(vec, torch, 0, kw["GAME_STEPS"])
def run(K, nev):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(K):
        if nev >= 3: ev[t][0].record()
        vec.step(acts[t % 64], auto_reset=True)
        if nev >= 2: ev[t][1].record()
        vec.observe()
        if nev >= 1: ev[t][2].record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for rep in range(3):
    for nev in (3, 2, 1, 0):
        print(f"events per step {nev}: {run(200, nev):.4f} ms/step", flush=True)
