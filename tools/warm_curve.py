"""How do the kernel times evolve with time under load?  (clock ramp / thermal behaviour of the box)"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("marl-ctf-development_amd")
E = 65536
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(E, device=0, **kw)
acts = torch.empty((64, E, vec.N_AGENTS), dtype=torch.int8, device="cuda")
for t in range(64):
    vec.random_actions(acts[t], seed=1, step=t)
t_start = time.time()
for chunk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(100)]
    for i in range(100):
        ev[i][0].record(); vec.step(acts[i % 64], auto_reset=True); ev[i][1].record(); vec.observe(); ev[i][2].record()
    torch.cuda.synchronize()
    s = np.median([e[0].elapsed_time(e[1]) for e in ev]); o = np.median([e[1].elapsed_time(e[2]) for e in ev])
    print(f"t={time.time()-t_start:6.2f}s  step {s:.4f} ms  observe {o:.4f} ms", flush=True)
