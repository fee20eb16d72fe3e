"""A/B of CTF_OBS_XCD=0/1 for the wave-per-env render and the compact render in one process (0_the_split and 8_arena at 65 536 envs)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pkg = importlib.import_module("marl-ctf-development_amd")

def t(fn, reps=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

for name in ("split", "arena"):
    kw = bench.WORKLOADS[name][1](pkg)
    vec = pkg.VecGridworldCtf(65536, device=0, **kw)
    acts = torch.zeros((65536, vec.N_AGENTS), dtype=torch.int8, device="cuda")
    for s in range(30):
        vec.random_actions(acts, 3, s); vec.step(acts, auto_reset=True)
    vec.observe(); vec.observe_codes()
    for rnd in range(3):
        for x in ("0", "1"):
            os.environ["CTF_OBS_XCD"] = x
            os.environ["CTF_OBS_TILES"] = "0"
            a = t(lambda: vec.observe())
            c = t(lambda: vec.observe_codes())
            os.environ["CTF_OBS_TILES"] = "1"
            print(f"{name} XCD={x}: k_observe {a:.4f} ms   k_observe_codes {c:.4f} ms", flush=True)
    del vec
