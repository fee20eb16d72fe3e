"""Which allocations are fast?  Many same-sized output buffers, timed; then freed and reallocated."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("marl-ctf-development_amd")
E = 65536
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(E, device=0, **kw)
acts = torch.zeros((E, 8), dtype=torch.int8, device="cuda")
orig = vec.obs
def timeit(buf, reps=40):
    vec.obs = buf
    for _ in range(6):
        vec.step(acts, auto_reset=True); vec.observe()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        vec.step(acts, auto_reset=True); a.record(); vec.observe(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
print("orig   %#x  %.4f" % (orig.data_ptr(), timeit(orig)))
for rnd in range(2):
    bufs = [torch.empty_like(orig) for _ in range(8)]
    for b in bufs:
        print("round %d %#x  %.4f ms" % (rnd, b.data_ptr(), timeit(b)))
    del bufs, b
    vec.obs = orig
    torch.cuda.empty_cache()
print("free/total GiB", [x / 2**30 for x in torch.cuda.mem_get_info()])
